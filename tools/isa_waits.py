#!/usr/bin/env python3
"""Static scan of the generated gfx950 code of every kernel for loads that are waited for at once (dev tool).

What it looks for is what cost the CSR kernels 2-3 % and the wave-cooperative multiply much more (profiles/r04/NOTES_ragged_kernels.md):
    a global_load followed within three instructions by `s_waitcnt vmcnt(0)` -- a load the compiler sank into the
    branch of its only use, a `cond ? A[i] : B[i]` turned into a branch around two loads, a loop-carried register
    the compiler cannot prove landed.
Dependent loads (a search, a chain walk) show up too: the list is where to LOOK, not a verdict.

usage: isa_waits.py [file.hip ...]      (default: every csgn_amd/csrc/csgn_*.hip; compiles each with hipcc -S)
"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_of(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(tempfile.mkdtemp(prefix="isa_waits_"), os.path.basename(src) + ".s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-Wno-inline-asm",
                    "-I" + ROOT + "/include", "-I" + ROOT + "/csgn_amd/csrc", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True)
        return [re.sub(r"\(.*", "", re.sub(r"^void |csgn::\(anonymous namespace\)::", "", x)) for x in r.stdout.splitlines()]
    except Exception:
        return names


def scan(path):
    rows, name, lines = [], None, []

    def flush():
        if not name:
            return
        loads = [i for i, l in enumerate(lines) if re.match(r"\s*global_load", l)]
        at_once = 0
        for i in loads:
            for j in range(i + 1, min(i + 4, len(lines))):
                if "s_waitcnt vmcnt(0)" in lines[j]:
                    at_once += 1
                    break
                if re.match(r"\s*global_(load|store)", lines[j]):
                    break
        stores = sum(1 for l in lines if re.match(r"\s*global_store", l))
        if loads:
            rows.append((name, len(loads), at_once, stores))

    for l in open(path):
        m = re.match(r"^(_Z\S+):\s", l)
        if m:
            flush()
            name, lines = m.group(1), []
        elif not l.lstrip().startswith(";"):
            lines.append(l)
    flush()
    return rows


def main():
    srcs = sys.argv[1:] or sorted(os.path.join(ROOT, "csgn_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "csgn_amd", "csrc"))
                                  if f.startswith("csgn_") and f.endswith(".hip") and f not in ("csgn_capi.hip", "csgn_shard.hip"))
    for src in srcs:
        rows = scan(asm_of(src))
        names = demangle([r[0] for r in rows])
        seen = set()
        print(os.path.basename(src))
        for (_, nl, w, ns), n in sorted(zip(rows, names), key=lambda x: -x[0][2]):
            short = re.sub(r"<.*", "", n)
            if w < 2 or short in seen:                     # one line per kernel template: its worst instantiation
                continue
            seen.add(short)
            print("    %-34s %3d loads, %3d waited for at once, %3d stores   (%s)" % (short, nl, w, ns, n[:70]))


if __name__ == "__main__":
    main()
