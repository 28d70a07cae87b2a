#!/usr/bin/env python3
"""Static check of k_mul_ragged_coop<.., PIPE = true> (csgn_amd/csrc/csgn_mul.hip): its operand loads are inline
assembly the compiler does not know to be asynchronous, and the waits in front of their use are hand-counted.  What
must hold in the generated code, on every path from such a load:

    no instruction reads or writes the load's destination registers before an `s_waitcnt vmcnt(N)` that retires it
    (N <= the vector-memory instructions issued after the load on that path; vmcnt counts loads and stores together,
    in issue order).

The compiler keeps the registers reserved (it believes the value is already there) but is free to COPY them -- live
range splitting, a phi -- and a copy in front of the wait would read a register the load has not written yet.  This
script compiles the file to gfx950 assembly, walks the control-flow graph from every inline-assembly load and fails if
that happens.  It also checks that v127 -- the one destination of the kernel's operand touches (loads whose results
nobody reads) -- appears nowhere else.  The stores between the loads are taken as issued (a store is skipped only when no lane of its block is
live, which the kernel rules out for a group filled before the end of its stretch: see the kernel's comment).

usage: check_coop_isa.py [file.s]     (without an argument: compiles csgn_mul.hip with hipcc; exit code 1 on a finding)
"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(tempfile.mkdtemp(prefix="coop_isa_"), "mul.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-Wno-inline-asm", "-I" + ROOT + "/include",
                    "-I" + ROOT + "/csgn_amd/csrc", "-S", "--cuda-device-only", "-o", out,
                    ROOT + "/csgn_amd/csrc/csgn_mul.hip"], check=True, stderr=subprocess.DEVNULL)
    return out


def kernels(path):
    """{mangled name: [lines]} of every instantiation (PIPE = true: ...Lb1E...)"""
    res, cur, name = {}, None, None
    for ln in open(path):
        m = re.match(r"^(_ZN4csgn\S*k_mul_ragged_coop\S*):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.lstrip().startswith(".amdhsa_kernel") or re.match(r"^\.Lfunc_end", ln):
                res[name] = cur
                cur = None
            else:
                cur.append(ln.rstrip("\n"))
    return res


def vregs(text):
    """VGPR numbers an operand string mentions"""
    s = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        s.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        s.add(int(a))
    return s


def check(name, lines):
    # instructions: (text, in_asm), labels -> index
    ins, labels, in_asm = [], {}, False
    for ln in lines:
        t = ln.split(";")[0].strip() if "#ASM" not in ln else ln.strip()
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not t or t.startswith("."):
            m = re.match(r"^(\.LBB\S+):", t)
            if m:
                labels[m.group(1)] = len(ins)
            continue
        ins.append((t, in_asm))
    label_at = set(labels.values())
    is_vmem = lambda t: re.match(r"^(global|buffer|flat|scratch)_(load|store|atomic)", t) is not None
    has_store_before = lambda i, stop: any(re.match(r"^global_store", ins[j][0]) for j in range(i, min(stop, len(ins))))

    def block_end(i):                     # next branch or label boundary after i (for the execnz case)
        j = i
        while j < len(ins) and not re.match(r"^s_(c?branch|endpgm)", ins[j][0]):
            j += 1
        return j + 1

    def exec_written_in_block(i):
        j = i - 1
        while j >= 0 and j not in label_at and not re.match(r"^s_(c?branch|endpgm)", ins[j][0]):
            if re.match(r"^s_\w+saveexec|^s_\w+\s+exec\b", ins[j][0]):
                return True
            j -= 1
        if j >= 0 and j in label_at and not re.match(r"^s_(c?branch|endpgm)", ins[j][0]):
            return bool(re.match(r"^s_\w+saveexec|^s_\w+\s+exec\b", ins[j][0]))
        return False

    def succ(i):
        t = ins[i][0]
        if t.startswith("s_endpgm"):
            return []
        m = re.match(r"^s_(c?branch\S*)\s+(\.LBB\S+)", t)
        if not m:
            return [i + 1]
        kind, tgt = m.group(1), labels[m.group(2)]
        if kind == "branch":
            return [tgt]
        if kind in ("cbranch_execnz", "cbranch_execz") and not exec_written_in_block(i):
            # exec untouched since the block began: the structurizer's way of writing "always" / "never" (a wave
            # does not run with an empty exec mask outside a region that a skip branch guards)
            return [tgt] if kind == "cbranch_execnz" else [i + 1]
        if kind == "cbranch_execz" and has_store_before(i + 1, block_end(i + 1)):
            return [i + 1]                # the skip around a store: not taken (see the docstring)
        if kind == "cbranch_execnz" and has_store_before(tgt, block_end(tgt)):
            return [tgt]
        return [i + 1, tgt]

    findings, nloads = [], 0
    # the operand touches (loads whose results nobody reads) all write v127, which the kernel keeps out of the register
    # allocator's hands (amdgpu_num_vgpr): nothing else may mention it, or a touch still in flight would land in a live value
    for t, in_asm in ins:
        if 127 in vregs(t.split(None, 1)[1] if " " in t else "") and not (in_asm and t.startswith("global_load_dword v127,")):
            findings.append("%s: `%s` uses v127, the destination of the operand touches" % (name[:60], t))
    for i0, (t0, asm0) in enumerate(ins):
        m = re.match(r"^global_load_dword(?:x[24])?\s+(v\[\d+:\d+\]|v\d+)", t0)
        if not (asm0 and m) or m.group(1) == "v127":
            continue
        nloads += 1
        dest = vregs(m.group(1))
        best = {}                          # index -> smallest count of younger vector-memory instructions seen
        work = [(j, 0, i0) for j in succ(i0)]
        parent = {}
        while work:
            i, cnt, par = work.pop()
            if i >= len(ins) or (i in best and best[i] <= cnt):
                continue
            best[i] = cnt
            parent[i] = par
            t = ins[i][0]
            w = re.match(r"^s_waitcnt.*vmcnt\((\d+)\)", t)
            if w and cnt >= int(w.group(1)):
                continue                   # retired on this path
            again = ins[i][1] and re.match(r"^global_load_dword(?:x[24])?\s+(v\[\d+:\d+\]|v\d+)", t)
            if again and vregs(again.group(1)) & dest and not vregs(t.split(",", 1)[1]) & dest:
                continue                   # the same registers loaded again: loads return in order, this walk ends, that one's begins
            if t.startswith("v_readfirstlane_b32") and ((i + 1 < len(ins) and ins[i + 1][0].startswith("v_readfirstlane_b32")) or
                                                        ins[i - 1][0].startswith("v_readfirstlane_b32")):
                work.extend((j, cnt, i) for j in succ(i))
                continue                   # a run of readfirstlanes of one register: how the compiler materialises undefined scalars
            if i != i0 and vregs(t.split(None, 1)[1] if " " in t else "") & dest:
                findings.append("%s: `%s` touches the destination of `%s` before its wait (younger vmem ops: %d)"
                                % (name[:60], t, t0, cnt))
                if os.environ.get("COOP_ISA_TRACE"):
                    chain, j = [], i
                    while j != i0:
                        if is_vmem(ins[j][0]) or re.match(r"^s_(c?branch|waitcnt.*vmcnt)", ins[j][0]):
                            chain.append("%d:%s" % (j, ins[j][0][:48]))
                        j = parent[j]
                    print("      path:", " <- ".join(chain))
                continue
            work.extend((j, cnt + (1 if is_vmem(t) else 0), i) for j in succ(i))
    return nloads, findings


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
    ks = kernels(path)
    if not ks:
        print("no k_mul_ragged_coop<..., true> kernel found in", path)
        return 1
    bad = 0
    for name, lines in sorted(ks.items()):
        n, f = check(name, lines)
        print("%-70s %3d inline-assembly loads, %d findings" % (name[:70], n, len(f)))
        for x in f[:10]:
            print("   ", x)
        bad += len(f)
        if n == 0 and "Lb1E" in name:
            print("    no inline-assembly load found: the kernel changed, update this check")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
