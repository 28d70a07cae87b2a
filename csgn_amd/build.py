"""Build the in-tree native libraries for gfx950.

    python -m csgn_amd.build            # libcsgn_hip.so, libcsgn_shard.so, libcertFHE.so

hipcc cross-compiles without a GPU; the .so files land in csgn_amd/lib/ (git-ignored, but
they travel to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(LIBDIR, "libcsgn_hip.so")
CERTFHE_LIB = os.path.join(LIBDIR, "libcertFHE.so")
SHARD_LIB = os.path.join(LIBDIR, "libcsgn_shard.so")
CERTFHE_SHARD_LIB = os.path.join(LIBDIR, "libcertFHE_shard.so")

HIP_SOURCES = ["csgn_capi.hip", "csgn_circuit.hip", "csgn_mul.hip", "csgn_add.hip", "csgn_smallops.hip", "csgn_decrypt.hip", "csgn_encrypt.hip",
               "csgn_permute.hip", "csgn_compact.hip", "csgn_harness.hip", "csgn_bitlen.hip", "csgn_tuning.cpp"]
HIP_HEADERS = ["csgn_common.h", "csgn_kernels.h", "csgn_device.h", "csgn_tuning.h", "csgn_capi_util.h"]
OBJDIR = os.path.join(LIBDIR, "obj")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def check_coop_isa(asm_path=None) -> None:
    """The wave-cooperative ragged multiply keeps loads outside the compiler's books (inline-assembly loads into a
    reserved v127, hand-counted s_waitcnt): tools/check_coop_isa.py reads the generated code and fails on any
    instruction that touches a loaded register before its wait.  Run wherever the library is compiled:
    a toolchain that breaks the invariant must break the build, not the products."""
    tool = os.path.join(ROOT, "tools", "check_coop_isa.py")
    subprocess.check_call([sys.executable, tool] + ([asm_path] if asm_path else []))


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """One object per translation unit (compiled in parallel, rebuilt only when it or a header changed), then the
    link; csgn_mul.hip is also compiled to assembly for the ISA check of the wave-cooperative kernel."""
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(INCLUDE, "csgn_hip.h")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-inline-asm",
             "-I" + INCLUDE, "-I" + CSRC]
    jobs, objs = [], []
    for name in HIP_SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJDIR, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [_hipcc()] + flags + (["-x", "hip"] if name.endswith(".cpp") else []) + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    mul_asm = os.path.join(OBJDIR, "csgn_mul.s")
    mul_src = os.path.join(CSRC, "csgn_mul.hip")
    checked = os.path.join(OBJDIR, "coop_isa.ok")
    asm_job = None
    if force or _stale(mul_asm, [mul_src] + hdrs):
        cmd = [_hipcc()] + flags + ["-S", "--cuda-device-only", "-o", mul_asm, mul_src]
        asm_job = (cmd, subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
    for cmd, proc in jobs + ([asm_job] if asm_job else []):
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if force or _stale(checked, [mul_asm, os.path.join(ROOT, "tools", "check_coop_isa.py")]):
        check_coop_isa(mul_asm)
        open(checked, "w").write("ok\n")
    if force or _stale(HIP_LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


def build_shard(force: bool = False, verbose: bool = False) -> str:
    """libcsgn_shard.so (include/csgn_shard.h): batch partition + the RCCL all-gather of result term
    counts.  Links librccl directly; kept apart from libcsgn_hip.so so that single-GPU users do
    not map the 570 MB RCCL image."""
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "csgn_shard.hip")
    deps = [src, os.path.join(INCLUDE, "csgn_shard.h"), os.path.join(INCLUDE, "csgn_hip.h")]
    if force or _stale(SHARD_LIB, deps):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(_hipcc())), "lib")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + INCLUDE,
               "-o", SHARD_LIB, src, "-L" + rocm_lib, "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SHARD_LIB


def build_certfhe(force: bool = False, verbose: bool = False):
    """The drop-in certFHE:: C++ classes over the C ABI (csgn_amd/csrc/certfhe/*.cpp)."""
    src_dir = os.path.join(CSRC, "certfhe")
    if not os.path.isdir(src_dir):
        return None
    srcs = sorted(os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".cpp"))
    if not srcs:
        return None
    hdr_dir = os.path.join(INCLUDE, "certfhe")
    hdrs = [os.path.join(hdr_dir, f) for f in os.listdir(hdr_dir)] if os.path.isdir(hdr_dir) else []
    if force or _stale(CERTFHE_LIB, srcs + hdrs + [HIP_LIB]):
        cmd = ["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-I" + INCLUDE, "-I" + hdr_dir,
               "-o", CERTFHE_LIB] + srcs + ["-L" + LIBDIR, "-lcsgn_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CERTFHE_LIB


def build_certfhe_shard(force: bool = False, verbose: bool = False):
    """certFHE::ShardGroup / ShardedBatch (include/certfhe/ShardedBatch.h): the class-level face of the
    sharded batch, over libcsgn_hip.so + libcsgn_shard.so.  A library of its own so that
    libcertFHE.so does not pull in RCCL."""
    src_dir = os.path.join(CSRC, "certfhe_shard")
    srcs = sorted(os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".cpp"))
    hdr_dir = os.path.join(INCLUDE, "certfhe")
    hdrs = [os.path.join(hdr_dir, f) for f in os.listdir(hdr_dir)] + [os.path.join(INCLUDE, "csgn_shard.h")]
    if force or _stale(CERTFHE_SHARD_LIB, srcs + hdrs + [HIP_LIB, SHARD_LIB, CERTFHE_LIB]):
        cmd = ["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-pthread", "-I" + INCLUDE, "-I" + hdr_dir,
               "-o", CERTFHE_SHARD_LIB] + srcs + ["-L" + LIBDIR, "-lcertFHE", "-lcsgn_shard", "-lcsgn_hip",
                                                  "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CERTFHE_SHARD_LIB


BENCH_NATIVE = os.path.join(ROOT, "tools", "bin", "bench_native")


def build_bench_native(force: bool = False, verbose: bool = False) -> str:
    """tools/bin/bench_native: bench.py's measurement as a torch-free C++ program over the C ABI (thread per GPU,
    strict RCCL); bench.py --native-ranks starts it."""
    src = os.path.join(ROOT, "tools", "bench_native.cpp")
    os.makedirs(os.path.dirname(BENCH_NATIVE), exist_ok=True)
    deps = [src, HIP_LIB, SHARD_LIB, os.path.join(INCLUDE, "csgn_hip.h"), os.path.join(INCLUDE, "csgn_shard.h")]
    if force or _stale(BENCH_NATIVE, deps):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(_hipcc())), "lib")
        cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-I" + INCLUDE, "-o", BENCH_NATIVE, src, "-L" + LIBDIR,
               "-lcsgn_shard", "-lcsgn_hip", "-lpthread", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + rocm_lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return BENCH_NATIVE


def build_all(force: bool = False, verbose: bool = False):
    out = [build_hip(force, verbose), build_shard(force, verbose)]
    c = build_certfhe(force, verbose)
    if c:
        out.append(c)
        out.append(build_certfhe_shard(force, verbose))
    out.append(build_bench_native(force, verbose))
    return out


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv, verbose=True)))
