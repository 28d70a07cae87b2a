"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/csgn_hip.h
declares, its host-only metadata helpers agree with the oracle, and -- on a box without a
GPU -- compute entry points fail loudly instead of falling back to the CPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from csgn_amd import build, capi
    build.build_hip()
    return capi.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "csgn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csgn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from csgn_amd import capi
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in csgn_hip.h but not exported"
        assert name in capi.SIGNATURES, f"{name} missing from csgn_amd.capi.SIGNATURES"
    assert sorted(capi.SIGNATURES) == names
    assert lib.csgn_abi_version() == 1


def test_no_torch_types_or_oracle_in_the_product():
    """The boundary is plain C; the product never reaches into oracle/."""
    hdr = open(os.path.join(ROOT, "include", "csgn_hip.h")).read()
    assert "torch" not in hdr and "at::" not in hdr
    for base, _, files in os.walk(os.path.join(ROOT, "csgn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                assert "oracle.binding" not in src and "libcsgn_oracle" not in src and "csgn_oracle_" not in src, f
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "csgn_oracle_" not in open(os.path.join(base, f)).read(), f


@pytest.mark.parametrize("n", [1, 63, 64, 65, 128, 129, 1247, 4096, 4097])
def test_metadata_helpers_match_oracle(lib, oracle, n):
    assert lib.csgn_default_len(n) == oracle.default_len(n)
    dl = oracle.default_len(n)
    for d in (1, 4, 16):
        assert lib.csgn_context_s(n, d) == oracle.context_s(n, d)
    for (l1, l2) in [(dl, dl), (dl, 2 * dl), (3 * dl, 5 * dl), (0, dl), (dl, 0)]:
        assert lib.csgn_mul_len(n, l1, l2) == oracle.lib.csgn_oracle_mul_len(dl, l1, l2)
    bl = np.zeros(3 * dl, dtype=np.uint64)
    assert lib.csgn_bitlen_canonical(n, 3, bl.ctypes.data) == 0
    assert np.array_equal(bl, oracle.bitlen(n, 3))
    if n >= 16:
        key = np.random.default_rng(n).permutation(n)[:min(16, n)].astype(np.uint64)
        mask = np.zeros(dl, dtype=np.uint64)
        assert lib.csgn_key_mask(n, key.ctypes.data, key.size, mask.ctypes.data) == 0
        assert np.array_equal(mask, oracle.key_mask(n, key))
        bad = key.copy()
        bad[0] = n
        assert lib.csgn_key_mask(n, bad.ctypes.data, bad.size, mask.ctypes.data) == -1
        assert b"outside" in lib.csgn_last_error()


def test_mul_dispatch_names(lib, knobs):
    """csgn_mul_uniform_kernel reports the measured dispatch rule without a GPU."""
    for k in ("CSGN_MUL_FLAT", "CSGN_MUL_TOUCH"):
        knobs.unset(k)
    name = lambda n, t1, t2, pairs=1 << 16: lib.csgn_mul_uniform_kernel(n, pairs, t1, t2).decode()
    assert name(1247, 1, 1) == "k_and_stream"
    assert name(1247, 1024, 1024, 128) == "k_touch+k_mul_flat"     # the bench launch: 128 pairs, 42 MB of operands
    assert name(1247, 1024, 1024, 1) == "k_mul_tiled"              # a single product is not a stream
    assert name(4096, 256, 256) == "k_touch+k_mul_flat"
    assert name(1247, 16, 8) == "k_touch+k_mul_flat"               # output >= 4x operands: touch + flat
    assert name(1247, 8, 8) == "k_mul_tiled"                       # a whole small pair per 128-thread workgroup
    assert name(1247, 64, 64, 1) == "k_mul_tiled"
    assert name(1247, 4, 4) == "k_mul_tiled"                       # thin and small: 64-thread workgroups, 8 rows per tile
    assert name(1247, 2, 2) == "k_mul_flat"                        # two rows: the flat kernel
    assert name(1247, 1024, 1) == "k_mul_flat"                     # rows of 10 units: the flat kernel
    assert name(4096, 64, 4) == "k_mul_tiled" and name(4096, 2, 2) == "k_mul_flat"
    assert name(1247, 2, 383) == "k_mul_tiled"                     # thin product, long rows
    assert name(1300, 128, 128) == "k_mul_tiled"                   # odd dL: 8-byte units
    assert name(1300, 200, 2) == "k_mul_flat"
    knobs.set("CSGN_MUL_FLAT", "-1")
    assert name(1247, 1024, 1024, 128) == "k_mul_tiled"


def test_fastdiv_helper_is_exact(lib):
    rng = np.random.default_rng(0)
    ds = list(range(1, 130)) + [320, 1000, 4095, 4096, 4097, 10240, 65535, 65536, 3276800, 2**31 - 1, 2**31, 2**32 - 1]
    for d in ds:
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2**31 - 1, 2**31, 2**32 - 1] + rng.integers(0, 2**32, 40).tolist()
        for n in ns:
            n &= 0xFFFFFFFF
            assert lib.csgn_debug_fastdiv(n, d) == n // d, (n, d)


def test_fails_loudly_without_gpu(lib):
    """No CPU fallback: on a box with no HIP device csgn_init and compute calls report
    CSGN_ERR_NO_DEVICE / HIP errors.  (On a GPU box this test just checks init succeeds.)"""
    import torch
    n = C.c_int(0)
    rc = lib.csgn_device_count(C.byref(n))
    if torch.cuda.is_available():
        assert rc == 0 and n.value >= 1
        return
    assert rc == -3 and n.value == 0
    assert lib.csgn_init(0) == -3
    assert b"no CPU fallback" in lib.csgn_last_error()
    from csgn_amd.capi import CsgnError
    from csgn_amd.batch import HipPath
    with pytest.raises(CsgnError):
        HipPath(0)
    buf = np.zeros(64, dtype=np.uint64)
    rc = lib.csgn_mul_uniform(1247, 1, 1, 1, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, 0, None)
    assert rc < 0, "compute call must not succeed without a GPU"
    # a circuit can be described on the host, but building it (allocation + graph capture) cannot succeed
    c = C.c_void_p()
    assert lib.csgn_circuit_create(1247, 4, C.byref(c)) == 0
    a, b, r = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert lib.csgn_circuit_input(c, 1, C.byref(a)) == 0 and lib.csgn_circuit_input(c, 2, C.byref(b)) == 0
    assert lib.csgn_circuit_mul(c, a, b, C.byref(r)) == 0 and lib.csgn_circuit_value_terms(c, r) == 2
    assert lib.csgn_circuit_add(c, r, 17, C.byref(r)) == -1         # no such value
    assert lib.csgn_circuit_build(c) < 0
    assert lib.csgn_circuit_run(c, None) == -1
    lib.csgn_circuit_destroy(c)


def test_integration_md_binding_stub_compiles(tmp_path):
    """The reference-side binding shown in INTEGRATION.md (section B) is real code: extract it
    and compile it against include/csgn_hip.h and the Context class."""
    import subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "namespace hipbind" in b]
    assert len(stub) == 1
    src = tmp_path / "binding_stub.cpp"
    src.write_text('#include <vector>\n#include "Context.h"\n' + stub[0] +
                   "\nint main() { certFHE::Context c(1247, 16); uint64_t n = 0; (void)c; (void)n;"
                   " (void)&certFHE::hipbind::multiply; (void)&certFHE::hipbind::add;"
                   " (void)&certFHE::hipbind::decrypt; return 0; }\n")
    p = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "include", "certfhe"), str(src)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_integration_md_sharded_class_snippet_compiles(tmp_path):
    """INTEGRATION.md section C's C++ example of certFHE::ShardedBatch is real code against the shipped
    header: wrapped into a function and compiled (syntax and types only)."""
    import subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```cpp\n(.*?)```", text, flags=re.S) if "ShardGroup gpus" in b]
    assert len(blocks) == 1
    lines = blocks[0].splitlines()
    head = [l for l in lines if l.startswith("#include") or l.startswith("using namespace")]
    body = [l for l in lines if l not in head]
    src = tmp_path / "sharded_snippet.cpp"
    src.write_text("\n".join(head) + "\nvoid example(const SecretKey &key, const std::vector<unsigned char> &bits_a,\n"
                   "             const std::vector<unsigned char> &bits_b, uint64_t seed_a, uint64_t seed_b, const Permutation &perm)\n{\n"
                   + "\n".join(body) + "\n(void)counts; (void)plain; (void)fast;\n}\nint main() { return 0; }\n")
    p = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "include", "certfhe"), str(src)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_tuning_knobs_api(lib, knobs):
    """csgn_set_tuning / csgn_get_tuning / csgn_reset_tuning: per-thread knobs, no environment
    reads after load."""
    from csgn_amd import capi
    names = capi.tuning_names()
    assert {"mul_flat", "mul_touch", "ragged_c", "perm_ballot", "dec_loop", "enc_lds"} <= set(names)
    assert len(names) == len(set(names))
    before = capi.get_tuning("mul_ti")
    capi.set_tuning("mul_ti", 9)
    assert capi.get_tuning("mul_ti") == 9 and capi.get_tuning("CSGN_MUL_TI") == 9
    os.environ["CSGN_MUL_TI"] = "77"                 # the environment is NOT consulted again
    try:
        assert capi.get_tuning("mul_ti") == 9
        capi.reset_tuning()
        assert capi.get_tuning("mul_ti") == before
    finally:
        del os.environ["CSGN_MUL_TI"]
    v = C.c_int(0)
    assert lib.csgn_set_tuning(b"no_such_knob", 1) == -1 and b"no_such_knob" in lib.csgn_last_error()
    assert lib.csgn_get_tuning(b"no_such_knob", C.byref(v)) == -1
    assert lib.csgn_set_tuning(None, 1) == -1


def test_tuning_knobs_are_per_host_thread(lib, knobs):
    """a knob set by one host thread must not change another thread's dispatch (the C ABI
    is called from one host thread per GPU).  Each thread starts from the defaults and keeps its own values."""
    import threading
    from csgn_amd import capi
    default = capi.get_tuning("mul_flat")
    capi.set_tuning("mul_flat", -1)
    seen = {}

    def other():
        seen["start"] = capi.get_tuning("mul_flat")            # NOT the main thread's -1
        capi.set_tuning("mul_flat", 3)
        capi.set_tuning("mul_touch", 0)
        capi.set_tuning("perm_ballot", 1)
        seen["own"] = capi.get_tuning("mul_flat")
        # dispatch follows the thread's own knobs (no GPU needed for the name)
        seen["kernel"] = lib.csgn_mul_uniform_kernel(1247, 128, 1024, 1024).decode()

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["start"] == default and seen["own"] == 3 and seen["kernel"] == "k_mul_flat"
    assert capi.get_tuning("mul_flat") == -1 and capi.get_tuning("perm_ballot") == 0
    assert lib.csgn_mul_uniform_kernel(1247, 128, 1024, 1024).decode() == "k_mul_tiled"
    capi.reset_tuning()
    assert lib.csgn_mul_uniform_kernel(1247, 128, 1024, 1024).decode() == "k_touch+k_mul_flat"
    # a caller that shares its GPU asks for the kernel that does not lean on the memory-side cache
    capi.set_tuning("shared_gpu", 1)
    assert lib.csgn_mul_uniform_kernel(1247, 128, 1024, 1024).decode() == "k_mul_tiled"
    capi.reset_tuning()


def test_environment_is_read_in_one_place_only():
    """getenv appears in csgn_tuning.cpp (a load-time snapshot) and nowhere else in the library."""
    src = os.path.join(ROOT, "csgn_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(src)):
        p = os.path.join(src, f)
        if os.path.isfile(p) and "getenv(" in open(p).read():
            hits.append(f)
    assert hits == ["csgn_tuning.cpp"], hits


def test_size_guards_do_not_wrap(lib):
    """Shape limits are evaluated without 64-bit wrap-around and before any device call
    (t1 = t2 = 2^30 at dL = 16 wrapped to 0 and passed)."""
    p = 0x1000                                        # never dereferenced: the guards come first
    assert lib.csgn_mul_uniform(1024, 1, 1 << 30, 1 << 30, p, p, p, 0, None) == -2
    assert lib.csgn_mul_uniform(1247, 1, 1 << 20, 1 << 20, p, p, p, 0, None) == -2
    assert lib.csgn_mul_uniform(1247, 1 << 62, 4, 4, p, p, p, 0, None) == -2
    assert lib.csgn_mul_ragged(1024, 5, p, p, p, p, p, p, 1 << 30, 1 << 30, 7, None) == -2
    assert lib.csgn_add_uniform(1247, 1, 1 << 63, 1 << 63, p, p, p, None) == -2
    assert lib.csgn_add_uniform(1247, 1 << 61, 100, 100, p, p, p, None) == -2
    assert lib.csgn_decrypt_uniform(1247, 1 << 40, 1 << 40, p, p, p, p, None) == -2
    assert lib.csgn_decrypt_product_uniform(1247, 1 << 40, 1 << 40, 1, p, p, p, p, p, None) == -2
    # compaction keeps slot indices in 32 bits: 2^31 terms or more are refused
    assert lib.csgn_compact_ragged(1247, 3, 1 << 31, 0, p, p, p + 0x100000, p, p, None) == -2
    assert b"2^31" in lib.csgn_last_error()
    c = C.c_void_p()
    assert lib.csgn_circuit_create(1024, 1 << 40, C.byref(c)) == 0
    v = C.c_uint32(0)
    assert lib.csgn_circuit_input(c, 1 << 30, C.byref(v)) == -2
    assert lib.csgn_circuit_input(c, 1, C.byref(v)) == 0
    lib.csgn_circuit_destroy(c)
    c = C.c_void_p()
    assert lib.csgn_circuit_create(1024, 2, C.byref(c)) == 0
    a, b, o = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    assert lib.csgn_circuit_input(c, 1 << 30, C.byref(a)) == 0
    assert lib.csgn_circuit_input(c, 1 << 30, C.byref(b)) == 0
    assert lib.csgn_circuit_mul(c, a, b, C.byref(o)) == -2
    lib.csgn_circuit_destroy(c)


def test_coop_kernel_isa_keeps_loaded_registers_untouched():
    """k_mul_ragged_coop's pipelined form issues its operand loads from inline assembly and retires them with
    hand-counted `s_waitcnt vmcnt(N)`: the compiler believes the loaded registers hold their values at once, so a copy
    of one (a live-range split, a phi) placed in front of the wait would read a register the load has not written yet.
    tools/check_coop_isa.py compiles csgn_mul.hip to gfx950 assembly and walks the control-flow graph from every such
    load to the wait that retires it; no instruction on the way may touch the destination."""
    import shutil
    import subprocess
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this machine")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_coop_isa.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 findings") == 8, r.stdout          # unit16 / unit8 x 2 / 4 blocks per group x pipelined / plain
