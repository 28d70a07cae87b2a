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


def test_mul_dispatch_names(lib, monkeypatch):
    """csgn_mul_uniform_kernel reports the measured dispatch rule without a GPU."""
    for k in ("CSGN_MUL_FLAT", "CSGN_MUL_TOUCH"):
        monkeypatch.delenv(k, raising=False)
    name = lambda n, t1, t2, pairs=1 << 16: lib.csgn_mul_uniform_kernel(n, pairs, t1, t2).decode()
    assert name(1247, 1, 1) == "k_and_stream"
    assert name(1247, 1024, 1024, 128) == "k_touch+k_mul_flat"     # the bench launch: 128 pairs, 42 MB of operands
    assert name(1247, 1024, 1024, 1) == "k_mul_tiled"              # a single product is not a stream
    assert name(4096, 256, 256) == "k_touch+k_mul_flat"
    assert name(1247, 8, 8) == "k_touch+k_mul_flat"                # output = 4x operands
    assert name(1247, 64, 64, 1) == "k_mul_tiled"
    assert name(1247, 4, 4) == "k_mul_flat"                        # operands too large a share to read twice
    assert name(1247, 1024, 1) == "k_mul_flat"                     # rows shorter than a workgroup
    assert name(1247, 2, 383) == "k_mul_tiled"                     # thin product, long rows
    assert name(1300, 128, 128) == "k_mul_tiled"                   # odd dL: 8-byte units
    assert name(1300, 200, 2) == "k_mul_flat"
    monkeypatch.setenv("CSGN_MUL_FLAT", "-1")
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
