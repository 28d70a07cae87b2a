"""libcsgn_shard.so (include/csgn_shard.h): the native multi-GPU driver -- contiguous batch
partition + the RCCL all-gather of per-pair result term counts -- and tools/shard_mul.cpp, the
thread-per-GPU program over it (no torch, no Python in the data path).

CPU box: partition arithmetic against csgn_amd.shard (which the gloo tests cover at world sizes
2 and 3), exported symbols vs the header, the tool builds and refuses to run without a GPU.
GPU box (-m gpu): the tool runs with every visible device (world 1 on the one-GPU box), RCCL is
initialised, the gathered counts and the product digest are checked against the oracle.
"""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "csgn_amd", "lib")
TOOL_SRC = os.path.join(ROOT, "tools", "shard_mul.cpp")
TOOL = os.path.join(ROOT, "tools", "bin", "shard_mul")


@pytest.fixture(scope="module")
def shard_lib():
    from csgn_amd import build, capi
    build.build_all()
    return capi.load_shard_library()


@pytest.fixture(scope="module")
def tool(shard_lib):
    os.makedirs(os.path.dirname(TOOL), exist_ok=True)
    deps = [TOOL_SRC, os.path.join(LIBDIR, "libcsgn_shard.so"), os.path.join(LIBDIR, "libcsgn_hip.so")]
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", TOOL,
                               TOOL_SRC, "-L" + LIBDIR, "-lcsgn_hip", "-lcsgn_shard", "-lpthread",
                               "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return TOOL


def test_header_symbols_are_exported_and_bound(shard_lib):
    from csgn_amd import capi
    text = open(os.path.join(ROOT, "include", "csgn_shard.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(csgn_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 14
    for n in names:
        assert hasattr(shard_lib, n), n
    assert sorted(capi.SHARD_SIGNATURES) == names


def test_shard_lib_calls_rccl_directly_and_not_torch():
    out = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(LIBDIR, "libcsgn_shard.so")],
                         capture_output=True, text=True, check=True).stdout
    for sym in ("ncclAllGather", "ncclCommInitAll", "ncclCommInitRank", "ncclGetUniqueId", "ncclBroadcast"):
        assert sym in out, sym
    deps = subprocess.run(["readelf", "-d", os.path.join(LIBDIR, "libcsgn_shard.so")], capture_output=True,
                          text=True, check=True).stdout
    assert "librccl.so" in deps and "torch" not in deps and "c10" not in deps


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_partition_matches_the_python_driver(shard_lib, world):
    from csgn_amd.shard import owner_of, shard_range
    lo, hi = C.c_uint64(), C.c_uint64()
    for total in [0, 1, 5, 7, 8, 9, 4096, 65536, 1000003, 1 << 20, (1 << 20) + 5, 1 << 40]:
        prev = 0
        for r in range(world):
            assert shard_lib.csgn_shard_range(total, r, world, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == shard_range(total, r, world)
            assert lo.value == prev and hi.value >= lo.value       # contiguous, ordered
            assert hi.value - lo.value in (total // world, -(-total // world))
            prev = hi.value
        assert prev == total
        for p in sorted(q for q in {0, 1, total // 3, total // 2, total - 1} if 0 <= q < total):
            o = shard_lib.csgn_shard_owner(p, total, world)
            assert o == owner_of(p, total, world)
            a, b = shard_range(total, o, world)
            assert a <= p < b
    assert shard_lib.csgn_shard_range(10, 2, 2, C.byref(lo), C.byref(hi)) == -1
    assert shard_lib.csgn_shard_range(10, -1, 2, C.byref(lo), C.byref(hi)) == -1
    assert shard_lib.csgn_shard_range(10, 0, 0, C.byref(lo), C.byref(hi)) == -1
    assert shard_lib.csgn_shard_owner(10, 10, 2) == -1


def test_tool_builds_and_fails_loudly_without_gpu(tool):
    import torch
    assert os.path.exists(tool)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([tool, "--pairs", "16"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "no CPU path" in p.stderr


@pytest.mark.gpu
def test_tool_shards_a_fresh_batch_and_gathers_counts_over_rccl(tool, oracle):
    """BASELINE config 4's program on whatever devices are visible (world 1 on the one-GPU box):
    fresh 1x1 pairs, RCCL initialised, ncclAllGather of the term counts, digest = oracle's."""
    pairs, n, dl = 1 << 16, 1247, 20
    p = subprocess.run([tool, "--pairs", str(pairs), "--steps", "3"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["n_gpus"] >= 1 and r["pairs"] == pairs and r["gathered_counts_wrong"] == 0
    assert r["gathered_counts_sum"] == pairs and r["shards"][0][0] == 0 and r["shards"][-1][1] == pairs
    assert r["fresh_ciphertexts"] is True and r["decrypted_bits_wrong"] == 0
    # the same fresh ciphertexts from the restated definition: key, bits, generator streams as in the tool
    key = np.array([(i * (n // 16) + 3) % n for i in range(16)], dtype=np.uint64)
    g = np.arange(pairs, dtype=np.uint64)
    pa = ((g * np.uint64(2654435761)) >> np.uint64(13)) & np.uint64(1)
    pb = ((g * np.uint64(40503) + np.uint64(7)) >> np.uint64(5)) & np.uint64(1)
    (ka, na), (kb, nb) = oracle.rng_from_seed(1234), oracle.rng_from_seed(1235)
    a = oracle.encrypt_keyed(n, key, pa.astype(np.uint8), ka, na, 8)
    b = oracle.encrypt_keyed(n, key, pb.astype(np.uint8), kb, nb, 8)
    assert int(r["products_digest"], 16) == oracle.digest(a & b)


@pytest.mark.gpu
def test_tool_all_pairs_shape_through_an_arena(tool, oracle):
    """64x64-term products streamed through a 4-slot arena; the digest covers the last launch."""
    pairs, t, n, dl = 10, 64, 1247, 20
    p = subprocess.run([tool, "--pairs", str(pairs), "--terms", str(t), "--slots", "4", "--gpus", "1", "--steps", "2"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["gathered_counts_sum"] == pairs * t * t and r["gathered_counts_wrong"] == 0
    a = oracle.synth(0x43534743 + 1, n, 0, pairs * t * dl)
    b = oracle.synth(0x43534743 + 2, n, 0, pairs * t * dl)
    want = 0
    per = t * t * dl
    for q in (8, 9):                                   # the last launch holds pairs 8 and 9
        prod, _ = oracle.mul(n, a[q * t * dl:(q + 1) * t * dl], b[q * t * dl:(q + 1) * t * dl])
        want = (want + oracle.digest(prod, q * per)) & (2**64 - 1)
    assert int(r["products_digest"], 16) == want


@pytest.mark.gpu
def test_native_comm_from_python_world_1(shard_lib):
    """csgn_comm_init_rank + gather + barrier through ctypes (the form bench.py uses per rank)."""
    import torch
    from csgn_amd import capi
    ident = C.create_string_buffer(capi.CSGN_COMM_ID_BYTES)
    capi.check_shard(shard_lib.csgn_comm_unique_id(ident))
    comm = C.c_void_p()
    capi.check_shard(shard_lib.csgn_comm_init_rank(ident, 0, 1, 0, C.byref(comm)))
    try:
        assert shard_lib.csgn_comm_world(comm) == 1 and shard_lib.csgn_comm_rank(comm) == 0
        local = torch.arange(1000, dtype=torch.int64, device="cuda") * 3
        out = torch.zeros(1000, dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        capi.check_shard(shard_lib.csgn_comm_gather_counts(comm, local.data_ptr(), 1000, out.data_ptr(), stream))
        capi.check_shard(shard_lib.csgn_comm_barrier(comm, stream))
        assert torch.equal(out, local)
        cnt = torch.zeros(77, dtype=torch.int64, device="cuda")
        capi.check_shard(shard_lib.csgn_shard_product_counts(77, None, None, 5, 9, cnt.data_ptr(), stream))
        torch.cuda.synchronize()
        assert bool((cnt == 45).all())
    finally:
        shard_lib.csgn_comm_destroy(comm)
