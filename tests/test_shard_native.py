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
    deps = [TOOL_SRC, os.path.join(LIBDIR, "libcsgn_shard.so"), os.path.join(LIBDIR, "libcsgn_hip.so"),
            os.path.join(LIBDIR, "libcertFHE_shard.so"), os.path.join(LIBDIR, "libcertFHE.so")]
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < max(os.path.getmtime(d) for d in deps):
        inc = os.path.join(ROOT, "include")
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-I" + inc, "-I" + os.path.join(inc, "certfhe"),
                               "-o", TOOL, TOOL_SRC, "-L" + LIBDIR, "-lcertFHE_shard", "-lcertFHE", "-lcsgn_shard",
                               "-lcsgn_hip", "-lpthread", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
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
    for sym in ("ncclAllGather", "ncclCommInitAll", "ncclCommInitRank", "ncclGetUniqueId", "ncclBroadcast",
                "ncclGetVersion", "ncclCommAbort", "ncclCommGetAsyncError"):
        assert sym in out, sym
    deps = subprocess.run(["readelf", "-d", os.path.join(LIBDIR, "libcsgn_shard.so")], capture_output=True,
                          text=True, check=True).stdout
    needed = " ".join(l for l in deps.splitlines() if "NEEDED" in l)
    assert "librccl.so" in needed and "torch" not in needed and "c10" not in needed


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


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
def test_gather_plan_offsets_and_lengths(shard_lib, world):
    """The arithmetic csgn_comm_gather_* run on (csgn_shard_gather_plan), for every world size of one
    node: slices are the contiguous shard ranges, cover [0, total) exactly, and `equal` (one
    ncclAllGather) is set iff every rank contributes the same count -- otherwise the grouped
    broadcast form runs, whose per-rank (offset, length) these are."""
    from csgn_amd.shard import shard_range
    lo, ln = (C.c_uint64 * world)(), (C.c_uint64 * world)()
    eq = C.c_int(-1)
    for total in [0, 1, 2, 7, 8, 9, 63, 64, 65, 1000, 65536, 65537, 1 << 20, (1 << 20) + 1, (1 << 20) + 7, (1 << 33) + 3]:
        assert shard_lib.csgn_shard_gather_plan(total, world, lo, ln, C.byref(eq)) == 0
        pos = 0
        for r in range(world):
            a, b = shard_range(total, r, world)
            assert (lo[r], ln[r]) == (a, b - a)
            assert lo[r] == pos                       # no gap, no overlap: rank r's slice starts where r-1's ends
            pos += ln[r]
            assert ln[r] in (total // world, -(-total // world))
        assert pos == total
        assert eq.value == (1 if len(set(ln)) == 1 else 0)
        assert eq.value == 1 if total % world == 0 else True
    assert shard_lib.csgn_shard_gather_plan(10, 0, lo, ln, C.byref(eq)) == -1


def test_rccl_version_is_reported_and_a_skew_is_refused(shard_lib):
    """csgn_comm_init_* compare ncclGetVersion() of the librccl the process REALLY bound with the
    header the library was built against.  In this (torch) process that is torch's bundled copy: a
    different minor, so the strict form must refuse before touching any device."""
    from csgn_amd import capi
    rt, hd, path = capi.rccl_info()
    assert rt // 10000 == hd // 10000 == 2 and os.path.exists(path) and "rccl" in os.path.basename(path)
    ident = C.create_string_buffer(capi.CSGN_COMM_ID_BYTES)
    comm = C.c_void_p()
    rc = shard_lib.csgn_comm_init_rank(ident, 0, 1, 0, C.byref(comm))
    msg = shard_lib.csgn_shard_last_error().decode()
    if (rt // 100) != (hd // 100):
        assert rc == capi.CSGN_ERR_UNSUPPORTED and "minor version mismatch" in msg and path in msg
        assert not comm.value
    else:                                               # same RCCL: only the missing GPU can stop it here
        import torch
        assert rc == 0 or not torch.cuda.is_available()
        if rc == 0:
            shard_lib.csgn_comm_destroy(comm)


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
@pytest.mark.parametrize("extra", [["--fused", "1"], ["--force-uneven", "1"], ["--fused", "1", "--force-uneven", "1"]])
def test_tool_fused_chain_and_uneven_gather_give_the_same_products(tool, oracle, extra):
    """--fused: Enc*Enc in one kernel per GPU (csgn_encrypt_mul_keyed); --force-uneven: the grouped
    ncclBroadcast form of both gathers.  Same digest as the restated definition either way."""
    pairs, n = 50000, 1247
    p = subprocess.run([tool, "--pairs", str(pairs), "--steps", "2"] + extra, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["gathered_counts_wrong"] == 0 and r["decrypted_bits_wrong"] == 0 and r["gathered_counts_sum"] == pairs
    assert r["fused"] is ("--fused" in extra)
    key = np.array([(i * (n // 16) + 3) % n for i in range(16)], dtype=np.uint64)
    g = np.arange(pairs, dtype=np.uint64)
    pa = ((g * np.uint64(2654435761)) >> np.uint64(13)) & np.uint64(1)
    pb = ((g * np.uint64(40503) + np.uint64(7)) >> np.uint64(5)) & np.uint64(1)
    (ka, na), (kb, nb) = oracle.rng_from_seed(1234), oracle.rng_from_seed(1235)
    a = oracle.encrypt_keyed(n, key, pa.astype(np.uint8), ka, na, 8)
    b = oracle.encrypt_keyed(n, key, pb.astype(np.uint8), kb, nb, 8)
    assert int(r["products_digest"], 16) == oracle.digest(a & b)


@pytest.mark.gpu
def test_tool_all_pairs_shape(tool, oracle):
    """64x64-term products through ShardedBatch::operator*; the digest covers every pair."""
    pairs, t, n, dl = 10, 64, 1247, 20
    p = subprocess.run([tool, "--pairs", str(pairs), "--terms", str(t), "--gpus", "1", "--steps", "2"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["gathered_counts_sum"] == pairs * t * t and r["gathered_counts_wrong"] == 0
    a = oracle.synth(0x43534743 + 1, n, 0, pairs * t * dl)
    b = oracle.synth(0x43534743 + 2, n, 0, pairs * t * dl)
    want = 0
    per = t * t * dl
    for q in range(pairs):
        prod, _ = oracle.mul(n, a[q * t * dl:(q + 1) * t * dl], b[q * t * dl:(q + 1) * t * dl])
        want = (want + oracle.digest(prod, q * per)) & (2**64 - 1)
    assert int(r["products_digest"], 16) == want


@pytest.mark.gpu
def test_tool_exits_non_zero_when_a_rank_fails(tool):
    """VERDICT r2 weak #2 / ADVICE: a failing rank must release its peers and end the program with a
    non-zero exit code and its message -- within seconds, not at a driver time limit."""
    import time
    t0 = time.time()
    p = subprocess.run([tool, "--pairs", "65536", "--steps", "3", "--fail-rank", "0", "--fail-step", "1", "--deadline", "60"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 1, (p.returncode, p.stderr[-1000:])
    assert "injected failure" in p.stderr and "rank 0" in p.stderr
    assert time.time() - t0 < 60


def _world1_comm(shard_lib):
    from csgn_amd import capi
    ident = C.create_string_buffer(capi.CSGN_COMM_ID_BYTES)
    capi.check_shard(shard_lib.csgn_comm_unique_id(ident))
    comm = C.c_void_p()
    capi.check_shard(shard_lib.csgn_comm_init_rank_ex(ident, 0, 1, 0, capi.CSGN_COMM_ALLOW_MINOR_SKEW, C.byref(comm)))
    return comm


@pytest.mark.gpu
def test_native_comm_from_python_world_1(shard_lib):
    """csgn_comm_init_rank_ex + gather + barrier through ctypes (the form bench.py uses per rank)."""
    import torch
    from csgn_amd import capi
    comm = _world1_comm(shard_lib)
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
        capi.check_shard(shard_lib.csgn_comm_check(comm))
    finally:
        shard_lib.csgn_comm_destroy(comm)


@pytest.mark.gpu
def test_init_all_and_destroy_leave_the_callers_device_alone(shard_lib):
    """csgn_comm_init_all_ex walks over the devices and csgn_comm_destroy switches to the communicator's:
    both put the calling thread's current device back (a host thread that owns GPU 0 must not find itself on
    GPU 7 after tearing a group down).  With one GPU on the box the device number cannot differ, so the check
    is on the contract's other half as well: every communicator comes back usable and is destroyed cleanly."""
    import torch
    from csgn_amd import capi
    before = torch.cuda.current_device()
    comms = (C.c_void_p * 1)()
    devs = (C.c_int * 1)(0)
    capi.check_shard(shard_lib.csgn_comm_init_all_ex(1, devs, capi.CSGN_COMM_ALLOW_MINOR_SKEW, comms))
    try:
        assert comms[0] and shard_lib.csgn_comm_device(C.c_void_p(comms[0])) == 0
        assert torch.cuda.current_device() == before
        capi.check_shard(shard_lib.csgn_comm_barrier(C.c_void_p(comms[0]), capi.CSGN_STREAM_OF_COMM))
    finally:
        shard_lib.csgn_comm_destroy(C.c_void_p(comms[0]))
    assert torch.cuda.current_device() == before
    x = torch.arange(10, device="cuda")                        # the thread's context still works
    assert int(x.sum()) == 45


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["explicit", "null", "comm"])
def test_gather_is_ordered_behind_its_producer_on_the_launch_stream(shard_lib, which):
    """the gather must run on the stream it is GIVEN.  NULL is the legacy default
    stream (as in csgn_hip.h), not the communicator's private stream, so producer and gather on the
    same handle are ordered.  Every step refills `local` with step-dependent counts right before
    the gather, behind a long kernel that keeps the stream busy; a gather on any other stream would
    read the previous step's values."""
    import torch
    from csgn_amd import capi
    comm = _world1_comm(shard_lib)
    n = 1 << 20
    try:
        if which == "explicit":
            ts = torch.cuda.Stream()
            handle = ts.cuda_stream
            assert handle != 0
        elif which == "null":
            ts = torch.cuda.default_stream()
            handle = 0
        else:                                        # the communicator's own stream, asked for by name
            own = shard_lib.csgn_comm_stream(comm)
            ts = torch.cuda.ExternalStream(own)
            handle = capi.CSGN_STREAM_OF_COMM
        ballast = torch.empty(1 << 28, dtype=torch.int64, device="cuda")       # 2 GiB: ~0.6 ms per fill
        local = torch.zeros(n, dtype=torch.int64, device="cuda")
        outs = [torch.zeros(n, dtype=torch.int64, device="cuda") for _ in range(6)]
        torch.cuda.synchronize()
        with torch.cuda.stream(ts):
            for step in range(6):
                ballast.fill_(step)                  # keeps the stream busy ahead of the producer
                cnt_ptr = local.data_ptr()
                capi.check_shard(shard_lib.csgn_shard_product_counts(
                    n, None, None, step + 1, step + 3, cnt_ptr, ts.cuda_stream))
                capi.check_shard(shard_lib.csgn_comm_gather_counts(comm, cnt_ptr, n, outs[step].data_ptr(), handle))
        torch.cuda.synchronize()
        for step in range(6):
            assert bool((outs[step] == (step + 1) * (step + 3)).all()), (which, step)
    finally:
        shard_lib.csgn_comm_destroy(comm)


@pytest.mark.gpu
def test_uneven_shard_branch_runs_on_hardware_at_world_1(shard_lib):
    """The grouped-ncclBroadcast form of the gather (what B % G != 0 takes) forced at world 1 through the
    per-communicator option: root 0's broadcast into its slice, out of place from d_local."""
    import torch
    from csgn_amd import capi
    comm = _world1_comm(shard_lib)
    try:
        capi.check_shard(shard_lib.csgn_comm_set_option(comm, capi.CSGN_COMM_OPT_FORCE_GROUPED_BROADCAST, 1))
        assert shard_lib.csgn_comm_set_option(comm, 99, 1) == capi.CSGN_ERR_INVALID
        for n in (1, 7, 1000, 65537):
            local = torch.arange(n, dtype=torch.int64, device="cuda") * 5 + 1
            out = torch.zeros(n, dtype=torch.int64, device="cuda")
            s = torch.cuda.current_stream().cuda_stream
            capi.check_shard(shard_lib.csgn_comm_gather_counts(comm, local.data_ptr(), n, out.data_ptr(), s))
            lb = (torch.arange(n, device="cuda") % 2).to(torch.uint8)
            ob = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
            capi.check_shard(shard_lib.csgn_comm_gather_bytes(comm, lb.data_ptr(), n, ob.data_ptr(), s))
            capi.check_shard(shard_lib.csgn_comm_barrier(comm, s))
            assert torch.equal(out, local) and torch.equal(ob, lb)
    finally:
        shard_lib.csgn_comm_destroy(comm)


@pytest.mark.gpu
def test_abort_releases_the_communicator_and_later_calls_fail_fast(shard_lib):
    """csgn_comm_abort (ncclCommAbort): idempotent, and afterwards gather / barrier / check return an error
    at once instead of enqueueing work on a dead communicator; destroy still frees everything."""
    import torch
    from csgn_amd import capi
    comm = _world1_comm(shard_lib)
    local = torch.ones(16, dtype=torch.int64, device="cuda")
    out = torch.zeros(16, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    capi.check_shard(shard_lib.csgn_comm_set_timeout_ms(comm, 5000))
    capi.check_shard(shard_lib.csgn_comm_gather_counts(comm, local.data_ptr(), 16, out.data_ptr(), s))
    capi.check_shard(shard_lib.csgn_comm_barrier(comm, s))
    assert shard_lib.csgn_comm_abort(comm) == 0 and shard_lib.csgn_comm_abort(comm) == 0
    assert shard_lib.csgn_comm_gather_counts(comm, local.data_ptr(), 16, out.data_ptr(), s) != 0
    assert "aborted" in shard_lib.csgn_shard_last_error().decode()
    assert shard_lib.csgn_comm_barrier(comm, s) != 0
    assert shard_lib.csgn_comm_check(comm) != 0
    assert shard_lib.csgn_comm_destroy(comm) == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_abort_from_another_thread_while_the_owner_is_in_gather_and_barrier(shard_lib):
    """ncclCommAbort frees the communicator, so an abort from a peer's thread must not overtake an
    owner that has passed its `aborted` check and is on its way into RCCL.  The owner thread runs gather +
    barrier back to back (ctypes releases the GIL inside the calls); the main thread aborts in mid-stream.
    Every call either succeeds or fails with "aborted" -- no crash, no hang -- and destroy still works."""
    import threading
    import time
    import torch
    for _ in range(3):
        comm = _world1_comm(shard_lib)
        shard_lib.csgn_comm_set_timeout_ms(comm, 5000)
        local = torch.ones(4096, dtype=torch.int64, device="cuda")
        out = torch.zeros(4096, dtype=torch.int64, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        seen = {"ok": 0, "failed": 0, "messages": set()}
        stop = threading.Event()

        def owner():
            while not stop.is_set():
                for call in (lambda: shard_lib.csgn_comm_gather_counts(comm, local.data_ptr(), 4096, out.data_ptr(), s),
                             lambda: shard_lib.csgn_comm_barrier(comm, s),
                             lambda: shard_lib.csgn_comm_check(comm)):
                    if call() == 0:
                        seen["ok"] += 1
                    else:
                        seen["failed"] += 1
                        seen["messages"].add(shard_lib.csgn_shard_last_error().decode())
                if seen["failed"] > 30:
                    break

        th = threading.Thread(target=owner)
        th.start()
        time.sleep(0.05)
        assert shard_lib.csgn_comm_abort(comm) == 0
        time.sleep(0.02)
        stop.set()
        th.join(timeout=30)
        assert not th.is_alive()
        assert seen["ok"] > 0 and seen["failed"] > 0
        assert all("aborted" in m for m in seen["messages"]), seen["messages"]
        assert shard_lib.csgn_comm_destroy(comm) == 0
        torch.cuda.synchronize()
