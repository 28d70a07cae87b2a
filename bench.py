#!/usr/bin/env python3
"""bench.py -- ciphertext-mults/sec at N=1247 with 1024-term x 1024-term operands.

One "step" = one pass of the hot path over one batch: `batch` independent
Ciphertext x Ciphertext products (1024 x 1024 terms each, Context(1247,16)) computed by
csgn_mul_uniform through the C ABI.  The products (167.8 MB each) do not fit HBM as a
batch, so they are streamed through a fixed output arena of `slots` result buffers
(pair p -> slot p % slots, one launch per `slots` pairs; SURVEY 8d "streaming rule").
Operands are synthetic (seeded splitmix64 words, csgn_synth_fill) and already resident in
HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--slots S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank owns `batch` pairs (weak scaling) and its own arena;
the only exchange is an all-gather (RCCL over xGMI) of the per-pair result term counts at the
end of every step, inside the timed region.  `python bench.py --gpus N` with no WORLD_SIZE in the
environment starts the N ranks itself: the parent makes no GPU call, checks that N devices are
visible (exit 2 otherwise -- never a silent fall back to fewer GPUs) and runs
`python -m torch.distributed.run --nproc-per-node N ... bench.py` as a child process.  Under an
external torchrun, --gpus must equal WORLD_SIZE (exit 2 otherwise).
The gather goes through the native C ABI (csgn_comm_gather_counts of libcsgn_shard.so ->
ncclAllGather, include/csgn_shard.h) on the SAME explicit HIP stream as the multiply and the events
that time it.  The ranks rendezvous over gloo (the ncclUniqueId is 128 bytes; barrier and the
max-over-ranks time are host-side too), so a rank holds ONE RCCL communicator -- the native one --
and no torch NCCL process group; `config.collective` names the RCCL version and file the process
bound (a torch process has torch/lib/librccl.so mapped already: a minor-version difference from the
header libcsgn_shard.so was built against is accepted with CSGN_COMM_ALLOW_MINOR_SKEW and reported
there; a major one is refused).  Set-up failures are handled by agreement, never rank by rank: if the
native library cannot be loaded on every rank the job exits 1; if the communicator cannot be formed
on every rank, ALL ranks (a MIN all-reduce over gloo decides) drop it together and form
torch.distributed's RCCL group for the same gather, and `config.collective` says FALLBACK.  Once the
timed loop runs, a rank that fails aborts its communicator and exits non-zero (torchrun then ends the
others).  `--collective torch` is the opt-in alternative (torch.distributed's own RCCL process group).

Rank 0 prints ONE JSON line (contract in the task description), including
  "roofline":     algorithmic bytes per launch / measured launch duration vs the 8 TB/s HBM peak
  "cpu_baseline": the genuine reference's Ciphertext::operator* (oracle/_ref, kind "reference")
                  or our scalar port of it (kind "port") timed on one host core, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_BITS = 1247
D_KEY = 16
HBM_PEAK_BPS = 8.0e12            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 0x43534743


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="pairs per GPU per step")
    ap.add_argument("--slots", type=int, default=128, help="result buffers in the output arena")
    ap.add_argument("--terms", type=int, default=1024, help="terms per operand")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", dest="cpu_all_cores", action="store_true", default=True,
                    help="also time the CPU baseline batch-parallel on the host's core share, at most 16 "
                         "(extra 'cpu_baseline_all_cores' object; SURVEY 8d, BASELINE.md section 4 mode 2); default")
    ap.add_argument("--no-cpu-all-cores", dest="cpu_all_cores", action="store_false")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the suite of secondary operations that is timed after (and off) the headline's region")
    ap.add_argument("--secondary-seconds", type=float, default=25.0, help="wall-clock budget of that suite")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL and run the term-count all-gather even with one rank "
                         "(exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--collective", choices=["native", "torch"], default="native",
                    help="native: csgn_comm_gather_counts (libcsgn_shard.so, ncclAllGather called directly); "
                         "torch: torch.distributed.all_gather_into_tensor")
    ap.add_argument("--native-ranks", action="store_true",
                    help="run the measurement in tools/bin/bench_native instead of torch rank processes: ONE process, "
                         "one host thread per GPU over the C ABI, one HIP runtime and the RCCL libcsgn_shard.so was built "
                         "against (strict version check).  Same workload, same JSON line; the CPU baseline is added here.")
    ap.add_argument("--spawn", action="store_true",
                    help="start the rank processes through torch.distributed.run even for --gpus 1")
    ap.add_argument("--verify-slots", type=int, default=8, help="arena slots compared with the oracle after timing")
    ap.add_argument("--dev-ranks-share-gpu", action="store_true",
                    help="DEVELOPMENT ONLY: every rank uses GPU 0 and the process group runs on gloo, so that the "
                         "world > 1 logic (shards, offsets, verification, max-over-ranks) can be rehearsed on a "
                         "one-GPU box; RCCL refuses two ranks on one device, so no RCCL number comes out of it")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """Parent of an N-rank run: no GPU call is made here (torch.cuda.device_count() does not
    initialise the device on this image); the ranks are fresh child processes."""
    import socket
    import subprocess
    import torch
    visible = torch.cuda.device_count()
    if args.dev_ranks_share_gpu and visible >= 1:
        visible = args.gpus
    if visible < args.gpus:
        print(f"bench.py: --gpus {args.gpus} requested but {visible} device(s) visible; refusing to run "
              f"on fewer GPUs than asked", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print("# bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def native_ranks(args) -> int:
    """--native-ranks: the torch-free driver does the GPU work; this process makes no GPU call at all."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "bin", "bench_native")
    if not os.path.exists(tool):
        from csgn_amd import build
        build.build_bench_native(verbose=False)
    cmd = [tool, "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch),
           "--slots", str(args.slots), "--terms", str(args.terms), "--force-collective", "1" if args.force_collective else "0"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("# bench.py: " + " ".join(cmd), file=sys.stderr)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    if p.returncode != 0:
        return p.returncode
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]     # RCCL may print a banner before it
    out = json.loads(line)
    if args.gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.terms, args.cpu_seconds)
    print(json.dumps(out))
    return 0


def cpu_baseline(terms: int, budget_s: float):
    """Time the reference's own Ciphertext::operator* (or our port of it) on ONE host core
    for a bounded sample of the same workload shape."""
    import numpy as np
    from oracle import binding
    orc = binding.Oracle()
    dl = orc.default_len(N_BITS)
    a = orc.synth(SEED + 1, N_BITS, 0, terms * dl)
    b = orc.synth(SEED + 2, N_BITS, 0, terms * dl)
    ref = binding.load_ref()
    if ref is not None:
        kind = "reference"
        run = lambda iters: ref.time_mul(N_BITS, D_KEY, a, b, iters)
    else:
        kind = "port"

        def run(iters):
            t0 = time.perf_counter()
            for _ in range(iters):
                orc.mul_reference_cost(N_BITS, a, b)
            return time.perf_counter() - t0
    t1 = run(1)                                   # also warms the allocator
    if t1 < 1e-3:                                 # tiny shapes (--terms 1): calibrate on a longer loop
        t1 = run(2000) / 2000
        budget_s = min(budget_s, 3.0)
    iters = max(1, min(64 if t1 > 1e-2 else 50_000_000, int(budget_s / max(t1, 1e-9))))
    t = run(iters)
    return {
        "value": iters / t,
        "unit": "mult/s",
        "cores": 1,
        "kind": kind,
        "sample": f"{iters} x ({terms}x{terms}-term c*c, N={N_BITS}) in {t:.2f}s, single thread "
                  f"({'Ciphertext::operator* of oracle/_ref' if kind == 'reference' else 'oracle mul_reference_cost'})",
    }


def _cpu_worker(args):
    terms, iters = args
    from oracle import binding
    orc = binding.Oracle()
    dl = orc.default_len(N_BITS)
    a = orc.synth(SEED + 1, N_BITS, 0, terms * dl)
    b = orc.synth(SEED + 2, N_BITS, 0, terms * dl)
    ref = binding.load_ref()
    t0 = time.perf_counter()
    if ref is not None:
        ref.time_mul(N_BITS, D_KEY, a, b, iters)
    else:
        for _ in range(iters):
            orc.mul_reference_cost(N_BITS, a, b)
    return time.perf_counter() - t0


def cpu_baseline_all_cores(terms: int, per_core_rate: float, budget_s: float):
    """The same single-threaded product, one independent stream of pairs per host core
    (the reference has no threading of its own; this is the batch-parallel upper bound).
    Workers are plain child processes of this script (`--cpu-worker`), no GPU, no torch."""
    import subprocess
    from oracle import binding
    # a one-GPU box grants a 16-core CPU share whatever sched_getaffinity reports (256 on the pool
    # hosts); every worker also holds ~0.7 GB of product buffers, so the pool is capped
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("CSGN_BENCH_CPU_CORES", "16")))
    iters = max(1, int(per_core_rate * budget_s * 0.25))     # memory-bound: expect < linear scaling
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(terms), str(iters)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(cores)]
    ok = True
    deadline = t0 + 120.0                    # a bounded extra: never let it hold the bench line back
    for p in procs:
        try:
            ok = (p.wait(timeout=max(1.0, deadline - time.perf_counter())) == 0) and ok
        except subprocess.TimeoutExpired:
            p.kill()                         # this exact child
            p.wait()
            ok = False
    wall = time.perf_counter() - t0
    return {
        "value": cores * iters / wall if ok else None,
        "unit": "mult/s",
        "cores": cores,
        "kind": "reference" if binding.load_ref() is not None else "port",
        "sample": f"{cores} processes x {iters} x ({terms}x{terms}-term c*c) in {wall:.2f}s wall "
                  f"(process start-up included)",
    }


def hbm_clock_info(device: int):
    """What the box says about its memory clock, next to the 8.0 TB/s spec constant the roofline uses
    (BASELINE.md section 3).  Best effort: torch's device properties, then the amdgpu sysfs DPM table."""
    info = {"spec_peak_GBps": HBM_PEAK_BPS / 1e9}
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        clk = getattr(p, "memory_clock_rate", None)          # kHz
        bus = getattr(p, "memory_bus_width", None)           # bits
        if clk and bus:
            info["memory_clock_MHz"] = clk / 1e3
            info["memory_bus_bits"] = bus
            # HBM3E moves 4 bits per pin per reported clock on this part (2 x DDR): 8192 pins x 2 GHz x 4 = 8.2 TB/s
            info["derived_peak_GBps"] = clk * 1e3 * bus / 8 * 4 / 1e9
            info["derived_formula"] = "memory_clock x 4 transfers x bus_bits / 8"
    except Exception as e:                                    # never let a diagnostic cost the contract line
        info["torch_error"] = repr(e)
    try:
        import glob
        import re
        active = set()
        for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk")):
            with open(path) as f:
                for ln in f:
                    m = re.match(r"\s*\d+:\s*(\d+)\s*[Mm][Hh]z\s*\*", ln)
                    if m:
                        active.add(int(m.group(1)))
        if active:
            info["pp_dpm_mclk_active_MHz"] = sorted(active)      # the level every card of the node holds
            if "derived_peak_GBps" not in info:
                # 8 HBM3E stacks x 1024 pins (spec), 4 transfers per pin per reported clock: 2000 MHz -> 8.19 TB/s
                info["memory_bus_bits"] = 8192
                info["derived_peak_GBps"] = max(active) * 1e6 * 4 * 8192 / 8 / 1e9
                info["derived_formula"] = "pp_dpm_mclk x 4 transfers x 8192 pins (spec bus width) / 8"
    except Exception as e:
        info["sysfs_error"] = repr(e)
    return info


def kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) of the multiply kernels' source: ties a PMC capture to a build."""
    import hashlib
    h = hashlib.sha256()
    for f in ("csgn_mul.hip", "csgn_device.h", "csgn_common.h"):
        with open(os.path.join(ROOT, "csgn_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel: str, terms: int, pairs_per_launch: float):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile
    (profiles/traffic_current.json, produced by tools/prof_pmc.sh + tools/pmc_summary.py: separate
    rocprofv3 --pmc passes for WRITE_SIZE and FETCH_SIZE).  Counters cannot be read from inside the
    process being timed, so this is a recorded measurement; it is returned only when it was taken on
    this exact launch shape AND on the kernel source this run was built from, otherwise None."""
    path = os.path.join(ROOT, "profiles", "traffic_current.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, None
    if (t.get("kernel") == kernel and t.get("n_bits") == N_BITS and t.get("terms") == terms
            and t.get("pairs_per_launch") == pairs_per_launch
            and t.get("kernel_source_sha16") == kernel_source_hash()):
        return t.get("hbm_bytes_per_launch"), t.get("captured", "profiles/traffic_current.json")
    return None, None


def secondary_suite(hip, budget_s: float = 25.0):
    """The other operations of the hot path on the driver's clock (the reference times every
    operation, tests/timings.cpp:17-66): after the headline's timed region and OFF it, each case is timed with HIP
    events on the current stream (>= 10 ms of back-to-back warm-up, then the median of five brackets of >= 2 ms),
    inputs rotating through more than the 256 MB memory-side cache where they are smaller, and CHECKED against the
    oracle on sampled elements (words, or bits under a short key so that parities are not trivially zero).
    Returns [{"name", "workload", "bytes" (algorithmic), "ms", "frac" (of 8.0 TB/s), "verified"}, ...];
    a case that raises is reported with "error" instead of a time, and cases past the wall-clock budget are
    "skipped" -- the contract line never depends on this suite."""
    import ctypes as C
    import statistics
    import numpy as np
    import torch
    from csgn_amd.capi import check
    from oracle.binding import Oracle
    orc = Oracle()
    lib = hip.lib
    t_begin = time.perf_counter()
    rows = []

    def timed(fn):
        def bracket(k):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(k):
                fn()
            b.record()
            b.synchronize()
            return a.elapsed_time(b) / 1e3 / k
        est = bracket(1)
        spent = est
        while spent < 10e-3:
            k = max(1, min(64, int(3e-3 / max(est, 1e-6))))
            est = bracket(k)
            spent += est * k
        k = max(1, min(64, int(2e-3 / max(est, 1e-6)) + 1))
        return statistics.median([bracket(k) for _ in range(5)])

    def rotate(make, in_bytes):
        nsets = max(1, min(8, -(-int(300e6) // max(1, in_bytes)))) if in_bytes < 300e6 else 1
        sets = [make(k) for k in range(nsets)]
        turn = [0]

        def nxt():
            v = sets[turn[0] % nsets]
            turn[0] += 1
            return v
        return sets, nxt

    def case(name, workload, body):
        if time.perf_counter() - t_begin > budget_s:
            rows.append({"name": name, "workload": workload, "skipped": "wall-clock budget of the suite spent"})
            return
        try:
            alg, secs, ok = body()
            rows.append({"name": name, "workload": workload, "bytes": int(alg), "ms": secs * 1e3,
                         "frac": alg / secs / HBM_PEAK_BPS, "verified": bool(ok)})
        except Exception as e:                      # never let a secondary figure cost the contract line
            rows.append({"name": name, "workload": workload, "error": repr(e)[:300]})
        torch.cuda.empty_cache()

    n, d, dl = N_BITS, D_KEY, hip.default_len(N_BITS)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask, dkey = hip.upload(hip.key_mask(n, key)), hip.upload(key)
    key2 = np.random.default_rng(2).permutation(n)[:2].astype(np.uint64)      # short key: one synthetic term in 4 hits
    dmask2 = hip.upload(hip.key_mask(n, key2))
    el = lambda t, i, per: hip.download(t[i * per:(i + 1) * per])

    def config5(nb, dk, flags):
        def body():
            B, levels = 4096, 16
            dlc = hip.default_len(nb)
            k5 = np.random.default_rng(5).permutation(nb)[:dk].astype(np.uint64)
            m5, dk5 = hip.upload(hip.key_mask(nb, k5)), hip.upload(k5)
            c = C.c_void_p()
            check(lib.csgn_circuit_create(nb, B, C.byref(c)))
            try:
                def new(fn, *a):
                    v = C.c_uint32()
                    check(fn(c, *a, C.byref(v)))
                    return v.value
                nin = 1 + levels // 2 + 2 * (levels // 2)
                ids = [new(lib.csgn_circuit_input, 1) for _ in range(nin)]
                x, k = ids[0], 1
                for level in range(1, levels + 1):
                    if level % 2:
                        x = new(lib.csgn_circuit_add, x, ids[k]); k += 1
                    else:
                        x = new(lib.csgn_circuit_mul, x, new(lib.csgn_circuit_add, ids[k], ids[k + 1])); k += 2
                bid = new(lib.csgn_circuit_decrypt, x, m5.data_ptr())
                if flags:
                    check(lib.csgn_circuit_optimize(c, flags))
                check(lib.csgn_circuit_build(c))
                st = (C.c_uint64 * 8)()
                check(lib.csgn_circuit_stats(c, st))
                plain = np.random.default_rng(6).integers(0, 2, size=(nin, B)).astype(np.uint8)
                fresh = hip.encrypt_device_rng(nb, dk, hip.upload(plain.reshape(-1)), dk5, m5, seed=9)
                for i in range(nin):
                    check(lib.csgn_memcpy_d2d(lib.csgn_circuit_value(c, ids[i]), fresh[i * B * dlc:].data_ptr(), B * dlc * 8, hip.stream))
                secs = timed(lambda: check(lib.csgn_circuit_run(c, hip.stream)))
                gb = torch.empty(B, dtype=torch.uint8, device=hip.device)
                check(lib.csgn_memcpy_d2d(gb.data_ptr(), lib.csgn_circuit_bits(c, bid), B, hip.stream))
                xb, k = plain[0].copy(), 1
                for level in range(1, levels + 1):
                    if level % 2:
                        xb ^= plain[k]; k += 1
                    else:
                        xb &= plain[k] ^ plain[k + 1]; k += 2
                ok = np.array_equal(hip.download(gb), xb)
                # element 0 through the oracle: the circuit's words end to end, decrypted by the oracle
                hf = hip.download(fresh).reshape(nin, B, dlc)
                h, k = hf[0, 0], 1
                for level in range(1, levels + 1):
                    if level % 2:
                        h, _ = orc.add(h, hf[k, 0]); k += 1
                    else:
                        h, _ = orc.mul(nb, h, orc.add(hf[k, 0], hf[k + 1, 0])[0]); k += 2
                ok = ok and orc.decrypt_canonical(nb, k5, h) == int(xb[0])
                return int(st[1]), secs, ok
            finally:
                lib.csgn_circuit_destroy(c)
        return body
    c5 = lambda nb, dk: (f"BASELINE config 5: Context({nb},{dk}), depth-16 add/multiply circuit (766 terms) + decrypt, batch 4096, "
                         f"one hipGraph launch")
    for nb, dk in ((4096, 32), (N_BITS, D_KEY)):
        case(f"config5_graph_compiled_n{nb}", c5(nb, dk) + "; COMPILED (csgn_circuit_optimize: products written into the sums that "
             "consume them, input copies in one prologue launch, last product fused into the decrypt); bytes = the emitted kernels' algorithmic bytes", config5(nb, dk, 23))

    def mul_1x1():
        B = 1 << 20
        out = hip.empty_words(B * dl)
        sets, nxt = rotate(lambda k: (hip.synth_fill(11 + 2 * k, n, 0, B * dl), hip.synth_fill(12 + 2 * k, n, 0, B * dl)), 2 * B * dl * 8)
        run = lambda: (lambda lr: hip.mul_uniform(n, B, 1, 1, lr[0], lr[1], out=out))(nxt())
        secs = timed(run)
        l, r = sets[0]
        hip.mul_uniform(n, B, 1, 1, l, r, out=out)
        ok = all(np.array_equal(el(out, i, dl), orc.mul(n, el(l, i, dl), el(r, i, dl))[0]) for i in (0, 12345, B - 1))
        return B * 8 * dl * 3, secs, ok
    case("mul_1x1", f"Ciphertext*Ciphertext 1x1 terms, batch 2^20, N={n} (BASELINE configs 2/4 kernel: k_and_stream)", mul_1x1)

    def add_1024():
        B, T = 1024, 1024
        out = hip.empty_words(B * 2 * T * dl)
        sets, nxt = rotate(lambda k: (hip.synth_fill(21 + 2 * k, n, 0, B * T * dl), hip.synth_fill(22 + 2 * k, n, 0, B * T * dl)), 2 * B * T * dl * 8)
        def run():
            l, r = nxt()
            check(lib.csgn_add_uniform(n, B, T, T, l.data_ptr(), r.data_ptr(), out.data_ptr(), hip.stream))
        secs = timed(run)
        l, r = sets[0]
        check(lib.csgn_add_uniform(n, B, T, T, l.data_ptr(), r.data_ptr(), out.data_ptr(), hip.stream))
        ok = all(np.array_equal(el(out, i, 2 * T * dl), orc.add(el(l, i, T * dl), el(r, i, T * dl))[0]) for i in (0, B - 1))
        return B * 2 * 8 * dl * 2 * T, secs, ok
    case("add_1024", f"Ciphertext+Ciphertext 1024+1024 terms, batch 1024, N={n}", add_1024)

    def dec_1024():
        B, T = 4096, 1024
        W = hip.synth_fill(31, n, 0, B * T * dl)                                   # 671 MB: larger than the cache
        bits = torch.empty(B, dtype=torch.uint8, device=hip.device)
        scratch = torch.empty(int(lib.csgn_decrypt_scratch_bytes(B, B * T)), dtype=torch.uint8, device=hip.device)
        run = lambda: check(lib.csgn_decrypt_uniform(n, B, T, W.data_ptr(), dmask2.data_ptr(), bits.data_ptr(),
                                                     scratch.data_ptr(), hip.stream))
        secs = timed(run)
        got = hip.download(bits)
        ok = all(int(got[i]) == orc.decrypt_canonical(n, key2, el(W, i, T * dl)) for i in (0, 1, 2047, B - 1)) and 0 < got.sum() < B
        return B * T * 8 * dl, secs, ok
    case("decrypt_1024", f"SecretKey::decrypt of 1024-term ciphertexts, batch 4096, N={n}", dec_1024)

    def permute_1m():
        B = 1 << 20
        perm = np.random.default_rng(3).permutation(n)
        dperm = hip.upload(perm.astype(np.uint32))
        out = hip.empty_words(B * dl)
        sets, nxt = rotate(lambda k: hip.synth_fill(41 + k, n, 0, B * dl), B * dl * 8)
        run = lambda: check(lib.csgn_permute_uniform(n, B, 1, 0, nxt().data_ptr(), dperm.data_ptr(), out.data_ptr(), hip.stream))
        secs = timed(run)
        check(lib.csgn_permute_uniform(n, B, 1, 0, sets[0].data_ptr(), dperm.data_ptr(), out.data_ptr(), hip.stream))
        ok = all(np.array_equal(el(out, i, dl), orc.permute_ciphertext(n, perm.astype(np.uint64), el(sets[0], i, dl))) for i in (0, 777, B - 1))
        return B * 2 * 8 * dl, secs, ok
    case("permute_1m", f"Ciphertext::applyPermutation, batch 2^20 single-term ciphertexts, N={n}", permute_1m)

    def encrypt_1m():
        B = 1 << 20
        hplain = np.random.default_rng(4).integers(0, 2, B).astype(np.uint8)
        plain = hip.upload(hplain)
        fresh = hip.empty_words(B * dl)
        rng = hip.rng_from_seed(7, 8)
        run = lambda: hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=fresh)
        secs = timed(run)
        rk = np.array(list(rng.key), dtype=np.uint32)
        want = orc.encrypt_keyed(n, key, hplain[:64], rk, int(rng.nonce), 8)
        ok = np.array_equal(el(fresh, 0, 64 * dl), want)
        ok = ok and np.array_equal(hip.download(hip.decrypt_uniform(n, B, 1, fresh, dmask)), hplain)
        return B * 8 * dl, secs, ok
    case("encrypt_keyed_1m", f"SecretKey::encrypt by the keyed generator (ChaCha8), batch 2^20, N={n}: write-only, VALU-bound", encrypt_1m)

    def ragged(mean, count, cap, seed):
        def body():
            rng = np.random.default_rng(seed)
            lg = lambda: np.clip(rng.lognormal(np.log(mean) - 0.5, 1, count), 1, cap).astype(np.int64)
            t1, t2 = lg(), lg()
            offL, offR = np.zeros(count + 1, np.uint64), np.zeros(count + 1, np.uint64)
            offL[1:], offR[1:] = np.cumsum(t1), np.cumsum(t2)
            tot = int(np.sum(t1 * t2))
            alg = 8 * dl * (int(offL[-1]) + int(offR[-1]) + tot)
            dOL, dOR = hip.upload(offL), hip.upload(offR)
            sets, nxt = rotate(lambda k: (hip.synth_fill(51 + 2 * k, n, 0, int(offL[-1]) * dl),
                                          hip.synth_fill(52 + 2 * k, n, 0, int(offR[-1]) * dl)), 8 * dl * int(offL[-1] + offR[-1]))
            out, off_out = hip.empty_words(tot * dl), hip.empty_words(count + 1)
            handle, hplan = hip.mul_plan(), (C.c_uint64 * 4)()
            check(lib.csgn_mul_plan_ragged(handle, count, dOL.data_ptr(), dOR.data_ptr(), off_out.data_ptr(), C.byref(hplan), hip.stream))
            check(lib.csgn_mul_plan_trust(handle, 1))
            def run():
                l, r = nxt()
                check(lib.csgn_mul_planned(handle, n, l.data_ptr(), r.data_ptr(), out.data_ptr(), hip.stream))
            secs = timed(run)
            aplan = hip.empty_words(int(lib.csgn_mul_ragged_async_plan_words(count)))
            def run_async():
                l, r = nxt()
                hip.mul_ragged_async(n, l, dOL, r, dOR, tot, out=out, off_out=off_out, plan=aplan)
            secs_async = timed(run_async)
            l, r = sets[0]
            out.zero_()
            hip.mul_ragged_async(n, l, dOL, r, dOR, tot, out=out, off_out=off_out, plan=aplan)
            oo = hip.download(off_out)
            ok = hip.mul_ragged_async_result(aplan)[4] == 0 and int(oo[-1]) == tot
            big = int(np.argmax(t1 * t2))
            for i in (0, 1, big, count - 1):
                a = hip.download(l[int(offL[i]) * dl:int(offL[i + 1]) * dl])
                b = hip.download(r[int(offR[i]) * dl:int(offR[i + 1]) * dl])
                ok = ok and np.array_equal(hip.download(out[int(oo[i]) * dl:int(oo[i + 1]) * dl]), orc.mul(n, a, b)[0])
            lib.csgn_mul_plan_destroy(handle)
            body.extra = (alg, secs_async, ok)
            return alg, secs, ok
        return body
    for mean, count, cap in ((8, 1 << 18, 600), (16, 1 << 16, 1000)):
        b = ragged(mean, count, cap, mean)
        wl = f"ragged Ciphertext*Ciphertext, {count} pairs, log-normal term counts of mean ~{mean} (long tail to {cap}), N={n}"
        case(f"mul_ragged_mean{mean}_kernel", wl + ": multiply by a plan made once (csgn_mul_planned), cold operands", b)
        if hasattr(b, "extra"):
            alg, secs, ok = b.extra
            case(f"mul_ragged_mean{mean}_async", wl + ": csgn_mul_ragged_async (device-side plan + multiply, no host round trip)",
                 lambda: (alg, secs, ok))

    def ragged_singles(kind, bounded):
        """1 M single-term ciphertexts handed over as CSR: the ragged add (1 + 1 terms) / decrypt entry points, with and
        without the caller's bounds (csgn_*_ragged_bounded: bounds met with equality run the uniform kernels)"""
        def body():
            B = 1 << 20
            off = hip.upload(np.arange(B + 1, dtype=np.uint64))
            if kind == "add":
                sets, nxt = rotate(lambda k: (hip.synth_fill(71 + 2 * k, n, 0, B * dl), hip.synth_fill(72 + 2 * k, n, 0, B * dl)), 2 * B * dl * 8)
                out, off_out = hip.empty_words(2 * B * dl), hip.empty_words(B + 1)
                mx = 1 if bounded else 0
                def run():
                    l, r = nxt()
                    check(lib.csgn_add_ragged_bounded(n, B, mx, mx, l.data_ptr(), off.data_ptr(), r.data_ptr(), off.data_ptr(),
                                                      out.data_ptr(), off_out.data_ptr(), 2 * B, hip.stream))
                secs = timed(run)
                l, r = sets[0]
                check(lib.csgn_add_ragged_bounded(n, B, mx, mx, l.data_ptr(), off.data_ptr(), r.data_ptr(), off.data_ptr(),
                                                  out.data_ptr(), off_out.data_ptr(), 2 * B, hip.stream))
                oo = hip.download(off_out)
                ok = np.array_equal(oo, 2 * np.arange(B + 1, dtype=np.uint64))
                ok = ok and all(np.array_equal(el(out, i, 2 * dl), orc.add(el(l, i, dl), el(r, i, dl))[0]) for i in (0, 4242, B - 1))
                return 2 * 8 * dl * 2 * B, secs, ok
            sets, nxt = rotate(lambda k: hip.synth_fill(81 + k, n, 0, B * dl), B * dl * 8)
            bits = torch.empty(B, dtype=torch.uint8, device=hip.device)
            scratch = torch.empty(int(lib.csgn_decrypt_scratch_bytes(B, B)), dtype=torch.uint8, device=hip.device)
            run = lambda: check(lib.csgn_decrypt_ragged_bounded(n, B, B, 1 if bounded else 0, nxt().data_ptr(), off.data_ptr(),
                                                                dmask2.data_ptr(), bits.data_ptr(), scratch.data_ptr(), hip.stream))
            secs = timed(run)
            check(lib.csgn_decrypt_ragged_bounded(n, B, B, 1 if bounded else 0, sets[0].data_ptr(), off.data_ptr(), dmask2.data_ptr(),
                                                  bits.data_ptr(), scratch.data_ptr(), hip.stream))
            got = hip.download(bits)
            ok = all(int(got[i]) == orc.decrypt_canonical(n, key2, el(sets[0], i, dl)) for i in (0, 1, 77777, B - 1)) and 0 < got.sum() < B
            return B * 8 * dl, secs, ok
        return body
    for kind, what in (("add", "Ciphertext+Ciphertext, 2^20 sums of 1 + 1 terms given as CSR"), ("decrypt", "SecretKey::decrypt of 2^20 single-term ciphertexts given as CSR")):
        case(f"{kind}_ragged_singles", f"{what}, N={n}: csgn_{kind}_ragged (shapes unknown to the dispatch: the CSR kernels)", ragged_singles(kind, False))
        case(f"{kind}_ragged_singles_bounded", f"{what}, N={n}: csgn_{kind}_ragged_bounded with the caller's bound (met with equality: the uniform kernels)",
             ragged_singles(kind, True))

    def compact(frac):
        def body():
            B, T = 4096, 1024
            w = hip.synth_fill(61, n, 0, B * T * dl).view(B, T, dl)
            distinct = max(1, int(round(T * (1.0 - frac))))
            if distinct < T:
                g = torch.Generator(device=hip.device)
                g.manual_seed(61)
                src = torch.randint(0, distinct, (B, T - distinct), device=hip.device, generator=g)
                w[:, distinct:, :] = torch.gather(w[:, :distinct, :], 1, src.unsqueeze(-1).expand(-1, -1, dl))
            w = w.reshape(-1)
            off = torch.arange(0, (B + 1) * T, T, dtype=torch.int64, device=hip.device)
            out, off_out = hip.empty_words(B * T * dl), hip.empty_words(B + 1)
            scratch = torch.empty(int(lib.csgn_compact_scratch_bytes(n, B, B * T)), dtype=torch.uint8, device=hip.device)
            run = lambda: hip.compact_ragged(n, w, off, total_terms=B * T, max_terms=T, out=out, off_out=off_out,
                                             scratch=scratch, sync=False)
            secs = timed(run)
            oo = hip.download(off_out)
            kept = int(oo[-1])
            ok = all(np.array_equal(hip.download(out[int(oo[i]) * dl:int(oo[i + 1]) * dl]), orc.compact(n, el(w, i, T * dl)))
                     for i in (0, B - 1))
            return 8 * dl * (B * T + kept), secs, ok
        return body
    for frac in (0.0, 0.5):
        case(f"compact_{int(frac * 100)}pct", f"mod-2 compaction (extension), 4096 ciphertexts x 1024 terms, {int(frac * 100)} % duplicate terms, N={n}",
             compact(frac))

    def compact_large():
        # ciphertexts beyond one workgroup's group: pairs dealt to hash partitions, terms read twice
        B, T = 4, 1 << 20
        w = hip.synth_fill(63, n, 0, B * T * dl)
        off = torch.arange(0, (B + 1) * T, T, dtype=torch.int64, device=hip.device)
        out, off_out = hip.empty_words(B * T * dl), hip.empty_words(B + 1)
        scratch = torch.empty(int(lib.csgn_compact_scratch_bytes(n, B, B * T)), dtype=torch.uint8, device=hip.device)
        run = lambda: hip.compact_ragged(n, w, off, total_terms=B * T, max_terms=0, out=out, off_out=off_out,
                                         scratch=scratch, sync=False)
        secs = timed(run)
        kept = int(hip.download(off_out)[-1])
        ok = kept == B * T and bool(torch.equal(out, w))                 # distinct terms: nothing cancels, nothing moves
        # ... and the same path against the oracle where it finishes in a second: 3 x 6000 terms drawn from 3000
        Ts = 6000
        small = hip.synth_fill(64, n, 0, 3 * Ts * dl).view(3, Ts, dl)
        g = torch.Generator(device=hip.device)
        g.manual_seed(64)
        src = torch.randint(0, Ts // 2, (3, Ts // 2), device=hip.device, generator=g)
        small[:, Ts // 2:, :] = torch.gather(small[:, :Ts // 2, :], 1, src.unsqueeze(-1).expand(-1, -1, dl))
        small = small.reshape(-1)
        so, soo = hip.compact_ragged(n, small, torch.arange(0, 4 * Ts, Ts, dtype=torch.int64, device=hip.device), total_terms=3 * Ts)
        soo = hip.download(soo)
        ok = ok and all(np.array_equal(hip.download(so[int(soo[i]) * dl:int(soo[i + 1]) * dl]), orc.compact(n, el(small, i, Ts * dl)))
                        for i in range(3))
        return 8 * dl * (B * T + kept), secs, ok
    case("compact_large", f"mod-2 compaction (extension), 4 ciphertexts x 2^20 distinct terms (hash partitions: terms read twice), N={n}",
         compact_large)

    for nb, dk in ((4096, 32), (N_BITS, D_KEY)):
        case(f"config5_graph_tape_n{nb}", c5(nb, dk) + "; TAPE (every value materialised, one kernel per node)", config5(nb, dk, 0))
    return rows


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-worker":      # child of cpu_baseline_all_cores
        _cpu_worker((int(sys.argv[2]), int(sys.argv[3])))
        return
    args = parse_args()
    in_rank = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.native_ranks and not in_rank:
        sys.exit(native_ranks(args))
    if not in_rank and (args.gpus > 1 or args.spawn):
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}; refusing to run",
                  file=sys.stderr)
        sys.exit(2)
    # dmabuf IPC is the only mode this pool's host driver supports; RCCL's peer mappings between rank
    # processes fail with "hipIpcGetMemHandle: invalid argument" without it.  Set before anything
    # initialises HIP in this process (the spawning parent sets it for its children as well).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when a
    # communicator is created, so everything but the final line goes to stderr: fd 1 is pointed at
    # fd 2 for the run and the line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import ctypes
    import numpy as np
    import torch
    import torch.distributed as dist
    from csgn_amd import capi
    from csgn_amd.batch import HipPath
    from csgn_amd.shard import gather_term_counts, shard_range

    if args.dev_ranks_share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    use_dist = world > 1 or args.force_collective
    native = use_dist and args.collective == "native" and not args.dev_ranks_share_gpu
    if use_dist:
        if not in_rank:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # Native collective: the process group is only the out-of-band channel (128-byte id, host
        # barrier, max of a double), so it runs on gloo and the rank's ONE RCCL communicator is the
        # native one.  --collective torch builds torch's RCCL process group instead.
        if native or args.dev_ranks_share_gpu:
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: never resolve the hostname
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    host_group = native or args.dev_ranks_share_gpu          # collectives of `dist` take host tensors
    n_gpus = world

    hip = HipPath(local_rank)
    dev = hip.device
    # Everything of a step -- multiply, term counts, the gather, the events -- goes on ONE explicit
    # stream (a non-zero handle: NULL would mean the legacy default stream to the C ABI).
    run_stream = torch.cuda.Stream(device=dev)
    T = args.terms
    dl = hip.default_len(N_BITS)
    batch = args.batch
    slots = min(args.slots, batch)
    words_per_operand = T * dl
    words_per_product = T * T * dl
    bytes_per_mul = 8 * dl * (T + T + T * T)              # B_mul, SURVEY 8d
    launches_per_step = (batch + slots - 1) // slots

    # ---- operands resident in HBM.  The global batch is world*batch pairs; this rank owns
    # the contiguous range [lo, hi) and its words are a function of the GLOBAL pair index, so
    # per-pair results do not depend on the GPU count. ----
    total_pairs = world * batch
    lo, hi = shard_range(total_pairs, rank, world)
    assert hi - lo == batch
    left = hip.synth_fill(SEED + 1, N_BITS, lo * words_per_operand, batch * words_per_operand)
    right = hip.synth_fill(SEED + 2, N_BITS, lo * words_per_operand, batch * words_per_operand)
    arena = hip.empty_words(slots * words_per_product)
    counts = hip.empty_words(batch)                       # result term counts of this shard
    gathered = torch.empty((total_pairs,), dtype=torch.int64, device=dev) if use_dist else None

    # ---- the one exchange: all-gather of per-pair result term counts.  Native = the C ABI's own
    # RCCL communicator (libcsgn_shard.so).  Set-up is phased so that the ranks cannot drift apart:
    # (1) local load + id on rank 0, (2) agree that every rank got there, (3) ship the id and join,
    # (4) agree again.  A failure after (3) has begun is fatal for the job (exit 1), never a silent
    # switch to another path on some ranks only. ----
    collective = "none"
    comm = None
    shard_lib = None
    nccl_group = None                                          # only for the agreed fallback

    def all_ok(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], device=(torch.device("cpu") if host_group else dev))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    if use_dist and not native:
        collective = "torch.distributed.all_gather_into_tensor(result term counts) [%s]" % (
            "gloo, DEVELOPMENT rehearsal: ranks share one GPU" if args.dev_ranks_share_gpu else "RCCL, torch process group")
    if native:
        ident, err = [None], None
        try:                                                      # phase 1: local, cannot block
            shard_lib = capi.load_shard_library()
            rt_code, hd_code, rccl_path = capi.rccl_info()
            if rank == 0:
                buf = ctypes.create_string_buffer(capi.CSGN_COMM_ID_BYTES)
                capi.check_shard(shard_lib.csgn_comm_unique_id(buf))
                ident = [bytes(buf.raw)]
        except Exception as e:
            err = e
            print(f"# rank {rank}: native communicator unavailable: {e!r}", file=sys.stderr)
        if not all_ok(err is None):                               # phase 2
            if rank == 0:
                print("bench.py: libcsgn_shard.so / RCCL could not be set up on every rank; refusing to run "
                      "(use --collective torch for torch.distributed's own RCCL group)", file=sys.stderr)
            sys.exit(1)
        dist.broadcast_object_list(ident, src=0)                  # phase 3: from here on failure is fatal
        h = ctypes.c_void_p()
        # a torch process has torch/lib/librccl.so mapped already; binding to that one copy is what
        # keeps ONE RCCL in the process, so a minor-version skew against the build header is accepted
        # and REPORTED (a major skew is refused by the library)
        rc = shard_lib.csgn_comm_init_rank_ex(ident[0], rank, world, local_rank, capi.CSGN_COMM_ALLOW_MINOR_SKEW,
                                              ctypes.byref(h))
        if rc != 0:
            print(f"# rank {rank}: csgn_comm_init_rank_ex failed [{rc}]: "
                  f"{shard_lib.csgn_shard_last_error().decode(errors='replace')}", file=sys.stderr)
        else:
            comm = h
        if not all_ok(comm is not None):                          # phase 4: every rank knows the outcome
            # Some rank could not join.  Every rank agrees on that (the MIN above), so they leave the native
            # path TOGETHER -- no rank is left inside a collective -- and try torch.distributed's own RCCL
            # group for the same gather; if that cannot be formed either the job ends non-zero.
            if comm is not None:
                shard_lib.csgn_comm_abort(comm)
                shard_lib.csgn_comm_destroy(comm)
                comm = None
            if rank == 0:
                print("bench.py: native RCCL communicator could not be formed on every rank; all ranks switch to "
                      "torch.distributed's RCCL group", file=sys.stderr)
            try:
                nccl_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            except Exception as e:
                print(f"# rank {rank}: torch RCCL group failed too: {e!r}", file=sys.stderr)
                sys.exit(1)
            native = False
        if comm is not None:
            collective = ("csgn_comm_gather_counts -> ncclAllGather(result term counts) [RCCL %s from %s, native C ABI "
                          "built against %s; rendezvous over gloo, no torch NCCL group]" % (
                              capi.rccl_version_text(rt_code), rccl_path, capi.rccl_version_text(hd_code)))
        else:
            collective = ("torch.distributed.all_gather_into_tensor(result term counts) [RCCL, torch process group; "
                          "FALLBACK: the native communicator failed on some rank, all ranks agreed to switch]")
    torch.cuda.synchronize()

    def gather():
        if comm is not None:
            assert hip.stream == run_stream.cuda_stream and hip.stream != 0
            capi.check_shard(shard_lib.csgn_comm_gather_counts(comm, counts.data_ptr(), total_pairs,
                                                               gathered.data_ptr(), hip.stream))
        elif args.dev_ranks_share_gpu:
            gathered.copy_(gather_term_counts(counts.cpu(), total_pairs, force=True))
        else:
            gather_term_counts(counts, total_pairs, group=nccl_group, out=gathered, force=True)

    def product_counts():
        # per-pair result term counts of this step's products (newlen/dL, src/Ciphertext.cpp:146)
        if shard_lib is not None:
            capi.check_shard(shard_lib.csgn_shard_product_counts(batch, None, None, T, T, counts.data_ptr(), hip.stream))
        else:
            counts.fill_(T * T)

    def host_barrier():
        if use_dist:
            dist.barrier()

    def step(e_mul=None, e_step=None):
        if e_step:
            e_step[0].record()
        if e_mul:
            e_mul[0].record()
        hip.mul_uniform(N_BITS, batch, T, T, left, right, out=arena, out_slots=slots)
        if e_mul:
            e_mul[1].record()
        if use_dist:
            product_counts()
            gather()
        if e_step:
            e_step[1].record()

    # verification hooks (off the clock except the two mid-run digests above)
    from csgn_amd.capi import check
    mid_step = args.steps // 2
    mid_slots = [0, slots - 1] if slots > 1 else [0]
    mid_digests = None
    if not args.no_verify and rank == 0 and args.steps >= 3:
        mid_digests = torch.zeros(len(mid_slots), dtype=torch.int64, device=dev)
    mk = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev_mul = [mk() for _ in range(args.steps)]
    ev_step = [mk() for _ in range(args.steps)]
    try:
        with torch.cuda.stream(run_stream):                 # hip.stream is now run_stream's handle
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            host_barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step(ev_mul[k], ev_step[k])
                if k == mid_step and mid_digests is not None:
                    # a launch that is NOT the run's last one: digest two arena slots right behind the middle
                    # step (two reads of 168 MB, ~60 us inside a step of more than a second; checked after the loop)
                    for slot in mid_slots:
                        check(hip.lib.csgn_digest(arena[slot * words_per_product:].data_ptr(), words_per_product, 0,
                                                  mid_digests[mid_slots.index(slot):].data_ptr(), hip.stream))
            torch.cuda.synchronize()
            host_barrier()
            elapsed = time.perf_counter() - t0
        if comm is not None:
            capi.check_shard(shard_lib.csgn_comm_check(comm))      # an asynchronous RCCL error is a failed run
    except BaseException as e:
        # a failing rank must take the job down, not leave its peers inside a collective: abort the
        # communicator (releases them) and exit non-zero (torchrun ends the remaining ranks)
        print(f"# rank {rank}: FAILED in the timed loop: {e!r}", file=sys.stderr)
        if comm is not None:
            shard_lib.csgn_comm_abort(comm)
        sys.stderr.flush()
        os._exit(1)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=(torch.device("cpu") if host_group else dev))
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # kernel time from HIP events on the launch stream: each bracket holds launches_per_step
    # back-to-back launches of the all-pairs kernel
    mul_ms = [a.elapsed_time(b) for a, b in ev_mul]
    step_ms = sorted(a.elapsed_time(b) for a, b in ev_step)       # this rank's whole steps (mul + exchange)
    kernel_ms = sum(mul_ms)
    n_launches = launches_per_step * args.steps
    avg_launch_s = kernel_ms / 1e3 / n_launches
    # the kernel(s) the library dispatches this shape to ("k_touch+k_mul_flat": the operand touch
    # pass is inside the timed launch and charged to it)
    pairs_per_launch = batch / launches_per_step
    kernel_name = hip.lib.csgn_mul_uniform_kernel(N_BITS, batch, T, T).decode()
    achieved = pairs_per_launch * bytes_per_mul / avg_launch_s

    # ---- validity: the arena still holds the last `slots` products; compare slots spread over the
    # whole arena (first, last and evenly spaced ones) with the oracle ----
    verified = None
    verified_slots = []
    oracle_slots = []
    verify_notes = {}
    if not args.no_verify and rank == 0:
        from oracle.binding import Oracle
        orc = Oracle()
        ok = True
        first_of_last = (launches_per_step - 1) * slots
        live = batch - first_of_last                           # slots rewritten by the last launch
        pair_of_slot = [first_of_last + sl if sl < live else first_of_last - slots + sl for sl in range(slots)]
        # (a) EVERY slot, on the GPU: Dec(arena slot) under random keys against Dec(L_q) & Dec(R_q) of the pair
        # that wrote the slot last, computed from the OPERANDS (csgn_decrypt_product_uniform never sees the
        # arena).  Short keys (d = 2, 3) so that about one term in 4 / 8 hits and the parities are not trivially 0.
        if min(pair_of_slot) >= 0:
            idx = torch.tensor(pair_of_slot, dtype=torch.int64, device=dev)
            lsel = left.view(batch, words_per_operand)[idx].reshape(-1)
            rsel = right.view(batch, words_per_operand)[idx].reshape(-1)
            krng = np.random.default_rng(SEED ^ 0x5EED)
            mismatches, ones = 0, 0
            for trial in range(6):
                vkey = krng.permutation(N_BITS)[:2 + trial % 2].astype(np.uint64)
                vmask = hip.upload(hip.key_mask(N_BITS, vkey))
                got = hip.decrypt_uniform(N_BITS, slots, T * T, arena, vmask)
                want = hip.decrypt_combined_uniform(N_BITS, slots, T, T, lsel, rsel, vmask, product=True)
                mismatches += int((got != want).sum().item())
                ones += int(want.sum().item())
            ok = ok and mismatches == 0 and ones > 0
            verified_slots = list(range(slots))
            verify_notes["all_slots"] = (f"{slots} slots x 6 random keys (d=2,3): Dec(slot) == Dec(L)&Dec(R) from the operands; "
                                         f"{mismatches} mismatches, {ones} one-bits among {6 * slots}")
            del lsel, rsel
        # (b) two slots of a MID-RUN launch (digests taken behind step `mid_step`) against the oracle
        if mid_digests is not None:
            md = hip.download(mid_digests)
            for j, slot in enumerate(mid_slots):
                p = lo + pair_of_slot[slot]
                a = orc.synth(SEED + 1, N_BITS, p * words_per_operand, words_per_operand)
                b = orc.synth(SEED + 2, N_BITS, p * words_per_operand, words_per_operand)
                want, _ = orc.mul(N_BITS, a, b)
                ok = ok and (int(md[j]) == orc.digest(want))
            verify_notes["mid_run"] = f"slots {mid_slots} digested behind step {mid_step} of {args.steps}, equal to the oracle's products"
        want_n = max(1, min(args.verify_slots, slots))
        picks = sorted({int(round(i * (slots - 1) / max(1, want_n - 1))) for i in range(want_n)} | {0, slots - 1})
        for slot in picks:
            # the pair that wrote this slot last: in the last launch if it reached the slot, else one launch earlier
            p_local = first_of_last + slot if slot < live else first_of_last - slots + slot
            if p_local < 0:
                continue
            p = lo + p_local                                   # global pair index
            a = orc.synth(SEED + 1, N_BITS, p * words_per_operand, words_per_operand)
            b = orc.synth(SEED + 2, N_BITS, p * words_per_operand, words_per_operand)
            want, _ = orc.mul(N_BITS, a, b)
            got = hip.digest(arena[slot * words_per_product:(slot + 1) * words_per_product])
            ok = ok and (got == orc.digest(want))
            oracle_slots.append(slot)
            if slot not in verified_slots:
                verified_slots.append(slot)
        if use_dist:
            ok = ok and bool((gathered == T * T).all().item()) and gathered.numel() == world * batch
        verified = bool(ok)
        if not ok:
            print("# VERIFY FAILED: arena contents differ from the oracle", file=sys.stderr)

    if rank == 0:
        total_mults = world * batch * args.steps
        value = total_mults / elapsed
        traffic, traffic_src = measured_traffic(kernel_name, T, pairs_per_launch)
        out = {
            "metric": f"ciphertext-mults/sec (N={N_BITS}, {T}-term operands)",
            "value": value,
            "unit": "mult/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"Ciphertext*Ciphertext all-pairs AND, Context({N_BITS},{D_KEY}), "
                            f"{T}x{T} terms, batch={batch} pairs/GPU streamed through a "
                            f"{slots}-slot output arena ({slots * words_per_product * 8 / 2**30:.1f} GiB)",
                "n_bits": N_BITS, "terms": T, "batch_per_gpu": batch, "arena_slots": slots,
                "pairs_per_launch": pairs_per_launch, "seed": SEED,
                "bytes_per_mult": bytes_per_mul,
                "collective": collective,
                "verified_vs_oracle": verified,
                "verified_slots": sorted(verified_slots),
                "verified_slots_oracle_digest": oracle_slots,
                "verification": verify_notes,
                "hbm_clock": hbm_clock_info(local_rank),
                "step_ms_rank0": {"median": step_ms[len(step_ms) // 2], "min": step_ms[0], "max": step_ms[-1],
                                  "n": len(step_ms)},
                "value_from_median_step": world * batch / (step_ms[len(step_ms) // 2] / 1e3),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK_BPS / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BPS,
                "traffic": traffic,
                "traffic_unit": "bytes/launch (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes: %s)" % traffic_src
                                if traffic is not None else None,
                "algorithmic_bytes_per_launch": pairs_per_launch * bytes_per_mul,
                "kernel": kernel_name,
                "avg_launch_ms": avg_launch_s * 1e3,
                "launch_ms": {"median": sorted(mul_ms)[len(mul_ms) // 2] / launches_per_step,
                              "min": min(mul_ms) / launches_per_step, "max": max(mul_ms) / launches_per_step},
                "launches": n_launches,
            },
        }
        if world == 1 and not args.no_secondary:
            # the other operations of the path, on this run's clock but OFF the timed region (value, roofline and
            # ms_per_step above are already fixed); the headline's 20 GiB arena and operands are released first
            try:
                del arena, left, right
                torch.cuda.empty_cache()
                with torch.cuda.stream(run_stream):
                    out["secondary"] = secondary_suite(hip, args.secondary_seconds)
            except Exception as e:
                out["secondary"] = [{"error": repr(e)[:300]}]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, args.cpu_seconds)
            if args.cpu_all_cores:
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(
                        T, out["cpu_baseline"]["value"], args.cpu_seconds)
                except Exception as e:       # the extra figure must not cost the contract line
                    out["cpu_baseline_all_cores"] = {"value": None, "error": repr(e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)

    if comm is not None:
        shard_lib.csgn_comm_destroy(comm)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
