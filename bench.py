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
the only exchange is an all_gather (RCCL over xGMI) of the per-pair result term counts at the
end of every step, inside the timed region.

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
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL and run the term-count all-gather even with one rank "
                         "(exercises the N>1 code path on a 1-GPU box)")
    return ap.parse_args()


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


def measured_traffic(kernel: str, terms: int, pairs_per_launch: float):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile
    (profiles/traffic_current.json, produced by tools/prof_pmc.sh + tools/pmc_summary.py).
    Returned only when it was measured on this exact launch shape; otherwise None."""
    path = os.path.join(ROOT, "profiles", "traffic_current.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if (t.get("kernel") == kernel and t.get("n_bits") == N_BITS and t.get("terms") == terms
            and t.get("pairs_per_launch") == pairs_per_launch):
        return t.get("hbm_bytes_per_launch")
    return None


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-worker":      # child of cpu_baseline_all_cores
        _cpu_worker((int(sys.argv[2]), int(sys.argv[3])))
        return
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from csgn_amd.batch import HipPath
    from csgn_amd.shard import gather_term_counts, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_collective
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    hip = HipPath(local_rank)
    dev = hip.device
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
    counts = torch.full((batch,), T * T, dtype=torch.int64, device=dev)   # result term counts
    gathered = torch.empty((total_pairs,), dtype=torch.int64, device=dev) if use_dist else None
    torch.cuda.synchronize()

    def step():
        hip.mul_uniform(N_BITS, batch, T, T, left, right, out=arena, out_slots=slots)
        if use_dist:
            gather_term_counts(counts, total_pairs, out=gathered, force=True)

    for _ in range(args.warmup):
        step()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        hip.mul_uniform(N_BITS, batch, T, T, left, right, out=arena, out_slots=slots)
        ev[k][1].record()
        if use_dist:
            gather_term_counts(counts, total_pairs, out=gathered, force=True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # kernel time from HIP events on the launch stream: each bracket holds launches_per_step
    # back-to-back launches of the all-pairs kernel
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev)
    n_launches = launches_per_step * args.steps
    avg_launch_s = kernel_ms / 1e3 / n_launches
    # the kernel(s) the library dispatches this shape to ("k_touch+k_mul_flat": the operand touch
    # pass is inside the timed launch and charged to it)
    pairs_per_launch = batch / launches_per_step
    kernel_name = hip.lib.csgn_mul_uniform_kernel(N_BITS, batch, T, T).decode()
    achieved = pairs_per_launch * bytes_per_mul / avg_launch_s

    # ---- validity: the arena still holds the last `slots` products; check sampled ones ----
    verified = None
    if not args.no_verify and rank == 0:
        from oracle.binding import Oracle
        orc = Oracle()
        ok = True
        first_of_last = (launches_per_step - 1) * slots
        for slot in sorted({0, (batch - first_of_last) - 1}):
            p = lo + first_of_last + slot                      # global pair index
            a = orc.synth(SEED + 1, N_BITS, p * words_per_operand, words_per_operand)
            b = orc.synth(SEED + 2, N_BITS, p * words_per_operand, words_per_operand)
            want, _ = orc.mul(N_BITS, a, b)
            got = hip.digest(arena[slot * words_per_product:(slot + 1) * words_per_product])
            ok = ok and (got == orc.digest(want))
        if use_dist:
            ok = ok and bool((gathered == T * T).all().item()) and gathered.numel() == world * batch
        verified = bool(ok)
        if not ok:
            print("# VERIFY FAILED: arena contents differ from the oracle", file=sys.stderr)

    if rank == 0:
        total_mults = world * batch * args.steps
        value = total_mults / elapsed
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
                "collective": "all_gather(result term counts)" if use_dist else "none",
                "verified_vs_oracle": verified,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK_BPS / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BPS,
                "traffic": measured_traffic(kernel_name, T, pairs_per_launch),
                "traffic_unit": "bytes/launch (PMC, profiles/traffic_current.json)",
                "algorithmic_bytes_per_launch": pairs_per_launch * bytes_per_mul,
                "kernel": kernel_name,
                "avg_launch_ms": avg_launch_s * 1e3,
                "launches": n_launches,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, args.cpu_seconds)
            if args.cpu_all_cores:
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(
                        T, out["cpu_baseline"]["value"], args.cpu_seconds)
                except Exception as e:       # the extra figure must not cost the contract line
                    out["cpu_baseline_all_cores"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
