#!/usr/bin/env python3
"""The ragged entry points on runs of single terms and on a skewed batch, kernels only, for rocprofv3
--kernel-trace --stats and for wall-clock per call (HIP events): add 1 M 1+1, skewed add (one 1024-term + 65 535
singles a side), decrypt of 1 M single-term ciphertexts, one 1 M-term + 65 535 singles, csgn_mul_ragged_async on the
skewed batch.  Dev tool."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath, check
hip = HipPath(0)
n, dl = 1247, 20
CALLS = int(os.environ.get("CALLS", "20"))
def csr(c):
    o = np.zeros(len(c) + 1, dtype=np.uint64); o[1:] = np.cumsum(np.asarray(c, dtype=np.uint64)); return o
def timed(fn):
    def bracket(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k): fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / 1e3 / k
    est = bracket(1); spent = est
    while spent < 30e-3:
        k = max(1, min(64, int(5e-3 / max(est, 1e-6)))); est = bracket(k); spent += est * k
    k = max(1, min(64, int(2e-3 / max(est, 1e-6)) + 1))
    return statistics.median([bracket(k) for _ in range(9)])
key = np.random.default_rng(1).permutation(n)[:16].astype(np.uint64)
dmask = hip.upload(hip.key_mask(n, key))
for name, t1s, t2s in [("1M 1+1", [1] * (1 << 20), [1] * (1 << 20)), ("skewed 1024+65535x1", [1024] + [1] * 65535, [1024] + [1] * 65535)]:
    offL, offR = csr(t1s), csr(t2s)
    sets = [(hip.synth_fill(1 + 2 * k, n, 0, int(offL[-1]) * dl), hip.synth_fill(2 + 2 * k, n, 0, int(offR[-1]) * dl)) for k in range(3)]
    dOL, dOR = hip.upload(offL), hip.upload(offR)
    tot = int(offL[-1] + offR[-1])
    aout, aoff = hip.empty_words(tot * dl), hip.empty_words(len(t1s) + 1)
    turn = [0]
    def add_only():
        l, r = sets[turn[0] % 3]; turn[0] += 1
        check(hip.lib.csgn_add_ragged(n, len(t1s), l.data_ptr(), dOL.data_ptr(), r.data_ptr(), dOR.data_ptr(), aout.data_ptr(), aoff.data_ptr(), tot, hip.stream))
    t = timed(add_only)
    alg = 2 * 8 * dl * tot
    print(f"add_ragged {name:<22} kernels only, cold: {t*1e6:8.1f} us  {alg/t/1e9:7.0f} GB/s ({100*alg/t/8e12:4.1f}% of peak)", flush=True)
    m1, m2 = int(max(t1s)), int(max(t2s))
    def add_bounded():
        l, r = sets[turn[0] % 3]; turn[0] += 1
        check(hip.lib.csgn_add_ragged_bounded(n, len(t1s), m1, m2, l.data_ptr(), dOL.data_ptr(), r.data_ptr(), dOR.data_ptr(), aout.data_ptr(), aoff.data_ptr(), tot, hip.stream))
    t = timed(add_bounded)
    alg = 2 * 8 * dl * tot
    print(f"add_ragged_bounded({m1},{m2}) {name:<14} kernels only, cold: {t*1e6:8.1f} us  {alg/t/1e9:7.0f} GB/s ({100*alg/t/8e12:4.1f}% of peak)", flush=True)
    mt = int(np.sum(np.asarray(t1s, dtype=np.int64) * np.asarray(t2s, dtype=np.int64)))
    out, off_out = hip.empty_words(mt * dl), hip.empty_words(len(t1s) + 1)
    plan = hip.empty_words(int(hip.lib.csgn_mul_ragged_async_plan_words(len(t1s))))
    def mul_async():
        l, r = sets[turn[0] % 3]; turn[0] += 1
        hip.mul_ragged_async(n, l, dOL, r, dOR, mt, out=out, off_out=off_out, plan=plan)
    t = timed(mul_async)
    alg = 8 * dl * (int(offL[-1]) + int(offR[-1]) + mt)
    print(f"mul_ragged_async {name:<16} cold: {t*1e6:8.1f} us  {alg/t/1e9:7.0f} GB/s ({100*alg/t/8e12:4.1f}% of peak)", flush=True)
    del sets, aout, out
for name, ts in [("1M single terms", [1] * (1 << 20)), ("one 1M-term + 65535 singles", [1 << 20] + [1] * 65535)]:
    off = csr(ts); tot = int(off[-1])
    Ws = [hip.synth_fill(3 + k, n, 0, tot * dl) for k in range(3)]
    doff = hip.upload(off)
    bits = torch.empty(len(ts), dtype=torch.uint8, device=hip.device)
    scratch = torch.empty(int(hip.lib.csgn_decrypt_scratch_bytes(len(ts), tot)), dtype=torch.uint8, device=hip.device)
    turn = [0]
    def dec():
        w = Ws[turn[0] % 3]; turn[0] += 1
        check(hip.lib.csgn_decrypt_ragged(n, len(ts), tot, w.data_ptr(), doff.data_ptr(), dmask.data_ptr(), bits.data_ptr(), scratch.data_ptr(), hip.stream))
    t = timed(dec)
    print(f"decrypt_ragged {name:<28} kernels only: {t*1e6:8.1f} us  {tot*dl*8/t/1e9:7.0f} GB/s ({100*tot*dl*8/t/8e12:4.1f}% of peak)", flush=True)
    mx = int(max(ts))
    def decb():
        w = Ws[turn[0] % 3]; turn[0] += 1
        check(hip.lib.csgn_decrypt_ragged_bounded(n, len(ts), tot, mx, w.data_ptr(), doff.data_ptr(), dmask.data_ptr(), bits.data_ptr(), scratch.data_ptr(), hip.stream))
    t = timed(decb)
    print(f"decrypt_ragged_bounded(max {mx}) {name:<14} kernels only: {t*1e6:8.1f} us  {tot*dl*8/t/1e9:7.0f} GB/s ({100*tot*dl*8/t/8e12:4.1f}% of peak)", flush=True)
    del Ws
