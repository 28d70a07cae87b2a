#!/usr/bin/env python3
"""BASELINE config 5 as a circuit (csgn_circuit_*: one hipGraph launch): the TAPE (every value materialised, one kernel
per node), the COMPILED graph (csgn_circuit_optimize: liveness, producers placed into the sums that consume them, the
last product fused into the decrypt), the compiled graph with the final value kept as an output, and the same
operations issued one by one.  Prints, per form, the time of a run, the ALGORITHMIC bytes the graph's kernels move
(csgn_circuit_stats: SURVEY 8d's per-operation figures summed over the emitted kernels) and those bytes / time as
a fraction of the 8.0 TB/s HBM peak; checks bits (all forms) and words (tape, kept) against the one-by-one calls.
Dev tool; the numbers of record are bench.py's "secondary" entries."""
import ctypes as C, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd.capi import check

PEAK = 8.0e12
hip = HipPath(0)
lib = hip.lib


def timed(fn, rounds=30):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)


def describe(n, B, levels, dmask, flags=0, keep_final=False):
    c = C.c_void_p()
    check(lib.csgn_circuit_create(n, B, C.byref(c)))
    nin = 1 + levels // 2 + 2 * (levels // 2)
    ids = []
    for i in range(nin):
        v = C.c_uint32(); check(lib.csgn_circuit_input(c, 1, C.byref(v))); ids.append(v.value)
    x, k = ids[0], 1
    for level in range(1, levels + 1):
        v = C.c_uint32()
        if level % 2:
            check(lib.csgn_circuit_add(c, x, ids[k], C.byref(v))); k += 1
        else:
            r = C.c_uint32(); check(lib.csgn_circuit_add(c, ids[k], ids[k + 1], C.byref(r)))
            check(lib.csgn_circuit_mul(c, x, r.value, C.byref(v))); k += 2
        x = v.value
    bid = C.c_uint32(); check(lib.csgn_circuit_decrypt(c, x, dmask.data_ptr(), C.byref(bid)))
    if flags:
        check(lib.csgn_circuit_optimize(c, flags))
    if keep_final:
        check(lib.csgn_circuit_output(c, x))
    check(lib.csgn_circuit_build(c))
    st = (C.c_uint64 * 8)(); check(lib.csgn_circuit_stats(c, st))
    return c, ids, x, bid.value, list(st)


batches = [int(a) for a in sys.argv[1:]] or [1, 16, 256, 4096]
results = []
for n, d, levels in [(4096, 32, 16), (1247, 16, 16)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask = hip.upload(hip.key_mask(n, key)); dkey = hip.upload(key)
    for B in batches:
        nin = 1 + levels // 2 + 2 * (levels // 2)
        plain = np.random.default_rng(B).integers(0, 2, size=(nin, B)).astype(np.uint8)
        fresh = hip.encrypt_device_rng(n, d, hip.upload(plain.reshape(-1)), dkey, dmask, seed=B)
        inp = lambda i: fresh[i * B * dl:(i + 1) * B * dl]

        def eager():
            x, xt, k = inp(0), 1, 1
            for level in range(1, levels + 1):
                if level % 2:
                    x = hip.add_uniform(n, B, xt, 1, x, inp(k)); xt += 1; k += 1
                else:
                    rhs = hip.add_uniform(n, B, 1, 1, inp(k), inp(k + 1))
                    x = hip.mul_uniform(n, B, xt, 2, x, rhs); xt *= 2; k += 2
            return x, hip.decrypt_uniform(n, B, xt, x, dmask)

        xe, be = eager()
        te = timed(eager)
        line = {"n": n, "d": d, "depth": levels, "batch": B, "one_by_one_us": round(te, 1)}
        for name, flags, keep in (("tape", 0, False), ("compiled", 23, False), ("compiled_keep_x", 23, True),
                                  ("pushdown", 31, False)):
            c, ids, x, bid, st = describe(n, B, levels, dmask, flags, keep)
            for i in range(nin):
                check(lib.csgn_memcpy_d2d(lib.csgn_circuit_value(c, ids[i]), inp(i).data_ptr(), B * dl * 8, hip.stream))
            check(lib.csgn_circuit_run(c, hip.stream)); torch.cuda.synchronize()
            gb = torch.empty(B, dtype=torch.uint8, device=fresh.device)
            check(lib.csgn_memcpy_d2d(gb.data_ptr(), lib.csgn_circuit_bits(c, bid), B, hip.stream))
            same = torch.equal(gb, be)
            ptr = lib.csgn_circuit_value(c, x)
            if ptr is not None:
                terms = int(lib.csgn_circuit_value_terms(c, x))
                got = hip.empty_words(B * terms * dl)
                check(lib.csgn_memcpy_d2d(got.data_ptr(), ptr, B * terms * dl * 8, hip.stream))
                same = same and torch.equal(got, xe[:B * terms * dl])
            t = timed(lambda: check(lib.csgn_circuit_run(c, hip.stream)))
            line[name] = {"us": round(t, 1), "alg_bytes": st[1], "TBps": round(st[1] / t / 1e6, 3),
                          "frac": round(st[1] / (t * 1e-6) / PEAK, 4), "block_bytes": st[0], "kernels": st[3],
                          "placed": st[4], "fused": st[5], "dropped": st[6], "hoisted": st[7], "identical": bool(same)}
            lib.csgn_circuit_destroy(c)
        results.append(line)
        f = lambda k: f"{line[k]['us']:8.1f} us {line[k]['alg_bytes'] / 1e6:9.2f} MB {100 * line[k]['frac']:5.1f}%"
        print(f"Context({n},{d}) depth {levels} batch {B:5d}: one by one {te:8.1f} us | tape {f('tape')} | compiled "
              f"{f('compiled')} | keep x {f('compiled_keep_x')} | pushdown {line['pushdown']['us']:6.1f} us | identical "
              f"{all(line[k]['identical'] for k in ('tape', 'compiled', 'compiled_keep_x', 'pushdown'))}", flush=True)
print(json.dumps(results))
