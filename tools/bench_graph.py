#!/usr/bin/env python3
"""BASELINE config 5 as a captured circuit (csgn_circuit_*: one hipGraph launch) against the same
operations issued one by one, for small batches where the circuit is launch-bound (dev tool)."""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd.capi import check

hip = HipPath(0)
lib = hip.lib


def timed(fn, rounds=30):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)


for n, d, levels in [(4096, 32, 16), (1247, 16, 16)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask = hip.upload(hip.key_mask(n, key)); dkey = hip.upload(key)
    for B in (1, 16, 256, 4096):
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

        c = C.c_void_p()
        check(lib.csgn_circuit_create(n, B, C.byref(c)))
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
        check(lib.csgn_circuit_build(c))
        for i in range(nin):
            check(lib.csgn_memcpy_d2d(lib.csgn_circuit_value(c, ids[i]), inp(i).data_ptr(), B * dl * 8, hip.stream))
        check(lib.csgn_circuit_run(c, hip.stream)); torch.cuda.synchronize()
        xe, be = eager()
        terms = int(lib.csgn_circuit_value_terms(c, x))
        got = hip.empty_words(B * terms * dl)
        check(lib.csgn_memcpy_d2d(got.data_ptr(), lib.csgn_circuit_value(c, x), B * terms * dl * 8, hip.stream))
        gb = torch.empty(B, dtype=torch.uint8, device=got.device)
        check(lib.csgn_memcpy_d2d(gb.data_ptr(), lib.csgn_circuit_bits(c, bid.value), B, hip.stream))
        same = torch.equal(got, xe[:B * terms * dl]) and torch.equal(gb, be)
        tg = timed(lambda: check(lib.csgn_circuit_run(c, hip.stream)))
        te = timed(eager)
        print(f"Context({n},{d}) depth {levels} batch {B:5d}: graph {tg:8.1f} us | one by one {te:8.1f} us | "
              f"{te / tg:4.1f}x | identical {same} ({terms} terms)", flush=True)
        lib.csgn_circuit_destroy(c)
