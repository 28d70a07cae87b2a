#!/usr/bin/env python3
"""csgn_encrypt_keyed throughput: ChaCha rounds 8/12/20, wave kernel vs one-lane-per-ciphertext kernel,
and the explicit-randomness (parity) form beside it (dev tool).  GB/s of ciphertext written."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd import capi

hip = HipPath(0)


def timed(fn, rounds=9):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) / 1e3)
    return statistics.median(ts)


for n, d in [(1247, 16), (4096, 32), (2048, 16)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask, dkey = hip.upload(hip.key_mask(n, key)), hip.upload(key)
    for batch in (1 << 16, 1 << 20, 1 << 22):
        plain = hip.upload(np.random.default_rng(2).integers(0, 2, batch).astype(np.uint8))
        out = hip.empty_words(batch * dl)
        row = []
        for rounds in (8, 12, 20):
            rng = hip.rng_from_seed(3, rounds)
            for wave in ((1, -1, 8, 32, 64, 0) if batch >= 1 << 20 and rounds == 8 else (1,)):
                capi.reset_tuning()
                if wave == -1:
                    capi.set_tuning("enc_compact", 0)          # full LDS tables
                else:
                    capi.set_tuning("enc_wave", wave)
                t = timed(lambda: hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=out))
                row.append(f"chacha{rounds}{'' if wave == 1 else '/ct-kernel' if wave == 0 else '/full-tables' if wave == -1 else '/wg%d' % wave} {batch*dl*8/t/1e9:6.0f} GB/s ({batch/t/1e9:5.2f} Gct/s, {t*1e6:7.1f} us)")
        capi.reset_tuning()
        if batch == 1 << 20:
            rnd = hip.synth_fill(9, n, 0, batch * dl)
            chosen = hip.upload(np.random.default_rng(3).choice(key, batch).astype(np.uint32))
            last = hip.upload(np.random.default_rng(4).integers(0, 2, batch).astype(np.uint8))
            t = timed(lambda: hip.encrypt_explicit(n, d, plain, rnd, chosen, last, dmask))
            row.append(f"explicit {2*batch*dl*8/t/1e9:6.0f} GB/s in+out")
            del rnd
        print(f"N={n} batch={batch}: " + " | ".join(row), flush=True)
        del out
