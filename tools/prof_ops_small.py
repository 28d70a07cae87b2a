#!/usr/bin/env python3
"""A few launches of the VALU-heavy secondary kernels (keyed encrypt, bit-plane permutation) for
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES / --kernel-trace --stats (dev tool; see tools/prof_r02_ops.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath

hip = HipPath(0)
for n, d, batch in [(1247, 16, 1 << 22), (4096, 32, 1 << 20)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask, dkey = hip.upload(hip.key_mask(n, key)), hip.upload(key)
    plain = hip.upload(np.random.default_rng(2).integers(0, 2, batch).astype(np.uint8))
    out = hip.empty_words(batch * dl)
    perm = hip.upload(np.random.default_rng(3).permutation(n).astype(np.uint32))
    for rounds in (8, 20):
        rng = hip.rng_from_seed(3, rounds)
        for _ in range(3):
            hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=out)
    plain_b = hip.upload(np.random.default_rng(4).integers(0, 2, batch).astype(np.uint8))
    ra, rb = hip.rng_from_seed(3, 8), hip.rng_from_seed(4, 8)
    for _ in range(3):
        hip.encrypt_mul_keyed(n, d, plain, plain_b, dkey, dmask, ra, rb)      # the fused fresh chain
    for _ in range(3):
        hip.permute_uniform(n, batch, 1, out, perm)
    torch.cuda.synchronize()
    del out
print("done")
