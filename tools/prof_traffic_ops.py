#!/usr/bin/env python3
"""Three launches of every secondary kernel at a large size, with the algorithmic bytes of each printed, for the
HBM-traffic PMC passes of tools/prof_r03_traffic.sh (dev tool)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd.capi import check

hip = HipPath(0)
alg = {}
n, d = 1247, 16
dl = hip.default_len(n)
key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
dmask, dkey = hip.upload(hip.key_mask(n, key)), hip.upload(key)
B = 1 << 20
L, R = hip.synth_fill(1, n, 0, B * dl), hip.synth_fill(2, n, 0, B * dl)
out = hip.empty_words(2 * B * dl)
for _ in range(3):
    hip.mul_uniform(n, B, 1, 1, L, R, out=out)
alg["k_and_stream"] = {"read": 2 * B * dl * 8, "written": B * dl * 8, "what": "mul 1x1 x 1 M"}
for _ in range(3):
    check(hip.lib.csgn_add_uniform(n, B, 1, 1, L.data_ptr(), R.data_ptr(), out.data_ptr(), hip.stream))
alg["k_add_flat"] = {"read": 2 * B * dl * 8, "written": 2 * B * dl * 8, "what": "add 1+1 x 1 M"}
bits = torch.empty(B, dtype=torch.uint8, device=hip.device)
scratch = torch.empty(int(hip.lib.csgn_decrypt_scratch_bytes(4096, B)), dtype=torch.uint8, device=hip.device)
for _ in range(3):
    check(hip.lib.csgn_decrypt_uniform(n, 4096, 256, L.data_ptr(), dmask.data_ptr(), bits.data_ptr(), scratch.data_ptr(), hip.stream))
alg["k_term_hits_seg"] = {"read": B * dl * 8, "written": B // 8, "what": "decrypt, 4096 ciphertexts of 256 terms (pass 1)"}
plain = hip.upload(np.random.default_rng(2).integers(0, 2, B).astype(np.uint8))
rng = hip.rng_from_seed(3, 8)
for _ in range(3):
    hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=out[: B * dl])
alg["k_encrypt_wave"] = {"read": B, "written": B * dl * 8, "what": "keyed encrypt x 1 M"}
ra, rb = hip.rng_from_seed(3, 8), hip.rng_from_seed(4, 8)
for _ in range(3):
    hip.encrypt_mul_keyed(n, d, plain, plain, dkey, dmask, ra, rb)
alg["k_encrypt_mul_wave"] = {"read": 2 * B, "written": B * dl * 8, "what": "fused Enc*Enc x 1 M"}
perm = hip.upload(np.random.default_rng(3).permutation(n).astype(np.uint32))
for _ in range(3):
    check(hip.lib.csgn_permute_uniform(n, B, 1, 0, L.data_ptr(), perm.data_ptr(), out.data_ptr(), hip.stream))
alg["k_permute_planes3"] = {"read": B * dl * 8, "written": B * dl * 8, "what": "permutation x 1 M"}
t = 64
Lm, Rm = hip.synth_fill(5, n, 0, 4096 * t * dl), hip.synth_fill(6, n, 0, 4096 * t * dl)
big = hip.empty_words(4096 * t * t * dl)
for _ in range(3):
    hip.mul_uniform(n, 4096, t, t, Lm, Rm, out=big)
alg["k_mul_flat"] = {"read": 2 * 4096 * t * dl * 8, "written": 4096 * t * t * dl * 8, "what": "mul 64x64 x 4096 (+ k_touch2 reads the operands once more)"}
torch.cuda.synchronize()
print("ALG " + json.dumps(alg))
