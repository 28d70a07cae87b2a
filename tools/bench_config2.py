#!/usr/bin/env python3
"""BASELINE config 2 (65 536 independent fresh c*c at N=1247; 31 MB of traffic) and its end-to-end
form, with the launch cost separated from the kernel time (dev tool):
  a. csgn_mul_uniform 1x1, one call bracketed by HIP events (what a caller sees for one batch);
  b. the same call 200 times back to back (per-call time once launches overlap the previous kernel);
  c. Enc, Enc -> * -> Dec as ONE hipGraph (csgn_circuit_encrypt): per replay;
  d. the same four operations issued one by one;
  e. the chain as ONE fused node (csgn_circuit_encrypt_mul) in a graph, and as one csgn_encrypt_mul_keyed call.
Run it under `rocprofv3 --kernel-trace --stats` for the pure kernel durations (k_and_stream,
k_encrypt_wave, k_term_hits_seg); profiles/r02/config2.* keeps both."""
import ctypes as C
import json
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from csgn_amd.batch import HipPath
from csgn_amd.capi import check

hip = HipPath(0)
lib = hip.lib
n, d, dl = 1247, 16, 20
res = {}
for batch in (65536, 1 << 20):
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask, dkey = hip.upload(hip.key_mask(n, key)), hip.upload(key)
    L = hip.synth_fill(1, n, 0, batch * dl)
    R = hip.synth_fill(2, n, 0, batch * dl)
    out = hip.empty_words(batch * dl)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def one():
        hip.mul_uniform(n, batch, 1, 1, L, R, out=out)
    for _ in range(20):
        one()
    torch.cuda.synchronize()
    singles = []
    for _ in range(50):
        a, b = ev(), ev()
        a.record(); one(); b.record(); b.synchronize()
        singles.append(a.elapsed_time(b) * 1e3)
    a, b = ev(), ev()
    a.record()
    for _ in range(200):
        one()
    b.record(); b.synchronize()
    back_to_back = a.elapsed_time(b) * 1e3 / 200

    # end to end in one graph
    pa = hip.upload(np.random.default_rng(3).integers(0, 2, batch).astype(np.uint8))
    pb = hip.upload(np.random.default_rng(4).integers(0, 2, batch).astype(np.uint8))
    rng = hip.rng_from_seed(7, 8)
    c = C.c_void_p()
    check(lib.csgn_circuit_create(n, batch, C.byref(c)))
    def new(fn, *args):
        v = C.c_uint32()
        check(fn(c, *args, C.byref(v)))
        return v.value
    va = new(lib.csgn_circuit_encrypt, d, pa.data_ptr(), dkey.data_ptr(), dmask.data_ptr(), C.byref(rng), 0)
    vb = new(lib.csgn_circuit_encrypt, d, pb.data_ptr(), dkey.data_ptr(), dmask.data_ptr(), C.byref(rng), 1 << 40)
    vm = new(lib.csgn_circuit_mul, va, vb)
    bm = new(lib.csgn_circuit_decrypt, vm, dmask.data_ptr())
    check(lib.csgn_circuit_build(c))
    for _ in range(5):
        check(lib.csgn_circuit_run(c, hip.stream))
    torch.cuda.synchronize()
    graph = []
    for _ in range(50):
        a, b = ev(), ev()
        a.record(); check(lib.csgn_circuit_run(c, hip.stream)); b.record(); b.synchronize()
        graph.append(a.elapsed_time(b) * 1e3)
    lib.csgn_circuit_destroy(c)

    # the same chain as ONE fused node (csgn_circuit_encrypt_mul: both operands in registers, product + bits out)
    rng_b = hip.rng_from_seed(8, 8)
    c = C.c_void_p()
    check(lib.csgn_circuit_create(n, batch, C.byref(c)))
    vf, bf = C.c_uint32(), C.c_uint32()
    check(lib.csgn_circuit_encrypt_mul(c, d, pa.data_ptr(), pb.data_ptr(), dkey.data_ptr(), dmask.data_ptr(), C.byref(rng),
                                       C.byref(rng_b), 0, C.byref(vf), C.byref(bf)))
    check(lib.csgn_circuit_build(c))
    for _ in range(5):
        check(lib.csgn_circuit_run(c, hip.stream))
    torch.cuda.synchronize()
    fused_graph = []
    for _ in range(50):
        a, b = ev(), ev()
        a.record(); check(lib.csgn_circuit_run(c, hip.stream)); b.record(); b.synchronize()
        fused_graph.append(a.elapsed_time(b) * 1e3)
    lib.csgn_circuit_destroy(c)
    fused_call = []
    for _ in range(50):
        a, b = ev(), ev()
        a.record(); hip.encrypt_mul_keyed(n, d, pa, pb, dkey, dmask, rng, rng_b); b.record(); b.synchronize()
        fused_call.append(a.elapsed_time(b) * 1e3)

    ca, cb = hip.empty_words(batch * dl), hip.empty_words(batch * dl)
    def by_hand():
        hip.encrypt_keyed(n, d, pa, dkey, dmask, rng, 0, out=ca)
        hip.encrypt_keyed(n, d, pb, dkey, dmask, rng, 1 << 40, out=cb)
        hip.mul_uniform(n, batch, 1, 1, ca, cb, out=out)
        return hip.decrypt_uniform(n, batch, 1, out, dmask)
    for _ in range(5):
        by_hand()
    torch.cuda.synchronize()
    hand = []
    for _ in range(50):
        a, b = ev(), ev()
        a.record(); by_hand(); b.record(); b.synchronize()
        hand.append(a.elapsed_time(b) * 1e3)
    bytes_mul = batch * 3 * 8 * dl
    bytes_e2e = batch * 8 * dl * (2 + 3 + 1)          # two encrypts written, product read+read+written, decrypt read
    res[str(batch)] = {
        "mul_1x1_single_call_us": statistics.median(singles),
        "mul_1x1_back_to_back_us": back_to_back,
        "mul_1x1_GBps_back_to_back": bytes_mul / back_to_back / 1e3,
        "enc_enc_mul_dec_graph_us": statistics.median(graph),
        "enc_enc_mul_dec_one_by_one_us": statistics.median(hand),
        "fused_node_graph_us": statistics.median(fused_graph),
        "fused_csgn_encrypt_mul_keyed_call_us": statistics.median(fused_call),
        "mult_per_s_fused_graph": batch / statistics.median(fused_graph) * 1e6,
        "end_to_end_GBps_graph": bytes_e2e / statistics.median(graph) / 1e3,
        "mult_per_s_graph": batch / statistics.median(graph) * 1e6,
    }
print(json.dumps(res, indent=1))
