#!/usr/bin/env python3
"""encrypt: explicit-randomness vs device-RNG throughput (dev tool)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd import capi
hip = HipPath(0)
def timed(fn, rounds=9):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(rounds):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/1e3)
    return statistics.median(ts)
for n,d in [(1247,16),(4096,32)]:
    dl=hip.default_len(n); batch=1<<20
    key=np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask=hip.upload(hip.key_mask(n,key)); dkey=hip.upload(key)
    plain=hip.upload(np.random.default_rng(2).integers(0,2,batch).astype(np.uint8))
    rnd=hip.synth_fill(9,n,0,batch*dl)
    chosen=hip.upload(np.random.default_rng(3).choice(key,batch).astype(np.uint32))
    last=hip.upload(np.random.default_rng(4).integers(0,2,batch).astype(np.uint8))
    for form,env in (("seg",{}),("lds",{"CSGN_ENC_LDS":"1"})):
        capi.reset_tuning()
        for k,v in env.items(): capi.set_tuning(k, v)
        lds=form
        te=timed(lambda: hip.encrypt_explicit(n,d,plain,rnd,chosen,last,dmask))
        tr=timed(lambda: hip.encrypt_device_rng(n,d,plain,dkey,dmask,7))
        print(f"N={n} {lds}: explicit {batch*dl*8/te/1e9:7.0f} GB/s out ({2*batch*dl*8/te/1e9:7.0f} in+out) | device rng {batch*dl*8/tr/1e9:7.0f} GB/s out", flush=True)
