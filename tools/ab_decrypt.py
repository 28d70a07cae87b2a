#!/usr/bin/env python3
"""A/B of the decrypt pass-1 forms in one process (dev tool)."""
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
    dl=hip.default_len(n)
    key=np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask=hip.upload(hip.key_mask(n,key))
    for terms,batch in [(1,1<<20),(1024,4096),(1<<20,8)]:
        W=hip.synth_fill(3,n,0,batch*terms*dl)
        res={}
        for rnd in range(2):
            for form in ("0","1"):
                capi.set_tuning("dec_loop", form)
                t=timed(lambda: hip.decrypt_uniform(n,batch,terms,W,dmask))
                res.setdefault(form,[]).append(batch*terms*8*dl/t/1e9)
        print(f"N={n} T={terms} batch={batch}: seg {res['0']} GB/s | loop {res['1']} GB/s", flush=True)
        del W
