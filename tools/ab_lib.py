#!/usr/bin/env python3
"""Run tools/bench_ops.py-style decrypt timings against two library builds (dev tool).
usage: ab_lib.py <libA> <libB>   (each in its own subprocess, alternating, same device)"""
import os, subprocess, sys
code = r'''
import os, sys, statistics
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from csgn_amd.batch import HipPath
hip = HipPath(0)
def timed(fn, rounds=9):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(rounds):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/1e3)
    return statistics.median(ts)
out=[]
for n,d in [(1247,16),(4096,32)]:
    dl=hip.default_len(n)
    key=np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask=hip.upload(hip.key_mask(n,key))
    for terms,batch in [(1,1<<20),(1024,4096),(1<<20,8)]:
        W=hip.synth_fill(3,n,0,batch*terms*dl)
        t=timed(lambda: hip.decrypt_uniform(n,batch,terms,W,dmask))
        out.append("%.0f" % (batch*terms*8*dl/t/1e9)); del W
print(" ".join(out))
'''
for rnd in range(2):
    for lib in sys.argv[1:3]:
        env = dict(os.environ, CSGN_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
