#!/usr/bin/env python3
"""Ragged (CSR) multiply / add / decrypt throughput on skewed batches (dev tool)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath, check
from csgn_amd import capi
CHUNKS=[int(x) for x in os.environ.get('CHUNKS','0').split(',')]
NSETS=[3]
VARIANTS=[('same operands',{'NSETS':'1'}),('cold',{}),('cold, no touch',{'CSGN_RAGGED_TOUCH':'0'}),
          ('cold C=1',{'CSGN_RAGGED_C':'1'}),('cold C=2',{'CSGN_RAGGED_C':'2'}),('cold C=4',{'CSGN_RAGGED_C':'4'}),('cold C=4 M=2',{'CSGN_RAGGED_C':'4','CSGN_RAGGED_M':'2'}),('cold C=4 M=1',{'CSGN_RAGGED_C':'4','CSGN_RAGGED_M':'1'}),
          ('cold C=4 no touch',{'CSGN_RAGGED_C':'4','CSGN_RAGGED_TOUCH':'0'}),('cold C=4 no pf',{'CSGN_RAGGED_C':'4','CSGN_RAGGED_PF':'0'}),('cold C=8',{'CSGN_RAGGED_C':'8'}),('cold C=16',{'CSGN_RAGGED_C':'16'}),('cold C=8 M=2',{'CSGN_RAGGED_C':'8','CSGN_RAGGED_M':'2'}),('cold C=8 no pf',{'CSGN_RAGGED_C':'8','CSGN_RAGGED_PF':'0'}),('same C=4',{'NSETS':'1','CSGN_RAGGED_C':'4'}),
          ('CSR kernel forced: cold',{'CSGN_RAGGED_FLAT':'1'})]
if os.environ.get('SHORT'):          # SHORT=1: the default dispatch and the wave-cooperative kernel's choices
    VARIANTS=[('cold',{}),('same operands',{'NSETS':'1'}),
              ('CSR kernel (coop off)',{'CSGN_RAGGED_COOP':'0'}),
              ('one launch, in-kernel touch',{'CSGN_RAGGED_TOUCH':'0'}),
              ('one launch, no touch',{'CSGN_RAGGED_TOUCH':'0','CSGN_RAGGED_COOP_TOUCH':'0'}),
              ('slices behind k_touch_ragged',{'CSGN_RAGGED_COOP_TOUCH':'0'}),
              ('coop forced',{'CSGN_RAGGED_COOP':'1'}),('coop k2',{'CSGN_RAGGED_COOP_K':'2'}),('coop pipelined',{'CSGN_RAGGED_COOP_PIPE':'1'}),('span 32',{'CSGN_RAGGED_COOP_SPAN':'32'}),('span 128',{'CSGN_RAGGED_COOP_SPAN':'128'}),('touch 32 KiB',{'CSGN_RAGGED_COOP_TOUCH':'32'})]
hip = HipPath(0)
def timed(fn, rounds=7):
    """Steady-state time per call (as tools/bench_ops.py): >= 30 ms of back-to-back warm-up, then runs of K calls
    (>= 2 ms) between one pair of events -- the shader clock needs milliseconds to come back after an idle gap."""
    def bracket(k):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k): fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b)/1e3/k
    est=bracket(1); spent=est
    while spent<30e-3:
        k=max(1,min(64,int(5e-3/max(est,1e-6)))); est=bracket(k); spent+=est*k
    k=max(1,min(64,int(2e-3/max(est,1e-6))+1))
    return statistics.median([bracket(k) for _ in range(rounds)])
def csr(c):
    o=np.zeros(len(c)+1,dtype=np.uint64); o[1:]=np.cumsum(np.asarray(c,dtype=np.uint64)); return o
n=1247; dl=20
rng=np.random.default_rng(0)
lg=lambda mean,cnt,cap: np.clip(rng.lognormal(np.log(mean)-0.5,1,cnt),1,cap).astype(int)
for name,t1s,t2s in [
    ("lognormal mean~8 x262144 (long tail)", lg(8,1<<18,600), lg(8,1<<18,600)),
    ("lognormal mean~16 x65536 (long tail)", lg(16,1<<16,1000), lg(16,1<<16,1000)),
    ("uniform 64x64 x4096", [64]*4096, [64]*4096),
    ("lognormal mean~32 x16384", np.clip(rng.lognormal(3,1,16384),1,2000).astype(int), np.clip(rng.lognormal(3,1,16384),1,2000).astype(int)),
    ("one 1024x1024 + 65535 1x1", [1024]+[1]*65535, [1024]+[1]*65535),
    ("fresh 1x1 x1M (ragged path)", [1]*(1<<20), [1]*(1<<20)),
]:
    offL,offR=csr(t1s),csr(t2s)
    L=hip.synth_fill(1,n,0,int(offL[-1])*dl); R=hip.synth_fill(2,n,0,int(offR[-1])*dl)
    dL_,dR_=hip.upload(offL),hip.upload(offR)
    outw=int(np.sum(np.asarray(t1s,dtype=np.int64)*np.asarray(t2s,dtype=np.int64)))*dl
    alg=8*(int(offL[-1])*dl+int(offR[-1])*dl+outw)
    t=timed(lambda: hip.mul_ragged(n,L,dL_,R,dR_))
    print(f"mul_ragged {name:<32} {t*1e3:8.3f} ms  {alg/t/1e9:8.1f} GB/s ({100*alg/t/8e12:4.1f}% of peak), out {outw*8/1e6:.0f} MB  [plan + alloc + kernel]", flush=True)
    # the kernel alone (offsets planned once), COLD operands: three operand sets with the same
    # offsets taken in turn (> 256 MB together for the skewed batches), per setting
    out, off_out = hip.mul_ragged(n,L,dL_,R,dR_)
    mt1, mt2, tot = int(max(t1s)), int(max(t2s)), outw // dl
    # csgn_mul_ragged_async: plan kernels + multiply enqueued back to back, nothing read back (buffers preallocated,
    # the bound = the real size); against the kernel-only figure below this is what the host round trip used to cost
    aplan = hip.empty_words(int(hip.lib.csgn_mul_ragged_async_plan_words(len(t1s))))
    t=timed(lambda: hip.mul_ragged_async(n,L,dL_,R,dR_,tot,out=out,off_out=off_out,plan=aplan))
    assert hip.mul_ragged_async_result(aplan)[4] == 0
    print(f"mul_ragged_async {name:<26} {t*1e3:8.3f} ms  {alg/t/1e9:8.1f} GB/s ({100*alg/t/8e12:4.1f}% of peak)  [plan kernels + multiply, no host round trip, warm operands]", flush=True)
    # the plan object the kernel-only runs multiply by (trusted: the offsets do not change here)
    import ctypes as C
    handle = hip.mul_plan(); hplan = (C.c_uint64 * 4)()
    check(hip.lib.csgn_mul_plan_ragged(handle, len(t1s), dL_.data_ptr(), dR_.data_ptr(), off_out.data_ptr(), C.byref(hplan), hip.stream))
    check(hip.lib.csgn_mul_plan_trust(handle, 1))
    sets=[(L,R)]+[(hip.synth_fill(10+k,n,0,int(offL[-1])*dl), hip.synth_fill(20+k,n,0,int(offR[-1])*dl)) for k in range(2)]
    row=[]
    for label,env in VARIANTS:
        capi.reset_tuning()
        for k,v in env.items():
            if k.startswith('CSGN'): capi.set_tuning(k, v)
        NSETS[0]=int(env.get('NSETS','3'))
        turn=[0]
        def one():
            Lk,Rk=sets[turn[0]%NSETS[0]]; turn[0]+=1
            check(hip.lib.csgn_mul_planned(handle,n,Lk.data_ptr(),Rk.data_ptr(),out.data_ptr(),hip.stream))
        if env.get("PRETOUCH"):
            ts=[]
            for _ in range(9):
                Lk,Rk=sets[turn[0]%NSETS[0]]
                Lk.view(torch.int64).sum(); Rk.view(torch.int64).sum()      # operands into the caches, untimed
                a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
                a.record(); one(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/1e3)
            t=statistics.median(ts)
        else:
            t=timed(one, rounds=9)
        row.append(f"{label}: {alg/t/1e9:6.0f}")
    capi.reset_tuning()
    print(f"   kernel only, GB/s   " + "  ".join(row), flush=True)
    # the same through csgn_mul_ragged_async (cold operand sets in turn)
    turn=[0]
    def one_async():
        Lk,Rk=sets[turn[0]%3]; turn[0]+=1
        hip.mul_ragged_async(n,Lk,dL_,Rk,dR_,tot,out=out,off_out=off_out,plan=aplan)
    t=timed(one_async, rounds=9)
    print(f"   csgn_mul_ragged_async, cold operands: {alg/t/1e9:6.0f} GB/s ({t*1e6:7.1f} us)", flush=True)
    hip.lib.csgn_mul_plan_destroy(handle)
    del out
    alg=2*8*dl*(int(offL[-1])+int(offR[-1]))
    tot=int(offL[-1]+offR[-1])
    row=[]
    aturn=[0]
    def next_set():
        Lk,Rk=sets[aturn[0]%3]; aturn[0]+=1
        return Lk,Rk
    for ch in CHUNKS:
        capi.set_tuning("ragged_c", ch)
        def add_alloc():
            Lk,Rk=next_set(); hip.add_ragged(n,Lk,dL_,Rk,dR_, total_terms_out=tot)
        t=timed(add_alloc)
        row.append(f"C={ch}: {alg/t/1e9:6.0f}")
    print(f"add_ragged {name:<32} GB/s (alloc + kernel, cold operands)   " + "  ".join(row), flush=True)
    # the add kernels alone (output and offsets preallocated; k_off_sum + the CSR kernel), default knobs, the three
    # operand sets in turn (cold)
    capi.reset_tuning()
    aout, aoff = hip.empty_words(tot*dl), hip.empty_words(len(t1s)+1)
    def add_only():
        Lk,Rk=next_set()
        check(hip.lib.csgn_add_ragged(n,len(t1s),Lk.data_ptr(),dL_.data_ptr(),Rk.data_ptr(),dR_.data_ptr(),
                                      aout.data_ptr(),aoff.data_ptr(),tot,hip.stream))
    t=timed(add_only, rounds=9)
    print(f"add_ragged {name:<32} kernels only, cold: {alg/t/1e9:6.0f} GB/s ({100*alg/t/8e12:4.1f}% of peak, {t*1e6:7.1f} us)", flush=True)
    del L,R,aout,aoff,sets

# ragged decrypt: term lists of skewed lengths (the products of the batches above have this shape)
key = np.random.default_rng(1).permutation(n)[:16].astype(np.uint64)
dmask = hip.upload(hip.key_mask(n, key))
for name, ts in [
    ("uniform 4096 terms x4096", [4096] * 4096),
    ("lognormal mean~1100 x16384", np.clip(rng.lognormal(6.5, 1, 16384), 1, 200000).astype(int)),
    ("one 1M-term + 65535 single terms", [1 << 20] + [1] * 65535),
    ("1M single terms (ragged path)", [1] * (1 << 20)),
]:
    off = csr(ts)
    tot = int(off[-1])
    # inputs rotate through >= 600 MB (a 170 MB list re-read back to back comes out of the memory-side cache)
    nw = max(1, min(4, -(-int(600e6) // (tot * dl * 8)))) if tot * dl * 8 < 600e6 else 1
    Ws = [hip.synth_fill(3 + k, n, 0, tot * dl) for k in range(nw)]
    doff = hip.upload(off)
    wt = [0]
    def next_w():
        w = Ws[wt[0] % nw]; wt[0] += 1
        return w
    t = timed(lambda: hip.decrypt_ragged(n, next_w(), doff, dmask, total_terms=tot))
    print(f"decrypt_ragged {name:<34} {t*1e3:8.3f} ms  {tot*dl*8/t/1e9:8.1f} GB/s ({100*tot*dl*8/t/8e12:4.1f}% of peak), {tot*dl*8/1e6:.0f} MB, {nw} input set(s)", flush=True)
    # kernels only (bits and scratch preallocated)
    bits = torch.empty(len(ts), dtype=torch.uint8, device=hip.device)
    scratch = torch.empty(int(hip.lib.csgn_decrypt_scratch_bytes(len(ts), tot)), dtype=torch.uint8, device=hip.device)
    t = timed(lambda: check(hip.lib.csgn_decrypt_ragged(n, len(ts), tot, next_w().data_ptr(), doff.data_ptr(), dmask.data_ptr(),
                                                        bits.data_ptr(), scratch.data_ptr(), hip.stream)), rounds=9)
    print(f"decrypt_ragged {name:<34} kernels only: {tot*dl*8/t/1e9:8.1f} GB/s ({100*tot*dl*8/t/8e12:4.1f}% of peak, {t*1e6:7.1f} us)", flush=True)
    del Ws
