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
for n in (1247, 4096):
    dl = hip.default_len(n)
    for t1, t2, batch in [(2, 2, 262144), (4, 4, 131072), (8, 8, 32768), (8, 8, 131072), (16, 16, 16384), (4, 16, 32768), (16, 4, 32768), (8, 32, 8192), (32, 8, 8192), (32, 32, 4096), (64, 64, 1024)]:
        # three operand sets in turn: cold operands, as in a chain where each product is new
        sets = [(hip.synth_fill(1 + k, n, 0, batch * t1 * dl), hip.synth_fill(11 + k, n, 0, batch * t2 * dl)) for k in range(3)]
        out = hip.empty_words(batch * t1 * t2 * dl)
        row = []
        for name, kn in [("default", {}), ("touch0", {"mul_touch": 0}), ("touch3", {"mul_touch": 3}), ("touch1", {"mul_touch": 1}), ("touch2", {"mul_touch": 2}), ("tiled", {"mul_flat": -1}), ("flat2 t0", {"mul_flat": 2, "mul_touch": 0})]:
            capi.reset_tuning()
            for k, v in kn.items():
                capi.set_tuning(k, v)
            it = [0]
            def run():
                L, R = sets[it[0] % 3]; it[0] += 1
                hip.mul_uniform(n, batch, t1, t2, L, R, out=out)
            t = timed(run)
            kname = hip.lib.csgn_mul_uniform_kernel(n, batch, t1, t2).decode()
            row.append("%s %.0f [%s]" % (name, batch * 8 * dl * (t1 + t2 + t1 * t2) / t / 1e9, kname))
        capi.reset_tuning()
        print(f"N={n} {t1}x{t2} x{batch}: " + " | ".join(row), flush=True)
        del sets, out
