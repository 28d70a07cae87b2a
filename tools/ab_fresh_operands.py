"""Are operands that the previous kernel just WROTE warm or cold for the multiply?  (dev probe)
L and R are produced by csgn_add_uniform right before every timed multiply, into buffers that rotate through
more memory than the 256 MB memory-side cache; multiply timed alone, default touch policy vs none."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd import capi
from csgn_amd.capi import check
hip = HipPath(0)
for n in (1247, 4096):
    dl = hip.default_len(n)
    for t, batch in [(8, 32768), (16, 16384), (32, 4096), (64, 1024), (8, 131072)]:
        h = t // 2
        A1 = hip.synth_fill(1, n, 0, batch * h * dl); A2 = hip.synth_fill(2, n, 0, batch * h * dl)
        B1 = hip.synth_fill(3, n, 0, batch * h * dl); B2 = hip.synth_fill(4, n, 0, batch * h * dl)
        op_bytes = batch * t * dl * 8
        nsets = max(3, int(600e6 // (2 * op_bytes)) + 1)
        Ls = [hip.empty_words(batch * t * dl) for _ in range(nsets)]
        Rs = [hip.empty_words(batch * t * dl) for _ in range(nsets)]
        out = hip.empty_words(batch * t * t * dl)
        row = []
        for name, kn in [("default", {}), ("touch0", {"mul_touch": 0}), ("default", {}), ("touch0", {"mul_touch": 0})]:
            ts = []
            for it in range(12):
                k = it % nsets
                capi.reset_tuning()
                check(hip.lib.csgn_add_uniform(n, batch, h, h, A1.data_ptr(), A2.data_ptr(), Ls[k].data_ptr(), hip.stream))
                check(hip.lib.csgn_add_uniform(n, batch, h, h, B1.data_ptr(), B2.data_ptr(), Rs[k].data_ptr(), hip.stream))
                for kk, v in kn.items():
                    capi.set_tuning(kk, v)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); hip.mul_uniform(n, batch, t, t, Ls[k], Rs[k], out=out); b.record(); b.synchronize()
                if it >= 2:
                    ts.append(a.elapsed_time(b) / 1e3)
            tm = statistics.median(ts)
            row.append("%s %.0f" % (name, batch * 8 * dl * (2 * t + t * t) / tm / 1e9))
        capi.reset_tuning()
        print(f"N={n} {t}x{t} x{batch} (operands {2*op_bytes/1e6:.0f} MB, {nsets} sets): " + " | ".join(row), flush=True)
        del Ls, Rs, out, A1, A2, B1, B2
