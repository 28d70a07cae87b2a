#!/usr/bin/env python3
"""A/B of the XCD block order of the CSR kernels (knob ragged_xcd_group) and of the wave-cooperative kernel
(ragged_coop_xcd_group): contiguous eighths (0) against groups of G blocks per XCD turn.  Dev tool."""
import os, sys, subprocess
for name, val in (("RAGGED_XCD_GROUP", "0"), ("RAGGED_XCD_GROUP", "64"), ("RAGGED_COOP_XCD_GROUP", "16"), ("RAGGED_COOP_XCD_GROUP", "64")):
    env = dict(os.environ, **{"CSGN_" + name: val}, SHORT="1", CHUNKS="0")
    print(f"== {name.lower()} = {val}", flush=True)
    out = subprocess.run([sys.executable, "tools/prof_ragged_ops.py"], env=env, capture_output=True, text=True).stdout
    print("\n".join(l for l in out.splitlines() if "add_ragged " in l or "async" in l), flush=True)
    out = subprocess.run([sys.executable, "tools/bench_ragged.py"], env=env, capture_output=True, text=True).stdout
    print("\n".join(l[:150] for l in out.splitlines() if "kernel only, GB/s" in l or "add_ragged" in l and "kernels only" in l), flush=True)
