#!/usr/bin/env python3
"""tools/bench_compact.py against two builds of libcsgn_hip.so, alternating on one box (dev tool).
usage: ab_compact.py <libA> <libB> [bench_compact arguments]"""
import os, subprocess, sys
for rnd in range(2):
    for lib in sys.argv[1:3]:
        env = dict(os.environ, CSGN_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "tools/bench_compact.py"] + sys.argv[3:], env=env, capture_output=True, text=True)
        for l in r.stdout.splitlines():
            if l.startswith("compact"):
                print(f"{os.path.basename(lib):<18} {l[:150]}", flush=True)
        if r.returncode:
            print(r.stderr[-400:])
