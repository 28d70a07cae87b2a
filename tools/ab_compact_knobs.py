#!/usr/bin/env python3
"""A/B of the compaction kernel's knobs on the 4096 x 1024-term cases (dev tool)."""
import os, sys, subprocess
for env in ({}, {"CSGN_COMPACT_STAGGER_US": "8"}, {"CSGN_COMPACT_STAGGER_US": "16"}, {"CSGN_COMPACT_STAGGER_US": "24"},
            {"CSGN_COMPACT_GRID": "768"}, {"CSGN_COMPACT_GRID": "1024"}):
    print("==", env or "defaults", flush=True)
    out = subprocess.run([sys.executable, "tools/bench_compact.py", "--only", "4096 x 1024 terms"], env=dict(os.environ, **env),
                         capture_output=True, text=True).stdout
    print("\n".join(l[:150] for l in out.splitlines() if l.startswith("compact")), flush=True)
