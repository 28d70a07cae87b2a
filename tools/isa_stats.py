#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc -S output (dev tool).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icsgn_amd/csrc -S --cuda-device-only -o /tmp/k.s csgn_amd/csrc/csgn_X.hip
    python tools/isa_stats.py /tmp/k.s <substring of the mangled kernel name> [--blocks]
"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
end = next(i for i in range(start, len(text)) if text[i].strip().startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in text[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append((name, cur))
        name, cur = t.split(":")[0], []
        continue
    if t.startswith("."):
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
allops = [o for _, b in blocks for o in b]
c = Counter(allops)
cls = lambda o: "valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "mem"
tot = Counter(cls(o) for o in allops)
print(f"{text[start].split(':')[0][:90]}\n total {len(allops)}  " + "  ".join(f"{k} {v}" for k, v in tot.items()))
print(" top:", ", ".join(f"{k} {v}" for k, v in c.most_common(18)))
if "--blocks" in sys.argv:
    for n, b in blocks:
        if len(b) >= 20:
            bc = Counter(cls(o) for o in b)
            print(f"  {n:<12} {len(b):5d}  valu {bc['valu']:5d} salu {bc['salu']:4d} mem {bc['mem']:4d}  top: " +
                  ", ".join(f"{k} {v}" for k, v in Counter(b).most_common(6)))
