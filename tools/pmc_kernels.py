#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (dev tool).  usage: pmc_kernels.py FILE [SUBSTR]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else ""
def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    if want in k:
        n = len(disp[k])
        print(k, "dispatches", n, {a: round(b / n) for a, b in sorted(v.items())})
