#!/usr/bin/env python3
"""Print a rocprofv3 *kernel_stats.csv compactly (dev tool): python tools/kstats.py <file.csv>"""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"].replace("csgn::(anonymous namespace)::", "").replace("void ", ""))[:44]
    print("%-44s calls %5s  avg %9.1f us  min %8.1f  max %8.1f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
