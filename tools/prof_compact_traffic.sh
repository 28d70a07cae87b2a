#!/bin/bash
# HBM traffic of the compaction kernels (PMC WRITE_SIZE and FETCH_SIZE in separate passes, one call per case:
# bench_compact.py --once) against the algorithmic bytes.  usage: bash tools/prof_compact_traffic.sh OUTDIR
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/c_$C -o c -- python3 tools/bench_compact.py --once --only "terms, " > $OUT/compact_pmc_$C.log 2>&1
  echo "$C rc=$?"
done
python3 - $OUT <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
cases = []          # (name, in terms, out terms, unit bytes) in run order
for ln in open(out + "/compact_pmc_WRITE_SIZE.log"):
    m = re.match(r"(compact .*?N=(\d+))\s+[\d.]+ ms\s+in\s+(\d+) out\s+(\d+) terms", ln)
    if m:
        cases.append((m.group(1).strip(), int(m.group(3)), int(m.group(4)), (int(m.group(2)) + 63) // 64 * 8))
per_case = {}       # counter -> list (per case) of {kernel: KiB}
for C in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob(out + "/c_" + C + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == C]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # a case = the dispatches from one k_cg_count to the next; --once makes two calls per case (set-up + timed): keep the last
    calls, cur = [], None
    for r in rows:
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("csgn::(anonymous namespace)::", "").replace("void ", "")).split("<")[0]
        if k == "k_cg_count":
            cur = collections.OrderedDict()
            calls.append(cur)
        if cur is not None and k.startswith(("k_c",)):
            cur[k] = cur.get(k, 0.0) + float(r["Counter_Value"])
    per_case[C] = [calls[2 * i + 1] for i in range(len(calls) // 2)]
lines = ["per call (KiB counters; FETCH_SIZE x2 on gfx950), MB written / MB read per kernel, and the call against its algorithmic bytes"]
for i, (name, tin, tout, tb) in enumerate(cases):
    if i >= len(per_case["WRITE_SIZE"]) or i >= len(per_case["FETCH_SIZE"]):
        break
    w, r = per_case["WRITE_SIZE"][i], per_case["FETCH_SIZE"][i]
    lines.append(name)
    tw = tr = 0.0
    for k in w:
        wmb, rmb = w[k] * 1024 / 1e6, r.get(k, 0.0) * 2 * 1024 / 1e6
        tw += wmb; tr += rmb
        lines.append("    %-16s written %9.1f MB   read %9.1f MB" % (k, wmb, rmb))
    alg = (tin + tout) * tb / 1e6
    lines.append("    total written %.1f + read %.1f = %.1f MB; algorithmic %.1f MB (in %.1f + out %.1f): %.3f x" % (tw, tr, tw + tr, alg, tin * tb / 1e6, tout * tb / 1e6, (tw + tr) / alg))
open(out + "/compact_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/c_WRITE_SIZE $OUT/c_FETCH_SIZE
