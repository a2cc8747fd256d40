#!/bin/bash
# Development tool: rocprofv3 --kernel-trace --stats of the stand-alone primitives (tools/prim_bench.py) at the BASELINE
# shapes; prints, per shape, the average duration of every pf:: kernel and - for the resampling scan - 8 N B / t against
# 8 TB/s ("resampling scan kernel alone: 8 B / particle").   Usage: tools/prim_profile.sh > out.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in "1048576 1" "65536 64" "4194304 1" "8192 128" "8192 1024"; do
  set -- $shape
  out=/tmp/prim_prof_$1_$2; rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/prim_bench.py $1 $2 > /tmp/prim_prof.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "# N = $1, B = $2  (prim_bench: 55 calls of each primitive)"
  python - "$f" $1 $2 <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pf::" in r["Name"]]
n, b = int(sys.argv[2]), int(sys.argv[3])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].split("(")[0].replace("void ", "")
    avg = float(r["AverageNs"]) / 1e3
    extra = ""
    if "k_scan" in name:
        extra = f"   scan roofline: 8 B x {n * b} / {avg:.2f} us = {8 * n * b / avg / 1e3:.0f} GB/s = {100 * 8 * n * b / avg / 1e3 / 8000:.1f} % of 8 TB/s"
    print(f"{name:60s} calls {r['Calls']:>4s}  avg {avg:8.2f} us{extra}")
PY
done
