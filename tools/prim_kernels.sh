#!/bin/bash
# per-kernel rocprofv3 averages of tools/prim_bench.py at one shape: tools/prim_kernels.sh N B
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_prim
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_prim -o p -- python $GRAFT_REPO_ROOT/tools/prim_bench.py $1 $2 > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_prim/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "pf::" in r["Name"]:
        print(f'{r["Name"][:72]:72s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:8.2f} us')
PY
