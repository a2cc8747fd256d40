#!/bin/bash
# HBM traffic of the cluster kernel (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes), per step
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/clu_traffic; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p_$c -o p -- env PF_CLUSTER=1 KB_T=200 KB_NO_TIMED=1 python $OLDPWD/tools/kbench.py ${1:-apf_lgo_128x8192} > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "cluster" in r["Kernel_Name"]:
            k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        print(f"{k:12s} per dispatch (200 steps) {v / max(n,1):14.1f} KiB-units   -> per step {v / max(n,1) / 200:10.1f}   dispatches {n}")
PY
