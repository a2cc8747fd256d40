#!/bin/bash
# PMC counters of the cluster kernel at 128 x 8192 (dev): VALU / SALU / LDS instructions and wave cycles per wave
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/clu_pmc; mkdir -p $OUT
cd /tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p_$(echo $c | cut -d' ' -f1) -o p -- env PF_CLUSTER=1 KB_T=200 KB_NO_TIMED=1 python $OLDPWD/tools/kbench.py $1 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "cluster" in r["Kernel_Name"]:
            k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        print(f"{k:24s} per dispatch {v / max(n,1):14.1f}   dispatches {n}")
PY
