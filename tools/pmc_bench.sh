#!/bin/bash
# SQ counters of the step kernel on the bench workload (per-wave averages): tools/pmc_bench.sh <workload> <counter...>
W=${1:-apf_lgo_1m}; shift
export TMPDIR=/tmp
OUT=/tmp/pmcb_$$
(cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -o p -- python $OLDPWD/bench.py --_inner --workload $W --T 40 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic > /dev/null 2>&1)
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused_step" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    n=len(next(iter(d.values())))
    if n < 5: continue
    w=sum(d.get("SQ_WAVES",[1]))/max(1,len(d.get("SQ_WAVES",[1])))
    print(k, "launches", n, "waves", w)
    for c,v in sorted(d.items()):
        m=sum(v)/len(v)
        print(f"   {c:28s} {m:14.1f}  per wave {m/w:10.1f}")
PY
rm -rf $OUT
