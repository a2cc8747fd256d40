#!/bin/bash
# Collects this round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   bench lines of every single-GPU workload (with cpu_baseline and PMC traffic) and rocprofv3 --kernel-trace --stats of the
#   same commands; everything lands under gpurun_out/<tag>/ and is condensed into profiles/ by tools/condense_profiles.py.
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in apf_lgo_1m sv_batch lorenz_mn smc2_shard; do
  STEPS=3; [ $w = apf_lgo_1m ] && STEPS=5
  python bench.py --workload $w --steps $STEPS --warmup 2 2>/dev/null | tail -1 > $OUT/bench_$w.json
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o p -- python $OLDPWD/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $OUT/prof_$w.log 2>&1)
done
python bench.py --workload smc2 --steps 1 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_smc2.json
find $OUT -name "*kernel_stats.csv" | head
