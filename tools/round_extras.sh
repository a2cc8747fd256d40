#!/bin/bash
# The rest of a round's measurement set next to tools/profile_round.sh (run through gpurun from the repo root): stand-alone
# primitives (bench + per-kernel rocprofv3 averages), online-move latency, the SMC^2 step() / fit() timings and the kernel
# statistics of the step() loop, the fuzz sweeps against the oracle, the one-GPU scaling model and the N > 1 rehearsal.
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
python tools/prim_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/prim_bench.txt
bash tools/prim_profile.sh 2>/dev/null > $OUT/primitives_kernels.txt
python tools/step_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/step_latency.txt
python tools/smc2_small.py 128 8192 500 2>&1 | grep -v amdgpu.ids > $OUT/smc2_small_128x8192.txt
python tools/smc2_small.py 1000 400 500 2>&1 | grep -v amdgpu.ids > $OUT/smc2_small_1000x400.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_smc2_step -o p -- python $GRAFT_REPO_ROOT/tools/smc2_step_profile.py 128 8192 500 > $OUT/prof_smc2_step.log 2>&1)
python tools/fuzz_parity.py 200 61 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/fuzz_parity.txt
FUZZ_CLUSTER=1 python tools/fuzz_parity.py 300 62 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/fuzz_parity.txt
python tools/smc2_scaling_model.py 2>&1 | grep -v amdgpu.ids > $OUT/smc2_scaling_model.txt
python tools/scale_preflight.py 2>&1 | grep -v "amdgpu.ids\|c10d\|Gloo" > $OUT/scale_preflight.txt
ls $OUT
