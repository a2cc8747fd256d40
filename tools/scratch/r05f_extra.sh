#!/bin/bash
# round 5 final extras: cluster-route shapes (default / per-step / members spread over the XCDs), the SMC^2 scaling model, a kernel
# trace of an SMC^2 fit at the per-rank shape of an 8-GPU config 5
OUT=$PWD/gpurun_out/r05f; mkdir -p $OUT; export TMPDIR=/tmp
S="apf_lgo_128x8192 sisr_boot_128x8192 apf_boot_128x8192 apf_lgo_64x16384 apf_lgo_256x4096 apf_lgo_256x8192 apf_lgo_16x8192"
for mode in "PF_NO_CLUSTER=1" "PF_CLUSTER=1"; do echo "== $mode"; env $mode KB_T=250 python tools/kbench.py $S 2>&1 | grep us/step; done > $OUT/kbench_cluster.txt
python - > $OUT/kbench_spread.txt 2>&1 <<'PY'
import os, sys, time, torch
sys.path.insert(0, "tools"); sys.argv = ["kbench"]
import kbench
from pyfilter_amd.hints import HINTS
for route in (4, 5):
    HINTS.route = route
    for name in ("apf_lgo_128x8192", "apf_lgo_64x16384"):
        f, o = kbench.make(*kbench.parse_shape(name))
        y = (0.3 * torch.randn(250)).cumsum(0).cuda()
        for _ in range(2): f.batch_filter(y, bar=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        print(f"route {route} ({'members on one XCD' if route == 4 else 'members spread over the XCDs, sc1 exchange'}) {name}: {1e6 * (time.perf_counter() - t0) / 3 / 250:.2f} us per step")
PY
python tools/smc2_scaling_model.py > $OUT/smc2_scaling_model.txt 2>&1
PF_NO_CLUSTER=1 python tools/smc2_scaling_model.py > $OUT/smc2_scaling_model_per_step.txt 2>&1
python tools/smc2_small.py 128 8192 500 > $OUT/smc2_small_128x8192.txt 2>&1
python tools/smc2_small.py > $OUT/smc2_small_1000x400.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_smc2_128 -o p -- python $OLDPWD/tools/smc2_small.py 128 8192 500 > $OUT/prof_smc2_128.log 2>&1)
find $OUT/prof_smc2_128 -name "*kernel_trace.csv" -delete
