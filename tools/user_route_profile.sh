#!/bin/bash
# Development tool: which kernels a move of a lambda-defined model costs (rocprofv3 --kernel-trace --stats of tools/kbench.py
# on the user_* configurations) and the us per step of the user routes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in user_sisr_boot_1m user_apf_lgo_1m; do
  out=/tmp/up_$c; rm -rf $out
  KB_T=50 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/kbench.py $c > /tmp/up.log 2>&1
  echo "# $c (T = 50; kernel, calls, average us, share)"
  python - "$(find $out -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:8.2f} {float(r['Percentage']):6.2f}")
PY
done
KB_T=100 python tools/kbench.py user_apf_lgo_1m user_sisr_boot_1m user_apf_lgo_1024x512 2>&1 | grep us/step
