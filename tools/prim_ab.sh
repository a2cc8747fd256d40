#!/bin/bash
# per-kernel rocprofv3 averages of the stand-alone primitives for several variant libraries (tools/build_variant.sh):
# tools/prim_ab.sh name1 name2 ...   (shapes: 2^20 x 1, 65536 x 64, 4M x 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  for shape in "1048576 1" "65536 64" "4194304 1"; do
    set -- $shape
    out=/tmp/prim_ab_${v}_$1_$2; rm -rf $out
    PF_AMD_LIB=$GRAFT_REPO_ROOT/pyfilter_amd/libpfamd_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/prim_bench.py $1 $2 > /tmp/prim_ab.log 2>&1
    grep "systematic(W)" /tmp/prim_ab.log | sed "s/^/[$v] /"
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    python - "$f" "$v" $1 $2 <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pf::k_scan" in r["Name"] or "pf::k_search" in r["Name"] or "pf::k_tile_sum" in r["Name"]]
print("   ", sys.argv[2], sys.argv[3], "x", sys.argv[4], "  ".join(f"{r['Name'].split('(')[0].replace('void pf::','')[:24]} {float(r['AverageNs'])/1e3:.2f}" for r in sorted(rows, key=lambda r: r["Name"])))
PY
  done
done
