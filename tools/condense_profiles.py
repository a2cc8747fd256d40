#!/usr/bin/env python
"""gpurun_out/<tag>/ (tools/profile_round.sh) -> profiles/<tag>_*: one kernel-stats table per workload (rocprofv3
--kernel-trace --stats: kernel, calls, total us, average us, share) and the bench lines."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
note = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
lines = []
for w in ("apf_lgo_1m", "sv_batch", "lorenz_mn", "smc2_shard"):
    files = glob.glob(os.path.join(src, f"prof_{w}", "**", "*kernel_stats.csv"), recursive=True)
    if files:
        rows = list(csv.DictReader(open(files[0])))
        with open(os.path.join(dst, f"{tag}_bench_{w}_kernel_stats.txt"), "w") as f:
            f.write(f"# {tag} {note}: rocprofv3 --kernel-trace --stats -- python bench.py --workload {w} --steps 2 --warmup 1 "
                    f"--no-cpu-baseline --no-traffic   (fp32)\n# kernel, calls, total_us, avg_us, pct\n")
            for r in rows[:12]:
                f.write(f"{r['Name'][:150]}\t{r['Calls']}\t{float(r['TotalDurationNs']) / 1e3:.1f}\t"
                        f"{float(r['AverageNs']) / 1e3:.3f}\t{float(r['Percentage']):.2f}\n")
    b = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(b) and os.path.getsize(b):
        lines.append(open(b).read().strip())
b = os.path.join(src, "bench_smc2.json")
if os.path.exists(b) and os.path.getsize(b):
    lines.append(open(b).read().strip())
with open(os.path.join(dst, f"{tag}_bench.json.log"), "w") as f:
    f.write(f"# {tag} {note}: python bench.py --workload <w>  (one JSON line per workload; the first is the headline)\n")
    f.write("\n".join(lines) + "\n")
for ln in lines:
    d = json.loads(ln)
    r = d.get("roofline") or {}
    print(d["config"]["workload"][:44], f"{d['value']:.3e}", r.get("kernel_us"), f"frac={r.get('frac')}", f"as_built={(r.get('as_built') or {}).get('frac')}",
          "traffic", r.get("traffic"), (r.get("bytes_per_launch") or {}).get("as_built"))
