#!/usr/bin/env python
"""Per-wave averages of SQ counters for one kernel of a kbench configuration (development tool).
Usage: python tools/pmc_one.py <config> <kernel substring> <counter> [counter ...]   (env PF_AMD_LIB selects the library)"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    config, kernel, counters = sys.argv[1], sys.argv[2], sys.argv[3:]
    if "SQ_WAVES" not in counters:
        counters.append("SQ_WAVES")
    out = tempfile.mkdtemp(prefix="pf_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "tools", "kbench.py"), config]
    env = dict(os.environ, TMPDIR="/tmp", PF_NO_GRAPH="1", KB_T=os.environ.get("KB_T", "20"), KB_NO_TIMED="1")
    subprocess.run(cmd, cwd="/tmp", env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                v = acc.setdefault(row["Counter_Name"], [0.0, 0])
                v[0] += float(row["Counter_Value"])
                v[1] += 1
    shutil.rmtree(out, ignore_errors=True)
    r = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
    w = r.get("SQ_WAVES", 0) or 1
    print(f"{config} {kernel}: launches {max(v[1] for v in acc.values()) if acc else 0} waves/launch {w:.0f} | " +
          " ".join(f"{k}={r[k] / w:.1f}" for k in sorted(r) if k != "SQ_WAVES"), flush=True)


if __name__ == "__main__":
    main()
