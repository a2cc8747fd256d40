#!/usr/bin/env python
"""Top kernels of a rocprofv3 ``*kernel_stats.csv`` found under a directory: python tools/scratch/kernel_stats_top.py DIR [n]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].split("(")[0].replace("void ", "")[:90]
    print(f"{name:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.2f} us  {r['Percentage']:>6s} %")
