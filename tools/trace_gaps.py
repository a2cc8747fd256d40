#!/usr/bin/env python
"""Gaps between consecutive kernels of a rocprofv3 ``*_kernel_trace.csv`` (last ``n`` dispatches): name, duration, idle time
since the previous kernel's end - where a chain of short dependent launches (a captured user-callable run) loses its time."""
import csv
import sys


def main(path, n=24):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("at::native::", "")[:70]
        print(f"{name:70s} dur {1e-3 * (e - s):7.2f} us  gap {0.0 if prev is None else 1e-3 * (s - prev):7.2f} us")
        prev = e
    t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    print(f"span {1e-3 * (t1 - t0):.1f} us over {len(rows)} kernels")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24)
