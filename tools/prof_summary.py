#!/usr/bin/env python
"""Dumps the per-kernel summary (calls, total / average duration in us, share) of a rocprofv3 ``*_results.db`` as a
small text table for ``profiles/`` (the .db itself is scratch under gpurun_out/)."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        if note:
            f.write(f"# {note}\n")
        f.write("# rocprofv3 --kernel-trace --stats : kernel, calls, total_us, avg_us, pct\n")
        for name, calls, total, avg, pct in rows:
            short = name if len(name) < 120 else name[:117] + "..."
            f.write(f"{short}\t{calls}\t{total:.1f}\t{avg:.3f}\t{pct:.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
