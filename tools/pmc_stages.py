#!/usr/bin/env python
"""Per-stage instruction profile of the fused step kernel (development tool).  For every PF_DEBUG_CUT value it runs a
short filter under ``rocprofv3 --pmc`` and prints the per-wave averages of the requested SQ counters, so that
differences between consecutive cuts give each stage's cost.
Usage: python tools/pmc_stages.py [config] [counter ...]    (cuts: 1 4 2 5 3 6 0 in pipeline order)"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = [1, 4, 2, 5, 3, 6, 0]
NAMES = {1: "entry", 4: "loads issued + normals", 2: "table map + search", 5: "gather + propagate + weight + stores",
         3: "push_round (thread partials)", 6: "finish (block reductions)", 0: "tile-local scan (full kernel)"}


def collect(config, counters, cut, kernel="k_fused_step"):
    out = tempfile.mkdtemp(prefix="pf_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "tools", "kbench.py"), config]
    T = 6
    env = dict(os.environ, TMPDIR="/tmp", PF_NO_GRAPH="1", PF_DEBUG_CUT=str(cut), PF_DEBUG_CUT_AT_END="1", KB_T=str(T),
               KB_NO_TIMED="1")
    subprocess.run(cmd, cwd="/tmp", env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                rows.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    shutil.rmtree(out, ignore_errors=True)
    res = {}
    for name, lst in rows.items():
        lst.sort()
        # the cut applies to the last-but-one step of every run (valid input state): launches T-2, 2T-2, ...
        sel = [v for i, (_, v) in enumerate(lst) if i % T == T - 2]
        res[name] = sum(sel) / max(len(sel), 1)
    return res


def build_dev():
    """libpfamd_dev.so: the float / scalar-state / VEC = 4 translation unit rebuilt with -DPF_DEVTOOLS (cycle stamps + stage
    cuts compiled in), linked with the production objects of the other units (build/obj, from __graft_entry__.build())."""
    out = os.path.join(ROOT, "pyfilter_amd", "libpfamd_dev.so")
    csrc = os.path.join(ROOT, "pyfilter_amd", "csrc")
    src = os.path.join(csrc, "pf_kernels.hip")
    objdir = os.path.join(ROOT, "build", "obj")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)):
        hipcc = "/opt/rocm/bin/hipcc"
        dev_obj = os.path.join(objdir, "pf_f32d1_v4_dev.o")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-DPF_DEVTOOLS",
                               "-DPF_TU_F32D1_ONLY", "-DPF_TU_VEC=4", "-o", dev_obj], cwd=csrc)
        others = [os.path.join(objdir, f) for f in ("pf_main.o", "pf_f32d1_v1_m0.o", "pf_f32d1_v1_m1.o", "pf_f32dn_m0.o",
                                                    "pf_f32dn_m1.o", "pf_f64_m0.o", "pf_f64_m1.o", "pf_col_f32.o", "pf_col_f64.o")]
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", dev_obj] + others + ["-o", out], cwd=csrc)
    return out


def main():
    if sys.argv[1:2] == ["--build"]:
        print(build_dev())
        return
    os.environ["PF_AMD_LIB"] = build_dev()
    config = sys.argv[1] if len(sys.argv) > 1 else "apf_lgo_1m"
    counters = sys.argv[2:] or ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES"]
    if "SQ_WAVES" not in counters:
        counters.append("SQ_WAVES")
    prev = None
    for cut in ORDER:
        r = collect(config, counters, cut)
        w = r.get("SQ_WAVES", 0) or 1
        per = {k: r[k] / w for k in r if k != "SQ_WAVES"}
        delta = {k: per[k] - (prev or {}).get(k, 0.0) for k in per}
        print(f"cut {cut} [{NAMES[cut]:40s}] waves {w:8.0f} " + " ".join(f"{k}={per[k]:8.1f} (+{delta[k]:7.1f})" for k in sorted(per)), flush=True)
        prev = per


if __name__ == "__main__":
    main()
