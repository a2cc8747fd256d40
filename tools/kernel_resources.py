#!/usr/bin/env python
"""Development (no GPU): LDS / scratch / VGPRs of every kernel of one translation unit, from the kernel descriptors of a -save-temps
build - for the working tree, and against another revision when one is named: what changed, what is new, what is gone.
    python tools/kernel_resources.py [unit] [rev]        unit: pf_main (default), pf_clu_f32, ... (tools/build_some.py's names)
Round 6: indexing a small per-thread array by a run-time value had the compiler promote it to LDS - 18.5 KB in k_scan, for every
caller - and nothing but this listing shows it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def descriptors(tree: str, flags, work: str):
    src = os.path.join(tree, "pyfilter_amd", "csrc", "pf_kernels.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, '-DPF_SOURCE_SHA256="x"', "-save-temps",
                           "-o", os.path.join(work, "x.o")] + list(flags), cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(work, "pf_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        num = lambda key: int(re.search(r"\.amdhsa_" + key + r"\s+(\d+)", m.group(2)).group(1))  # noqa: E731
        out[m.group(1)] = (num("group_segment_fixed_size"), num("private_segment_fixed_size"), num("next_free_vgpr"))
    return out


def demangle(names):
    try:
        res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, res))
    except OSError:
        return {n: n for n in names}


def main():
    unit = sys.argv[1] if len(sys.argv) > 1 else "pf_main"
    rev = sys.argv[2] if len(sys.argv) > 2 else None
    flags = next(f for f, obj in ge.build_units("/tmp") if os.path.basename(obj)[:-2] == unit)
    with tempfile.TemporaryDirectory() as w1:
        new = descriptors(ROOT, flags, w1)
    old = None
    if rev:
        with tempfile.TemporaryDirectory() as tree, tempfile.TemporaryDirectory() as w2:
            for path in subprocess.check_output(["git", "ls-tree", "-r", "--name-only", rev, "pyfilter_amd/csrc", "include"], cwd=ROOT, text=True).split():
                os.makedirs(os.path.dirname(os.path.join(tree, path)), exist_ok=True)
                with open(os.path.join(tree, path), "wb") as f:
                    f.write(subprocess.check_output(["git", "show", f"{rev}:{path}"], cwd=ROOT))
            old = descriptors(tree, flags, w2)
    names = demangle(sorted(set(new) | set(old or {})))
    fmt = lambda r: f"LDS {r[0]:6d} B  scratch {r[1]:4d} B  VGPRs {r[2]:3d}"  # noqa: E731
    if old is None:
        for k in sorted(new, key=lambda k: names[k]):
            print(f"{fmt(new[k])}  {names[k][:140]}")
        return
    for k in sorted(set(new) & set(old), key=lambda k: names[k]):
        if new[k] != old[k]:
            print(f"CHANGED  {fmt(old[k])}  ->  {fmt(new[k])}  {names[k][:120]}")
    for k in sorted(set(new) - set(old), key=lambda k: names[k]):
        print(f"NEW      {fmt(new[k])}  {names[k][:140]}")
    for k in sorted(set(old) - set(new), key=lambda k: names[k]):
        print(f"GONE     {fmt(old[k])}  {names[k][:140]}")
    print(f"{len(new)} kernels in {unit} now, {len(old)} at {rev}; scratch users now: {sum(1 for r in new.values() if r[1])}")


if __name__ == "__main__":
    main()
