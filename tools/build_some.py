#!/usr/bin/env python
"""Development: recompile SOME translation units of libpfamd.so (by object name, e.g. ``pf_main pf_clu_f32``) with the flags
``__graft_entry__.build_units`` gives them, and relink - minutes less than ``build(force=True)`` when an edit touches one header.
The digest compiled in is the tree's, so ``binary_matches_sources()`` holds only if every unit the edit reaches was named."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    want = set(sys.argv[1:])
    objdir = os.path.join(ROOT, "build", "obj")
    units = ge.build_units(objdir)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950"] + ge.COMPRESS + ["-O3", "-std=c++17", "-fPIC", "-c", os.path.join(ge.CSRC, "pf_kernels.hip"),
              f'-DPF_SOURCE_SHA256="{ge.source_digest()}"']
    procs = [subprocess.Popen(common + flags + ["-o", obj], cwd=ge.CSRC) for flags, obj in units
             if os.path.basename(obj)[:-2] in want]
    assert len(procs) == len(want), "unknown unit name"
    if any(p.wait() for p in procs):
        raise SystemExit("compile failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950"] + ge.COMPRESS + ["-shared", "-fPIC"] + [obj for _, obj in units] + ["-o", ge.LIB], cwd=ge.CSRC)
    print("relinked; matches sources:", ge.binary_matches_sources())


if __name__ == "__main__":
    main()
