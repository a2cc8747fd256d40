#!/usr/bin/env python
"""Development: where k_chunk_search (pf_systematic without a cdf) spends its time - cycle stamps of the middle workgroup of column 0,
from the instrumented build: tools/build_variant.sh dev "-DPF_DEVTOOLS" main;  PF_AMD_LIB=.../libpfamd_dev.so python tools/chunk_search_stages.py [N] [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402

_env.setup()
from pyfilter_amd import _lib as L  # noqa: E402

if os.environ.get("PF_AMD_LIB"):
    L.LIB_PATH = os.environ["PF_AMD_LIB"]
from pyfilter_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = L.load()
    off = C.c_size_t(0)
    lib.pf_debug_offset.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]
    L.check(lib.pf_debug_offset(n, b, C.byref(off)), "pf_debug_offset")
    g = torch.Generator(device="cuda").manual_seed(0)
    W, _, _ = ops.normalize_cols(torch.randn(b, n, device="cuda", generator=g))
    u = torch.rand(b, device="cuda", generator=g)
    ws = L.workspace(n, b, W.device)
    names = ["tile sums -> prefix table", "find the first chunk", "stage 5 chunks (loads, scans)", "counts + scatter", "heads -> ancestors, store"]
    acc = [0.0] * 5
    reps = 50
    for _ in range(reps):
        ops.systematic_cols(W, u, True)
        torch.cuda.synchronize()
        st = ws[off.value:off.value + 48].view(torch.int64).cpu().tolist()
        for i in range(5):
            acc[i] += (st[i + 1] - st[i]) / reps
    tot = sum(acc)
    print(f"k_chunk_search, N = {n}, B = {b}: middle workgroup of column 0, {tot:.0f} clock64 ticks")
    for nm, a in zip(names, acc):
        print(f"   {nm:34s} {a:8.0f} ticks  {100 * a / tot:5.1f} %")


if __name__ == "__main__":
    main()
