#!/usr/bin/env python
"""Stand-alone primitives: achieved HBM GB/s of normalize / systematic (scan + search) / gather / column moves at a few
shapes (development tool; algorithmic bytes per element are stated per line)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _env  # noqa: E402  (tools/_env.py: PF_AMD_LIB / PF_* of this process -> the package's explicit switches)

_env.setup()
from pyfilter_amd import ops  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    shapes = [(1 << 20, 1), (1 << 22, 1), (65536, 64), (8192, 1024), (1 << 20, 16)]
    if len(sys.argv) == 3:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    for n, b in shapes:
        g = torch.Generator(device="cuda").manual_seed(0)
        lw = torch.randn(b, n, device="cuda", generator=g)
        u = torch.rand(b, device="cuda", generator=g)
        W, _, _ = ops.normalize_cols(lw.clone())
        x = torch.randn(1, b, n, device="cuda", generator=g)
        idx = ops.systematic_cols(W, u, True)
        e = n * b
        rows = [
            ("normalize (logw -> W, in-place sanitise)", lambda: ops.normalize_cols(lw), 4 + 4 + 4 + 4),   # read, rewrite, read again, write W
            ("systematic(W) -> idx (no cdf where it applies)", lambda: ops.systematic_cols(W, u, True), 4 + 4 + 4 + 4),  # (the contract: read W, write cdf, read cdf, write idx)
            ("multinomial(W) iid draws -> idx", lambda: ops.multinomial_cols(W, 1234), 4 + 4 + 4 + 4),
            ("gather x[idx]", lambda: ops.gather_soa(x, idx), 4 + 4 + 4),
        ]
        def three_launches():
            ops.SYSTEMATIC_CDF_FREE = False
            try:
                return ops.systematic_cols(W, u, True)
            finally:
                ops.SYSTEMATIC_CDF_FREE = True

        rows.insert(2, ("systematic(W) with the cdf (tile sums, scan, search)", three_launches, 4 + 4 + 4 + 4))

        def logw_three():
            ops.SYSTEMATIC_CDF_FREE = False
            try:
                return ops.systematic_cols(lw, u, False)
            finally:
                ops.SYSTEMATIC_CDF_FREE = True

        rows.insert(3, ("systematic(logw) -> idx (no cdf where it applies)", lambda: ops.systematic_cols(lw, u, False), 4 + 4 + 4 + 4))
        rows.insert(4, ("systematic(logw) with the cdf (reduce, scan, search)", logw_three, 4 + 4 + 4 + 4))
        if n * b <= (1 << 24):  # the three-launch path (one offset per grid position always takes it): the A/B of the one-launch resampler
            u_exp = u.unsqueeze(1).expand(b, n).contiguous()
            rows.insert(2, ("systematic(W), three launches (u per position)", lambda: ops.systematic_cols(W, u_exp, True), 4 + 4 + 4 + 4))
        if b > 1:
            perm = torch.randperm(b, device="cuda")
            xx = x.permute(2, 1, 0)
            rows.append(("gather_filters (whole columns)", lambda: ops.gather_filters(xx, perm), 8))
        for name, fn, bpe in rows:
            us = timed(fn)
            print(f"N={n:8d} B={b:5d}  {name:45s} {us:8.1f} us  {bpe * e / us / 1e3:8.1f} GB/s ({bpe} B/elem)  {100 * bpe * e / us / 1e3 / 8000:5.1f} % of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
