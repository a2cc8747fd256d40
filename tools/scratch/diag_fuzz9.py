"""fuzz case 9 of seed 11 (rw2d_theta sisr bootstrap N=2048 B=9 T=7 oes=5): where do the NaN moments of filter 7 start?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu_ref
from oracle.cases import build_spec, simulate
from pyfilter_amd.filters.schedule import expand
from pyfilter_amd.hints import HINTS
from tests.helpers import build_filter_from_case

n, b, t_len, oes, seed = int(os.environ.get("N", 2048)), 9, 7, 5, 403092
case = dict(name="fuzz", model="rw2d_theta", filter="sisr", proposal="bootstrap", N=n, B=b, T=t_len, ess_threshold=0.5, seed=seed, observe_every_step=oes)
spec = build_spec(case, torch.float64)
gen = torch.Generator().manual_seed(seed)
d = (spec.dim,)
moves = expand(0, t_len, oes).moves
g = dict(z_tape=torch.randn((moves, n, b) + d, generator=gen, dtype=torch.float32), u_tape=torch.rand(moves, b, generator=gen, dtype=torch.float32),
         z0=torch.randn((n, b) + d, generator=gen, dtype=torch.float32))
y = simulate(case, spec, torch.float64)
y[3] = float("nan"); y[6] = float("nan")
x0 = cpu_ref.M.initial_sample(spec, g["z0"].double())
ref = cpu_ref.batch_filter(spec, "sisr", "bootstrap", y, x0, g["z_tape"].double(), g["u_tape"].double(), ess_threshold=0.5)
for route in (0, 1):
    HINTS.route = route
    for direct in (False, True):
        HINTS.direct = direct
        filt = build_filter_from_case(case, g, torch.float64, "cuda")
        res = filt.batch_filter(y.cuda(), bar=False)
        m = res.filter_means.cpu()
        bad = torch.isnan(m).any(-1)
        print("route", route, "direct", direct, "nan rows x filters:", bad.nonzero().tolist()[:10], " max |dm| (nan->0)",
              float(torch.nan_to_num(m - ref["filter_means"]).abs().max()), " ll diff", float(torch.nan_to_num(res.loglikelihood.cpu() - ref["loglikelihood"]).abs().max()),
              "ll nan", torch.isnan(res.loglikelihood).any().item())
        rows = filt._last_run.get("rows")
        if rows is not None:
            print("   reported rows shape", tuple(rows[0].shape))
        full = filt._last_run.get("means_full") if isinstance(filt._last_run, dict) else None
        pm = filt._last_run["plan"].means.cpu()  # (moves + 1, B, D): every move's row
        nanrows = torch.isnan(pm).any(-1).nonzero().tolist()
        print("   all-move rows with NaN (row, filter):", nanrows[:12], " plan.means shape", tuple(pm.shape))
