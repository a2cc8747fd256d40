import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_filters_gpu import _full_size_case
from tests.helpers import build_filter_from_case
from oracle import cpu_ref
os.environ["PF_PIPELINE"] = sys.argv[1]
n = int(sys.argv[2]); TT = int(sys.argv[3]); KEEP = int(sys.argv[4])
MODEL, FILT, PROP = (sys.argv[5:8] + ["sine", "apf", "bootstrap"])[:3] if len(sys.argv) >= 8 else ("sine", "apf", "bootstrap")
case, spec, g, y = _full_size_case(MODEL, FILT, PROP, n, 1, TT, seed=900 + n % 97)
y = y[:KEEP]; g["z_tape"] = g["z_tape"][:KEEP]; g["u_tape"] = g["u_tape"][:KEEP]
x0 = cpu_ref.M.initial_sample(spec, g["z0"].double())
ref = cpu_ref.batch_filter(spec, FILT, PROP, y, x0, g["z_tape"].double(), g["u_tape"].double(), ess_threshold=0.9, record_steps=True)
filt = build_filter_from_case(case, g, torch.float64, "cuda")
res = filt.batch_filter(y.cuda(), bar=False)
a = res.latest_state.previous_indices.cpu()[:, 0]; r = ref["prev_inds"][:, 0]
bad = (a != r).nonzero().reshape(-1)
print("mismatches", bad.numel(), "first", bad[:10].tolist(), "last", bad[-5:].tolist())
if bad.numel():
    i = bad[0].item()
    print("around first:", a[max(0,i-3):i+5].tolist(), r[max(0,i-3):i+5].tolist())
    d = (a - r)[bad]
    print("diff stats: min", d.min().item(), "max", d.max().item(), "unique small", torch.unique(d)[:10].tolist())
    blocks = torch.unique(bad // 1024)
    print("position tiles affected:", blocks.numel(), blocks[:20].tolist())

a2 = res.latest_state.previous_indices.cpu(); print("shape", a2.shape)
