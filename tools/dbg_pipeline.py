import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_filters_gpu import _full_size_case
from tests.helpers import build_filter_from_case
from oracle import cpu_ref

model, filt_name, prop, n, b, t_len = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
dt = torch.float64 if (len(sys.argv) < 8 or sys.argv[7] == "f64") else torch.float32
case, spec, g, y = _full_size_case(model, filt_name, prop, n, b, t_len, seed=900 + n % 97)
x0 = cpu_ref.M.initial_sample(spec, g["z0"].double())
ref = cpu_ref.batch_filter(spec, filt_name, prop, y, x0, g["z_tape"].double(), g["u_tape"].double(), ess_threshold=0.9, record_steps=True)
out = {}
for pipe in ("scan", "plan"):
    os.environ["PF_PIPELINE"] = pipe
    filt = build_filter_from_case(case, g, dt, "cuda")
    res = filt.batch_filter(y.cuda().to(dt), bar=False)
    out[pipe] = res
    d = (res.filter_means.cpu().double() - ref["filter_means"]).abs().reshape(t_len + 1, -1).max(1).values
    print(pipe, "max |mean diff| per row:", [f"{v:.2e}" for v in d.tolist()])
    print(pipe, "ll", res.loglikelihood.cpu().tolist(), "ref", ref["loglikelihood"].tolist())
    mism = (res.latest_state.previous_indices.cpu() != ref["prev_inds"]).sum().item()
    print(pipe, "final ancestor mismatches", mism)
