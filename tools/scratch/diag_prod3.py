import sys, math, torch, itertools
sys.path.insert(0, ".")
from oracle.cases import CASE_BY_NAME, build_spec
from oracle import models as M
from pyfilter_amd import ops
from pyfilter_amd.hints import HINTS
from tests.helpers import build_filter_from_case, load_golden
import tests.test_production_kernels_gpu as T
HINTS.route = 1
F32 = torch.float32
name = "lorenz_o3_sisr_lgo"
case = CASE_BY_NAME[name]
g = load_golden(name, "f64")
spec64 = build_spec(case, torch.float64)
n, b = case["N"], case["B"]
y = g["y"].to(F32)
filt = build_filter_from_case(case, g, F32, "cuda", tape=False, record_states=True)
filt.set_tape(u=g["u_tape"].to(F32))
es = filt._model.hidden.event_shape
x_prev, w_prev = g["x0"].to(F32), torch.zeros(g["x0"].shape[:2], dtype=F32)
idx_prev = torch.arange(n).unsqueeze(-1).expand(n, b).contiguous()
prev = T._teacher_state(es, 0, x_prev.cuda(), w_prev.clone().cuda(), torch.zeros(b).cuda(), idx_prev.cuda())
last = filt.batch_filter(y[0:1].cuda(), bar=False, init_state=prev).latest_state
torch.cuda.synchronize()
z = T._normals_ref_layout(filt, 1, n, b, 3, True)
xg = last.timeseries_state.value.cpu().double()
a, ob, os_ = [p.double() for p in spec64.obs_params]
def trial(tag, a_, b_, s_, y_, od=3):
    sp = M.ModelSpec(spec64.hidden, spec64.hidden_params, 3, spec64.dt, spec64.init, M.OBS_LINEAR, (a_, b_, s_), od)
    r = T._oracle_step(sp, case, y_, x_prev, w_prev, idx_prev, z[0], g["u_tape"].to(F32)[0], torch.float64)
    print(f"{(xg - r[0]).abs().max().item():.3e} max dx  {tag}")
y0 = y[0].double()
trial("as is", a, ob, os_, y0)
trial("b = 0", a, torch.zeros(3, dtype=torch.float64), os_, y0)
trial("drop row 3", a[:2], ob[:2], os_[:2], y0[:2], 2)
trial("drop row 1", a[1:], ob[1:], os_[1:], y0[1:], 2)
for perm in itertools.permutations(range(3)):
    p = list(perm)
    trial(f"s perm {p}", a, ob, os_[p], y0)
    trial(f"b perm {p}", a, ob[p], os_, y0)
    trial(f"y perm {p}", a, ob, os_, y0[p])
trial("s = 1", a, ob, torch.ones(3, dtype=torch.float64), y0)
trial("A^T", a.t().contiguous(), ob, os_, y0)
trial("y = y[1]", a, ob, os_, y[1].double())
