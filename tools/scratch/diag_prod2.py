import sys, math, torch
sys.path.insert(0, ".")
from oracle.cases import CASE_BY_NAME, build_spec
from pyfilter_amd import ops
from pyfilter_amd.hints import HINTS
from tests.helpers import build_filter_from_case, load_golden
import tests.test_production_kernels_gpu as T
HINTS.route = 1
F32 = torch.float32
name = sys.argv[1]
case = CASE_BY_NAME[name]
g = load_golden(name, "f64")
spec64 = build_spec(case, torch.float64)
n, b = case["N"], case["B"]
for dtype, mode in ((F32, "batch"), (F32, "single"), (torch.float64, "batch")):
    y = g["y"].to(dtype)
    filt = build_filter_from_case(case, g, dtype, "cuda", tape=False, record_states=(mode == "batch"))
    filt.set_tape(u=g["u_tape"].to(dtype))
    es = filt._model.hidden.event_shape
    x_prev, w_prev = g["x0"].to(dtype), torch.zeros(g["x0"].shape[:2], dtype=dtype)
    idx_prev = torch.arange(n).unsqueeze(-1).expand(n, b).contiguous()
    prev = T._teacher_state(es, 0, x_prev.cuda(), w_prev.clone().cuda(), torch.zeros(b, dtype=dtype).cuda(), idx_prev.cuda())
    if mode == "batch":
        last = filt.batch_filter(y[0:1].cuda(), bar=False, init_state=prev).latest_state
    else:
        last = filt.filter(y[0].cuda(), prev)
    torch.cuda.synchronize()
    tr = ops.debug_launch_trace(1)
    seed = filt._last_run["seed_eff"]
    z = ops.debug_draw_normals(seed, 1, n, b, 3, dtype, "cuda").permute(0, 3, 2, 1).cpu()
    r64 = T._oracle_step(spec64, case, y[0], x_prev, w_prev, idx_prev, z[0], g["u_tape"].to(dtype)[0], torch.float64)
    xg = last.timeseries_state.value.cpu().double()
    print(dtype, mode, tr, "max dx", (xg - r64[0]).abs().max().item(), "max dw", (last.weights.cpu().double() - r64[1]).abs().max().item())
