import sys, math, torch
sys.path.insert(0, ".")
from oracle.cases import CASE_BY_NAME, build_spec
from pyfilter_amd import ops
from pyfilter_amd.hints import HINTS
from tests.helpers import build_filter_from_case, load_golden
import tests.test_production_kernels_gpu as T
HINTS.route = 1
F32 = torch.float32
for name in sys.argv[1:]:
    case = CASE_BY_NAME[name]
    dt = "f32" if "f32" in case["dtypes"] else "f64"
    g = load_golden(name, dt)
    spec64 = build_spec(case, torch.float64)
    n, b = case["N"], case["B"]
    y = g["y"].to(F32)
    filt = build_filter_from_case(case, g, F32, "cuda", tape=False, record_states=True)
    filt.set_tape(u=g["u_tape"].to(F32))
    es = filt._model.hidden.event_shape
    x_prev, w_prev = g["x0"].to(F32), torch.zeros(g["x0"].shape[:2], dtype=F32)
    idx_prev = torch.arange(n).unsqueeze(-1).expand(n, b).contiguous()
    prev = T._teacher_state(es, 0, x_prev.cuda(), w_prev.clone().cuda(), torch.zeros(b).cuda(), idx_prev.cuda())
    res = filt.batch_filter(y[0:1].cuda(), bar=False, init_state=prev)
    torch.cuda.synchronize()
    print(ops.debug_launch_trace(1))
    z = T._normals_ref_layout(filt, 1, n, b, 3, True)
    r64 = T._oracle_step(spec64, case, y[0], x_prev, w_prev, idx_prev, z[0], g["u_tape"].to(F32)[0], torch.float64)
    last = res.latest_state
    xg = last.timeseries_state.value.cpu().double()
    print("x gpu", xg[:3, 0], "\nx or", r64[0][:3, 0], "\nmax dx", (xg - r64[0]).abs().max().item(), "scale", r64[0].abs().max().item())
    print("w gpu", last.weights.cpu()[:3, 0], "w or", r64[1][:3, 0])
