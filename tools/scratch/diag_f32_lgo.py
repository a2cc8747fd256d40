import sys, torch
sys.path.insert(0, ".")
from oracle import cpu_ref
from oracle.cases import CASE_BY_NAME, build_spec
from tests.helpers import DT, build_filter_from_case, load_golden
from pyfilter_amd.hints import HINTS
from pyfilter_amd.filters.particle.state import ParticleFilterCorrection
from pyfilter_amd.timeseries import TimeseriesState
for name in sys.argv[1:]:
    case = CASE_BY_NAME[name]
    g = load_golden(name, "f32"); g64 = load_golden(name, "f64")
    for route in (1, 2):
        HINTS.route = route
        filt = build_filter_from_case(case, g, torch.float32, "cuda")
        init = filt.initialize()
        es = init.timeseries_state.event_shape
        y = g["y"].cuda()
        for t in range(3):
            prev = init if t == 0 else ParticleFilterCorrection(TimeseriesState(t, g["step_x"][t-1].cuda(), es), g["step_w"][t-1].clone().cuda(), g["step_ll"][t-1].cuda(), g["step_idx"][t-1].cuda())
            st = filt.filter(y[t], prev)
            same = st.previous_indices.cpu() == g["step_idx"][t]
            xs = g["step_x"][t]
            # oracle f64 from the same f32 state
            spec = build_spec(case, torch.float64)
            xp = (g["x0"] if t == 0 else g["step_x"][t-1]).double(); wp = (torch.zeros(xp.shape[:2]) if t == 0 else g["step_w"][t-1]).double()
            ip = torch.arange(case["N"]).unsqueeze(-1).expand(case["N"], case["B"]) if t == 0 else g["step_idx"][t-1]
            if case["filter"] == "sisr":
                o = cpu_ref.sisr_step(spec, case["proposal"], y[t].cpu().double(), xp, wp, ip, g["z_tape"][t].double(), g["u_tape"][t].double(), case["ess_threshold"]*case["N"])
            else:
                o = cpu_ref.apf_step(spec, case["proposal"], y[t].cpu().double(), xp, wp, g["z_tape"][t].double(), g["u_tape"][t].double())
            dx_fix = (st.timeseries_state.value.cpu() - xs).abs()
            dx_or = (st.timeseries_state.value.cpu().double() - o[0]).abs()
            dfix_or = (xs.double() - o[0]).abs()
            print(name, "route", route, "t", t, "same anc", bool(same.all()), "| gpu-fix32", dx_fix.max().item(), "| gpu-oracle64", dx_or.max().item(), "| fix32-oracle64", dfix_or.max().item(),
                  "| w: gpu-or", (st.weights.cpu().double()-o[1]).abs().max().item(), "fix-or", (g["step_w"][t].double()-o[1]).abs().max().item())
