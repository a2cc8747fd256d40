import sys, math, torch
sys.path.insert(0, ".")
from oracle import models as M
from oracle import cpu_ref
from pyfilter_amd import ops, timeseries as ts
from pyfilter_amd.timeseries import models
from pyfilter_amd.filters.particle import SISR, APF, proposals
from pyfilter_amd.hints import HINTS
import tests.test_production_kernels_gpu as T
HINTS.route = 1
F32 = torch.float32
dev = "cuda"
t = lambda v, dt=F32: torch.tensor(v, dtype=dt, device=dev)
def run(A, bvec, svec, filt_cls=SISR, n=256, b=1):
    hidden = models.Lorenz63(t(10.0), t(28.0), t(8.0 / 3.0), t(1.0), dt=0.01, initial_mean=t([-5.91652, -5.52332, 24.5723]), initial_scale=t([math.sqrt(10.0)] * 3))
    o = len(bvec)
    ssm = ts.LinearStateSpaceModel(hidden, (t(A), t(bvec), t(svec)), torch.Size([o])).to(dev)
    filt = filt_cls(ssm, n, proposal=proposals.LinearGaussianObservations(), ess_threshold=0.5, record_states=True)
    filt.set_batch_shape(torch.Size([b]))
    g = torch.Generator().manual_seed(3)
    x0 = torch.tensor([-5.9, -5.5, 24.5]) + 3.0 * torch.randn(n, b, 3, generator=g)
    y = torch.tensor([[-5.0, -5.0, 24.0][:o]]) + 0.1
    filt.set_tape(u=torch.full((2, b), 0.37))
    es = hidden.event_shape
    idx_prev = torch.arange(n).unsqueeze(-1).expand(n, b).contiguous()
    prev = T._teacher_state(es, 0, x0.to(F32).cuda(), torch.zeros(n, b).cuda(), torch.zeros(b).cuda(), idx_prev.cuda())
    last = filt.batch_filter(y.to(F32).cuda(), bar=False, init_state=prev).latest_state
    torch.cuda.synchronize()
    tr = ops.debug_launch_trace(1)[0]
    z = T._normals_ref_layout(filt, 1, n, b, 3, True)
    spec = M.ModelSpec(M.HID_LORENZ63_EM, (10.0, 28.0, 8.0 / 3.0, 1.0), 3, 0.01, (torch.zeros(3), torch.ones(3)), M.OBS_LINEAR,
                       (torch.tensor(A, dtype=torch.float64), torch.tensor(bvec, dtype=torch.float64), torch.tensor(svec, dtype=torch.float64)), o)
    r = cpu_ref.sisr_step(spec, "lgo", y[0].double(), x0.double(), torch.zeros(n, b, dtype=torch.float64), idx_prev, z[0].double(), torch.full((b,), 0.37, dtype=torch.float64), 0.5 * n)
    xg = last.timeseries_state.value.cpu().double()
    d = (xg - r[0]).abs().amax(dim=(0, 1))
    print("SPEC", tr["SPEC"], "O", o, "A", A, "b", bvec, "s", svec, "-> max |dx| per component", [f"{v:.2e}" for v in d.tolist()])
I3 = [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]
run(I3, [0.0, 0.0, 0.0], [0.3, 0.3, 0.3])
run(I3, [0.1, -0.2, 0.3], [0.3, 0.3, 0.3])
run(I3, [0.0, 0.0, 0.0], [0.3, 0.4, 0.5])
run(I3[:2], [0.1, -0.2], [0.3, 0.4])
run([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 0.0]], [0.0, 0.0, 0.0], [0.3, 0.3, 1.0])
run([[0.8, 0.1, 0.0], [-0.2, 0.9, 0.05], [0.0, 0.3, 0.7]], [0.0, 0.0, 0.0], [0.3, 0.3, 0.3])
run([[1.0, 0.5, 0], [0, 1.0, 0], [0, 0, 1.0]], [0.0, 0.0, 0.0], [0.3, 0.3, 0.3])
run([[1.0, 0, 0], [0, 1.0, 0], [0, 0.5, 1.0]], [0.0, 0.0, 0.0], [0.3, 0.3, 0.3])
run([[1.0, 0, 0.5], [0, 1.0, 0], [0, 0, 1.0]], [0.0, 0.0, 0.0], [0.3, 0.3, 0.3])
