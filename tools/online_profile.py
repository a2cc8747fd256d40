"""cProfile of the online filter() move at 16 x 4096 (host cost per move)"""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import make
for cfg in (("sine", "apf", "lgo", 4096, 16), ("sine", "apf", "lgo", 8192, 128)):
    f, _ = make(*cfg)
    state = f.initialize()
    y = torch.tensor(0.1, device="cuda")
    for _ in range(20):
        state = f.filter(y, state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        state = f.filter(y, state)
    t_host = (time.perf_counter() - t0) / 300 * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 300 * 1e6
    print(cfg, f"host issue {t_host:.1f} us per move, incl. device drain {t_all:.1f} us")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        state = f.filter(y, state)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(14)
