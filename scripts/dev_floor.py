"""dev: how does the per-launch floor of the dual-sweep launches depend on the grid? (N = 2 and N = 32)"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd as h
from conftest import ks_setup, ks_paths
dev = torch.device("cuda", 0)
for n_a, n_e in [(500, 4), (2000, 4), (500, 11), (1000, 11), (2000, 11)]:
    t0 = time.perf_counter()
    m, ss, _ = ks_setup(n_a, n_e, 300)
    tss = time.perf_counter() - t0
    x, Z = ks_paths(m, ss, "x1", 0.01)
    P = 299
    hb = h.household_block(m); hb.set_boundary(ss.value, ss.D)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    d_agg = torch.empty(P, dtype=torch.float64, device=dev)
    for N in (2, 32):
        d_dx = torch.randn(2 * P * N, dtype=torch.float64, device=dev)
        d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
        for _ in range(3): hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
        hb.check()
        tm = hb.last_timings()
        b, f = tm["dual_backward"], tm["dual_forward"]
        print(f"{n_a}x{n_e} N={N}: backward {1e3*b['ms']/b['launches']:.2f} us/launch, forward {1e3*f['ms']/f['launches']:.2f} us/launch (ss {tss:.0f} s)", flush=True)
    hb.close()
