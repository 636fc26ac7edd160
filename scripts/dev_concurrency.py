"""experiment: do tangent sweeps of independent contexts (separate streams) overlap on the GPU?"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd as h
from hank_amd.hip import HouseholdBlock
from conftest import ks_setup, ks_paths
m, ss, _ = ks_setup(2000, 11, 300)
x, Z = ks_paths(m, ss, "x1", 0.01)
wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
P = 299
dev = torch.device("cuda", 0)
def mk():
    hb = HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, 300)
    hb.set_boundary(ss.value, ss.D); hb.primal(x[2:4]); return hb
for K, Nc in [(1, 32), (2, 16), (4, 8), (8, 4)]:
    hbs = [mk() for _ in range(K)]
    dxs = [torch.from_numpy(np.random.default_rng(k).standard_normal(2 * P * Nc)).to(dev) for k in range(K)]
    outs = [torch.empty(P * Nc, dtype=torch.float64, device=dev) for _ in range(K)]
    for hb, dx, o in zip(hbs, dxs, outs): hb.jvp_dev(dx.data_ptr(), Nc, o.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for hb, dx, o in zip(hbs, dxs, outs): hb.jvp_dev(dx.data_ptr(), Nc, o.data_ptr())
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 5
    print(f"K={K} chunks of N={Nc}: {el*1e3:.3f} ms per 32 tangents; single-context sweeps {hbs[0].last_timings()['tangent_backward']['ms']:.2f}+{hbs[0].last_timings()['tangent_forward']['ms']:.2f}")
    for hb in hbs: hb.close()
