"""experiment: do dual-sweep passes of independent contexts (separate streams) overlap on the GPU?"""
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
d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
def mk():
    hb = HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, 300)
    hb.set_boundary(ss.value, ss.D); return hb
for K, Nc in [(1, 32), (2, 32), (3, 32), (4, 32), (2, 16), (1, 64), (2, 64), (2, 128), (1, 256)]:
    hbs = [mk() for _ in range(K)]
    dxs = [torch.from_numpy(np.random.default_rng(k).standard_normal(2 * P * Nc)).to(dev) for k in range(K)]
    aggs = [torch.empty(P, dtype=torch.float64, device=dev) for _ in range(K)]
    outs = [torch.empty(P * Nc, dtype=torch.float64, device=dev) for _ in range(K)]
    def go():
        for hb, dx, a, o in zip(hbs, dxs, aggs, outs): hb.primal_jvp_dev(d_x.data_ptr(), dx.data_ptr(), Nc, a.data_ptr(), o.data_ptr())
    go(); torch.cuda.synchronize()
    for hb in hbs: hb.sync()
    t0 = time.perf_counter()
    for _ in range(5): go()
    for hb in hbs: hb.sync()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 5
    print(f"K={K} contexts x N={Nc}: {el*1e3:.3f} ms per {K*Nc} tangents = {K*Nc/el:.0f} JVPs/s", flush=True)
    for hb in hbs: hb.close()
