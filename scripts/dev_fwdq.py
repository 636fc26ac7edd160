"""dev: forward workgroups walking Q row tiles (HANK_FWD_Q): results must not depend on Q bit for bit; time per Q."""
import os, sys, time
from pathlib import Path
import numpy as np
import torch
torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402

m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, _ = ks_paths(m, ss, "x1", 0.01)
wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
dev = torch.device("cuda", 0)
d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
ref = {}
for Q in [int(q) for q in os.environ.get("QS", "1,2,4,8").split(",")]:
    os.environ["HANK_FWD_Q"] = str(Q); os.environ["HANK_SCHEDULE"] = "launch"
    hb = h.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    hb.set_boundary(ss.value, ss.D)
    for N in [int(n) for n in os.environ.get("NS", "32,64,128,256").split(",")]:
        g = torch.Generator(device=dev); g.manual_seed(N)
        d_dx = torch.randn(2 * P * N, dtype=torch.float64, device=dev, generator=g)
        d_agg = torch.empty(P, dtype=torch.float64, device=dev); d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
        for _ in range(2):
            hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
        hb.check()
        out = d_out.cpu().numpy().copy()
        same = np.array_equal(out, ref.setdefault(N, out))
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps):
            hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
        hb.sync()
        el = (time.perf_counter() - t0) / reps
        tm = hb.last_timings()
        print(f"Q={Q} N={N:4d}: {el*1e3:8.3f} ms/step {N/el:9.0f} JVPs/s | db {tm.get('dual_backward_ms', 0):.3f} df {tm.get('dual_forward_ms', 0):.3f} | bitwise == Q=1: {same}", flush=True)
    hb.close()
