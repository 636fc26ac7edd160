"""What `rocprofv3` should see of a y-iteration: ONE Float64 sweep (hank_primal), then JVP batches at that record (hank_jvp),
default schedule, 2000x11, T=300 (scripts/profile_jvp.sh wraps this in the kernel-trace and counter passes)."""
import os, sys
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402

N = int(os.environ.get("TANGENTS", "32")); REPS = int(os.environ.get("REPS", "8"))
m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, _ = ks_paths(m, ss, "x1", 0.01)
wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
hb = h.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
hb.set_boundary(ss.value, ss.D)
dev = torch.device("cuda", 0)
d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
d_dx = torch.randn(2 * P * N, dtype=torch.float64, device=dev)
d_agg = torch.empty(P, dtype=torch.float64, device=dev); d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
for _ in range(2):
    hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
for _ in range(REPS):
    hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr())
hb.sync(); hb.check()
print("stats", hb.stats())
