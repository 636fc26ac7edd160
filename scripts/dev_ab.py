#!/usr/bin/env python
"""dev: time hank_primal_jvp at the headline size under one schedule — used to A/B build knobs."""
import os, sys, time
from pathlib import Path
import numpy as np
import torch
torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402
os.environ["HANK_SCHEDULE"] = sys.argv[1] if len(sys.argv) > 1 else "launch"
m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, Z = ks_paths(m, ss, "x1", 0.01)
hb = h.household_block(m)
hb.set_boundary(ss.value, ss.D)
dev = torch.device("cuda", 0)
d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
for N in (1, 32, 256):
    d_dx = torch.randn(2 * P * N, dtype=torch.float64, device=dev)
    d_agg = torch.empty(P, dtype=torch.float64, device=dev); d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
    for _ in range(3):
        hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
    hb.check()
    reps = 20 if N <= 32 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
    hb.sync()
    el = (time.perf_counter() - t0) / reps
    tm = hb.last_timings()
    print(f"{os.environ['HANK_SCHEDULE']} N={N:4d}: {1e3 * el:8.3f} ms/step {N / el:9.0f} JVPs/s  " + " ".join(f"{k}={v['ms']:.3f}" for k, v in tm.items() if v['ms'] >= 0), flush=True)
