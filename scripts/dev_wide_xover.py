#!/usr/bin/env python
"""dev: where the on-chip wide family overtakes the per-period launches (HANK_WIDE_MIN), 2000x11, T=300:
    python scripts/dev_wide_xover.py [N ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import dev_wide as dw  # noqa: E402  (torch.cuda.init() first, then the library)
import numpy as np  # noqa: E402
import time  # noqa: E402
import torch  # noqa: E402

Ns = [int(a) for a in sys.argv[1:]] or [96, 112, 128, 144, 160]
m, ss, _ = dw.ks_setup(2000, 11, 300)
P = 299
x, Z = dw.ks_paths(m, ss, "x1", 0.01)
dev = torch.device("cuda", 0)
d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
for name, sched, env in (("wide", "wide", {}), ("launch", "auto", {"HANK_WIDE_MIN": 100000})):
    hb = dw.block(m, sched, **env)
    hb.set_boundary(ss.value, ss.D)
    for N in Ns:
        d_dx = torch.from_numpy(np.random.default_rng(N).standard_normal(2 * P * N)).to(dev)
        d_agg = torch.empty(P, dtype=torch.float64, device=dev); d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
        for _ in range(2):
            hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
        hb.check()
        t0 = time.perf_counter()
        for _ in range(5):
            hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
        hb.sync()
        el = (time.perf_counter() - t0) / 5
        hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr()); hb.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr())
        hb.sync()
        elj = (time.perf_counter() - t0) / 5
        print(f"{name:6s} N={N:4d}: primal_jvp {1e3 * el:7.2f} ms | jvp at recorded primal {1e3 * elj:7.2f} ms | {hb.info()['last_tangent_family_name']}", flush=True)
    hb.close()
