#!/usr/bin/env python
"""dev: soak the persistent sweeps — many primal / JVP / value-iteration / power-method launches back to back, results
compared bit for bit with the first of their kind, no fallback to the launches allowed."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, Z = ks_paths(m, ss, "x1", 0.01)
hb = h.household_block(m)
hb.set_boundary(ss.value, ss.D)
rng = np.random.default_rng(0)
ys = {N: rng.standard_normal((2, P, N)) for N in (1, 16, 32, 64)}
ref = {}
dev = {}
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    agg = hb.primal(x[2:4])
    assert np.array_equal(ref.setdefault("agg", agg), agg)
    for N, y in ys.items():
        d = hb.jvp(y)
        assert np.array_equal(ref.setdefault(N, d), d), N
    # the Dual pass (persistent for one-pass batches, the launches beyond) at an x that is never the one on record
    k = n % 2
    for q, N in enumerate((1, 16, 32, 64)):
        a2, d2 = hb.primal_jvp(x[2:4] * (1.0 + 1e-3 * (k + 1) + 1e-4 * q), ys[N])      # (a different x per call: no memo hit)
        assert np.array_equal(ref.setdefault(("dual_agg", k, N), a2), a2) and np.array_equal(ref.setdefault(("dual", k, N), d2), d2), (k, N)
    # the device-pointer Dual pass (what bench.py times: k_xdual_prologue / k_xdual_epilogue around the two sweeps)
    for N in (16, 32):
        if ("dev", N) not in dev:
            dev[("dev", N)] = (torch.from_numpy(np.asfortranarray(x[2:4] * 1.0007).reshape(-1, order="F").copy()).cuda(),
                               torch.from_numpy(ys[N].reshape(-1, order="F").copy()).cuda(),
                               torch.empty(P, dtype=torch.float64, device="cuda"), torch.empty(P * N, dtype=torch.float64, device="cuda"))
        dx_, dy_, da_, dd_ = dev[("dev", N)]
        hb.primal_jvp_dev(dx_.data_ptr(), dy_.data_ptr(), N, da_.data_ptr(), dd_.data_ptr()); hb.sync(); hb.check()
        a3, d3 = da_.cpu().numpy(), dd_.cpu().numpy()
        assert np.array_equal(ref.setdefault(("dev_agg", N), a3), a3) and np.array_equal(ref.setdefault(("dev", N), d3), d3), ("dev", N)
    if n % 10 == 0:
        v, pol, it, nrm = hb.vfi(np.ones((2000, 11)), [ss.vars["r"], ss.vars["w"]], 1e-11)
        assert np.array_equal(ref.setdefault("v", v), v) and it == ref.setdefault("it", it)
        D, st = hb.stationary_dist(ss.policies["KD"], D0=ss.D)
        assert np.array_equal(ref.setdefault("D", D), D)
        hb.set_boundary(ss.value, ss.D)
    n += 1
st = hb.stats()
print(f"{n} rounds in {time.perf_counter() - t0:.1f} s: {st}")
assert st["fallbacks"] == 0
