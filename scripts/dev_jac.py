#!/usr/bin/env python
"""dev: time the two branches of getSteadyStateJacobian (toeplitz | columns) at the headline grid and the HANK size."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_setup  # noqa: E402

m, ss, _ = ks_setup(2000, 11, 300)
for meth in ("toeplitz", "columns", "toeplitz"):
    t0 = time.perf_counter()
    J = h.getSteadyStateJacobian(ss, m, method=meth)
    print(f"2000x11 T=300 {meth}: {time.perf_counter() - t0:.4f} s", flush=True)
    if meth == "toeplitz":
        Jt = J.toarray()
    else:
        print("   max |toeplitz - columns| =", np.max(np.abs(Jt - J.toarray())), "scale", np.max(np.abs(Jt)), flush=True)
hb = h.household_block(m)
for _ in range(3):
    t0 = time.perf_counter(); hb.fake_news(); print(f"   hank_fake_news alone: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
from examples.solve_hank import build  # noqa: E402
m2, ss2 = build(1000, 7, 500)
for meth in ("toeplitz", "columns", "toeplitz"):
    t0 = time.perf_counter()
    J = h.getSteadyStateJacobian(ss2, m2, method=meth)
    print(f"HANK 1000x7 T=500 {meth}: {time.perf_counter() - t0:.4f} s", flush=True)
    if meth == "toeplitz":
        Jt = J.toarray()
    else:
        print("   max |toeplitz - columns| =", np.max(np.abs(Jt - J.toarray())), "scale", np.max(np.abs(Jt)), flush=True)
