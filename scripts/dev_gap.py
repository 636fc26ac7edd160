"""dev: does the time of a persistent sweep depend on what ran just before it? (back-to-back calls against calls after an idle gap)"""
import os, sys, time
from pathlib import Path
import numpy as np
import torch
torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ["HANK_SCHEDULE"] = os.environ.get("SCHED", "xcd")
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402
m, ss, _ = ks_setup(2000, 11, 300)
P = 299; N = 32
x, _ = ks_paths(m, ss, "x1", 0.01)
hb = h.household_block(m); hb.set_boundary(ss.value, ss.D)
dev = torch.device("cuda", 0)
d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
d_dx = torch.randn(2 * P * N, dtype=torch.float64, device=dev)
d_agg = torch.empty(P, dtype=torch.float64, device=dev); d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
def call():
    hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
for _ in range(3):
    call()
hb.sync()
def show(tag):
    tm = hb.last_timings()
    print(tag, " ".join(f"{k}={v['ms']:.3f}" for k, v in tm.items() if v["ms"] > 0.01), flush=True)
for gap in (0.0, 0.001, 0.01, 0.1, 0.5):
    for rep in range(3):
        for _ in range(5):
            call()
        hb.sync(); time.sleep(gap)
        call(); hb.sync()
        show(f"after 5 back-to-back + gap {gap:5.3f}s:")
for _ in range(10):
    call()
hb.sync(); show("10 back-to-back, last:   ")
