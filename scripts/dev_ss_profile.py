#!/usr/bin/env python
"""dev: where does the cold steady state go? (device VFI steps / stationary-distribution iterations / host)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from hank_amd import hip as hh  # noqa: E402

n_a, n_e, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2000, 11, 300)))
m = h.build_model_from_yaml(str(ROOT / "examples" / "krusell_smith.yaml"), overrides={"T": T, "dimensions": {"wealth": {"n": n_a}, "productivity": {"n": n_e}}})
acc = {"vfi_s": 0.0, "vfi_calls": 0, "vfi_steps": 0, "dist_s": 0.0, "dist_calls": 0, "dist_iters": 0}
_vfi, _sd = hh.HouseholdBlock.vfi, hh.HouseholdBlock.stationary_dist


def vfi(self, *a, **k):
    t0 = time.perf_counter(); out = _vfi(self, *a, **k); acc["vfi_s"] += time.perf_counter() - t0
    acc["vfi_calls"] += 1; acc["vfi_steps"] += out[2]
    return out


def sd(self, *a, **k):
    t0 = time.perf_counter(); out = _sd(self, *a, **k); acc["dist_s"] += time.perf_counter() - t0
    acc["dist_calls"] += 1; acc["dist_iters"] += int(out[1])
    return out


hh.HouseholdBlock.vfi, hh.HouseholdBlock.stationary_dist = vfi, sd
import cProfile, pstats, io
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
ss_i, ss_e = h.get_SteadyStates(m)
pr.disable()
tot = time.perf_counter() - t0
sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(28); print(sio.getvalue()[-6000:])
print(f"{n_a}x{n_e}: steady states {tot:.2f} s | VFI {acc['vfi_s']:.2f} s in {acc['vfi_calls']} calls, {acc['vfi_steps']} steps "
      f"({1e6 * acc['vfi_s'] / max(acc['vfi_steps'], 1):.1f} us/step) | stationary distribution {acc['dist_s']:.2f} s in {acc['dist_calls']} calls, "
      f"{acc['dist_iters']} iterations ({1e6 * acc['dist_s'] / max(acc['dist_iters'], 1):.1f} us/iteration) | rest {tot - acc['vfi_s'] - acc['dist_s']:.2f} s")
