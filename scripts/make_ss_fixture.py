"""Solve the Krusell-Smith steady state on a given grid with the host solver and store it as a
benchmark input fixture (examples/fixtures/ks_ss_<n_a>x<n_e>.npz): 64 s of host Newton at 2000x11
that bench.py and the full-size GPU tests would otherwise repeat on every fresh box."""
import os
import sys
from pathlib import Path

import numpy as np

os.environ["HANK_NO_SS_FIXTURE"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import ks_setup  # noqa: E402

n_a, n_e = int(sys.argv[1]), int(sys.argv[2])
m, ss, _ = ks_setup(n_a, n_e, 300)
out = ROOT / "examples" / "fixtures" / f"ks_ss_{n_a}x{n_e}.npz"
np.savez_compressed(out, a_grid=m.heterogeneity["wealth"].grid, z_grid=m.heterogeneity["productivity"].grid,
                    Pi=m.heterogeneity["productivity"].transition, value=ss.value, D=ss.D, policy=ss.policies["KD"],
                    **{f"var_{k}": v for k, v in ss.vars.items()})
print(out, out.stat().st_size)
