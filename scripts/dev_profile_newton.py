"""dev: where does the host time of the Newton loop go? (cProfile around NewtonRaphsonHANK; GRID=2000x11 | MODEL=hank)"""
import cProfile, os, pstats, sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h
import hank_amd.parallel  # noqa: F401
INNER = os.environ.get("INNER", "fixed_point")
if os.environ.get("MODEL") == "hank":
    from examples.solve_hank import build
    m, ss = build(1000, 7, 500)
    P = 499
    exog = {"ei": 0.0025 * 0.6 ** np.arange(P)}
    x0 = np.tile(np.array([ss.vars[k] for k in h.vars_of_type(m, "endogenous")]), P)
else:
    from conftest import ks_setup
    NA, NE = (int(v) for v in os.environ.get("GRID", "500x4").split("x"))
    m, ss, _ = ks_setup(NA, NE, 300)
    P = 299
    exog = {"Z": 1.0 + 0.01 * 0.8 ** np.arange(1, P + 1)}
    x0 = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")]), P)
J = h.getSteadyStateJacobian(ss, m)
h.NewtonRaphsonHANK(x0, J, exog, m, ss, ss, inner=INNER)      # warm
pr = cProfile.Profile(); pr.enable()
h.NewtonRaphsonHANK(x0, J, exog, m, ss, ss, inner=INNER)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(24)
