"""dev: where does the host time of the Newton loop go? (cProfile around NewtonRaphsonHANK on 500x4, T=300)"""
import cProfile, pstats, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd as h
import hank_amd.parallel
from conftest import ks_setup
import os
NA, NE = (int(v) for v in os.environ.get("GRID", "500x4").split("x"))
m, ss, _ = ks_setup(NA, NE, 300)
P = 299
Z = 1.0 + 0.01 * 0.8 ** np.arange(1, P + 1)
x0 = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")]), P)
J = h.getSteadyStateJacobian(ss, m)
h.NewtonRaphsonHANK(x0, J, {"Z": Z}, m, ss, ss)      # warm
pr = cProfile.Profile(); pr.enable()
h.NewtonRaphsonHANK(x0, J, {"Z": Z}, m, ss, ss)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
