"""dev: cost of switching the tangent batch width (workspace + graph re-capture)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd as h
from conftest import ks_setup, ks_paths
for (n_a, n_e) in [(500, 4), (2000, 11)]:
    m, ss, _ = ks_setup(n_a, n_e, 300)
    x, Z = ks_paths(m, ss, "x1", 0.01)
    hb = h.household_block(m); hb.set_boundary(ss.value, ss.D); hb.primal(x[2:4])
    rng = np.random.default_rng(0)
    for N in (256, 1, 256, 1, 1):
        y = rng.standard_normal((2, 299, N))
        t0 = time.perf_counter(); hb.jvp(y); t1 = time.perf_counter(); hb.jvp(y); t2 = time.perf_counter()
        print(f"{n_a}x{n_e} N={N}: first call {1e3*(t1-t0):.1f} ms, second {1e3*(t2-t1):.1f} ms", flush=True)
