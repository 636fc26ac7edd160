"""diagnostic: per-phase cycle shares of the backward cluster sweep (HANK_STAMPS build)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd.hip as hip
hip._LIB_PATH = hip._LIB_PATH.with_name('libhank_hip_stamps.so')
import hank_amd as h
from conftest import ks_setup, ks_paths
m, ss, _ = ks_setup(2000, 11, 300)
hb = h.household_block(m); hb.set_boundary(ss.value, ss.D)
x, Z = ks_paths(m, ss, "x1", 0.01)
hb.primal(x[2:4])
y = np.random.default_rng(0).standard_normal((2, 299, 32))
for _ in range(3): hb.jvp(y)
lib = hip.load_library()
out = (C.c_ulonglong * (256 * 8))()
lib.hank_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
n = lib.hank_debug_stamps(hb._ctx, out, 256)
a = np.array(out[:n * 8], dtype=np.float64).reshape(n, 8) / 299.0 / 100.0   # s_memtime ticks at 100 MHz -> us per period
names = ["loop top", "X compute+st", "publish(drain+bar+flag)", "coef loads issue", "wait(poll+bar)", "gather+Y", "end barrier"]
print("per-period us (mean / min / max over workgroups):")
for k, nm in enumerate(names): print(f"  {nm:28s} {a[:,k].mean():7.3f} {a[:,k].min():7.3f} {a[:,k].max():7.3f}")
print("  sum", a[:, :7].sum(axis=1).mean(), hb.last_timings())
