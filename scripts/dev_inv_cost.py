"""dev: what the first device inverse of a process costs at the one-asset HANK size (n = 3493), by linalg backend / routine."""
import sys, time, numpy as np, torch
mode = sys.argv[1] if len(sys.argv) > 1 else "inv"
t0 = time.perf_counter(); torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize(); print(f"[{mode}] cuda init {time.perf_counter()-t0:.3f}s")
if "cusolver" in mode:
    torch.backends.cuda.preferred_linalg_library("cusolver")
if "magma" in mode:
    torch.backends.cuda.preferred_linalg_library("magma")
n = 3493
rng = np.random.default_rng(0)
A = np.eye(n) + 0.01 * rng.standard_normal((n, n))
for rep in range(2):
    t0 = time.perf_counter(); d = torch.from_numpy(A).to("cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    if "solve" in mode:
        Ai = torch.linalg.solve(d, torch.eye(n, dtype=torch.float64, device="cuda"))
    else:
        Ai = torch.linalg.inv(d)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    b = torch.from_numpy(rng.standard_normal(n)).to("cuda"); y = (Ai @ b).cpu(); t3 = time.perf_counter()
    print(f"[{mode}] rep {rep}: H2D {t1-t0:.3f}s inverse {t2-t1:.3f}s first gemv+D2H {t3-t2:.4f}s")
