#!/usr/bin/env python
"""dev: the persistent tangent sweeps against the per-period launches at a recorded primal: bitwise dpol, aggregates to
rounding, and time per batch width.  usage: dev_tansweep.py [parity|time] [N ...]"""
import os
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


def block(m, schedule, xtan=None):
    os.environ["HANK_SCHEDULE"] = schedule
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    hb = h.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    os.environ.pop("HANK_SCHEDULE")
    return hb


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def parity(n_a, n_e, T, N, shock=0.05, envs=({},)):
    m, ss, _ = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, "x1", shock)
    y = np.random.default_rng(0).standard_normal((2, P, N))
    ref = None
    hb = block(m, "launch")
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])
    dagg0 = hb.jvp(y)
    dpol0 = hb.dpolicy_seq(N)
    hb.close()
    for env in envs:
        def run():
            hb = block(m, "xcd")
            hb.set_boundary(ss.value, ss.D)
            hb.primal(x[2:4])
            d = hb.jvp(y)
            dp = hb.dpolicy_seq(N)
            rep = np.array_equal(hb.jvp(y), d)
            st = hb.stats()
            hb.close()
            return d, dp, rep, st
        try:
            d, dp, rep, st = with_env(env, run)
        except Exception as ex:  # noqa: BLE001
            print(f"{n_a}x{n_e} T={T} N={N} {env}: FAILED {ex}", flush=True)
            continue
        sc = np.abs(dagg0).max()
        print(f"{n_a}x{n_e} T={T} N={N} {env}: dpol bitwise {np.array_equal(dp, dpol0)} (max diff {np.abs(dp - dpol0).max():.2e}) "
              f"dagg rel {np.abs(d - dagg0).max() / sc:.2e} repeat {rep} fallbacks {st['fallbacks']}", flush=True)


def timing(n_a, n_e, T, Ns, envs=({},), reps=None):
    m, ss, _ = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, "x1", 0.01)
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    for env in envs:
        def run():
            hb = block(m, env.get("SCHED", "xcd"))
            hb.set_boundary(ss.value, ss.D)
            d_agg = torch.empty(P, dtype=torch.float64, device=dev)
            hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr()); hb.check()
            for N in Ns:
                d_dx = torch.randn(2 * P * N, dtype=torch.float64, device=dev)
                d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
                for _ in range(2):
                    hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr())
                hb.check()
                r = reps or (10 if N <= 64 else 4)
                t0 = time.perf_counter()
                for _ in range(r):
                    hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr())
                hb.sync()
                el = (time.perf_counter() - t0) / r
                tm = hb.last_timings()
                hb.check()
                print(f"{env} {n_a}x{n_e} N={N:4d}: jvp {1e3 * el:7.3f} ms {N / el:9.0f} JVPs/s | back {tm['tangent_backward']['ms']:.3f} fwd {tm['tangent_forward']['ms']:.3f} "
                      f"| fallbacks {hb.stats()['fallbacks']}", flush=True)
            hb.close()
        try:
            with_env({k: v for k, v in env.items() if k != "SCHED"}, run)
        except Exception as ex:  # noqa: BLE001
            print(f"{env}: FAILED {ex}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "time"
    Ns = [int(v) for v in sys.argv[2:]] or [1, 8, 16, 32, 64, 128, 256]
    if what == "time":        # hank_jvp at a recorded primal: the persistent sweeps forced on every width, then the launches
        timing(2000, 11, 300, Ns, envs=({}, {"SCHED": "launch"}))
    if what == "parity":      # dpol of the persistent sweeps bit for bit against the launches
        for shape in ((50, 2, 100, 3), (37, 3, 9, 5), (130, 3, 20, 9), (200, 7, 40, 32), (40, 16, 8, 6), (37, 3, 9, 70), (500, 4, 300, 40)):
            parity(*shape)
        parity(2000, 11, 300, 32, shock=0.01)
