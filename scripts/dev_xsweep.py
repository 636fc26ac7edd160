#!/usr/bin/env python
"""dev: the XCD-local persistent sweeps against the oracle and against the per-period launches (parity + time)."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()       # before libhank_hip loads its HIP runtime (the other order leaves torch without a device)

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402
from oracle.oracle import pad_N  # noqa: E402


def block(m, schedule):
    if schedule != "auto":
        os.environ["HANK_SCHEDULE"] = schedule
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    hb = h.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    os.environ.pop("HANK_SCHEDULE", None)
    return hb


def parity(n_a, n_e, T, N, shock=0.05, pols=True):
    m, ss, orc = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, "x1", shock)
    y = np.random.default_rng(0).standard_normal((2, P, N))
    Nc = pad_N(min(N, 32))
    xr = np.zeros((P, 1 + Nc)); xw = np.zeros((P, 1 + Nc))
    xr[:, 0], xw[:, 0] = x[2], x[3]
    k = min(N, 32)
    xr[:, 1:1 + k], xw[:, 1:1 + k] = y[0][:, :k], y[1][:, :k]
    st, oagg, opol = orc.household_block(xr, xw, ss.value, ss.D, Nc)
    out = {}
    for sched in ("xcd", "launch"):
        hb = block(m, sched)
        hb.set_boundary(ss.value, ss.D)
        agg, dagg = hb.primal_jvp(x[2:4], y)
        e = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
        line = f"{sched:6s} {n_a}x{n_e} T={T} N={N}: agg {e(agg, oagg[:, 0]):.2e} dagg {e(dagg[:, :k], oagg[:, 1:1 + k]):.2e}"
        if pols:
            line += f" pol {e(hb.policy_seq().transpose(2, 0, 1), opol[..., 0]):.2e} dpol {e(hb.dpolicy_seq(N).transpose(2, 0, 1, 3)[..., :k], opol[..., 1:1 + k]):.2e}"
            D = hb.dist_seq()
            line += f" mass {np.max(np.abs(D.sum(axis=(0, 1)) - 1)):.1e}"
        a2 = hb.primal(x[2:4]); d2 = hb.jvp(y)
        line += f" split==dual {np.array_equal(a2, agg) and np.array_equal(d2, dagg)} repeat {np.array_equal(hb.jvp(y), dagg)} stats {hb.stats()}"
        print(line, flush=True)
        out[sched] = (agg, dagg)
        hb.close()
    print(f"   xcd vs launch: agg {np.max(np.abs(out['xcd'][0] - out['launch'][0])):.2e} dagg {np.max(np.abs(out['xcd'][1] - out['launch'][1])) / np.abs(out['launch'][1]).max():.2e}", flush=True)


def timing(n_a, n_e, T, Ns, scheds=("xcd", "launch")):
    m, ss, _ = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, "x1", 0.01)
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    for sched in scheds:
        hb = block(m, sched)
        hb.set_boundary(ss.value, ss.D)
        for N in Ns:
            d_dx = torch.randn(2 * P * N, dtype=torch.float64, device=dev)
            d_agg = torch.empty(P, dtype=torch.float64, device=dev); d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
            for _ in range(2):
                hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
            hb.check()
            reps = 10 if N <= 64 else 4
            t0 = time.perf_counter()
            for _ in range(reps):
                hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
            hb.sync()
            el = (time.perf_counter() - t0) / reps
            tm = hb.last_timings()
            ms = " ".join(f"{k[:1]}{k.split('_')[1][:1]} {v['ms']:.3f}" for k, v in tm.items() if v["ms"] >= 0)
            hb.check()
            t0 = time.perf_counter()
            for _ in range(reps):
                hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr())
            hb.sync()
            elj = (time.perf_counter() - t0) / reps
            print(f"{sched:6s} {n_a}x{n_e} T={T} N={N:4d}: {1e3 * el:8.3f} ms/step  {N / el:9.0f} JVPs/s  | jvp only {1e3 * elj:7.3f} ms {N / elj:9.0f} JVPs/s | {ms}", flush=True)
            hb.check()
        # primal only
        d_agg = torch.empty(P, dtype=torch.float64, device=dev)
        hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr()); hb.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
        hb.sync()
        print(f"{sched:6s} primal only: {1e2 * (time.perf_counter() - t0):.3f} ms", flush=True)
        hb.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "parity"):
        parity(50, 2, 100, 3)
        parity(37, 3, 9, 5)
        parity(50, 2, 100, 1, shock=0.8)
        parity(500, 4, 300, 4, shock=0.01)
        parity(40, 16, 8, 6)
        parity(37, 3, 9, 70, pols=False)
        parity(2000, 11, 300, 32, shock=0.01, pols=False)
    if what == "ra":      # run-ahead wave on/off per kernel (HANK_XRUNAHEAD bits: 1 tangent backward, 2 tangent forward, 4 primal forward)
        for mask in (0, 1, 2, 4, 7):
            os.environ["HANK_XRUNAHEAD"] = str(mask)
            print("HANK_XRUNAHEAD =", mask, flush=True)
            timing(2000, 11, 300, [1, 32], scheds=("xcd",))
    if what == "tl2":     # geometry knobs (HANK_RG_B / HANK_RG_F / HANK_FWD_SS in the environment) at the two widths that matter
        timing(2000, 11, 300, [32, 64], scheds=("launch",))
    if what == "tl":      # the launched sweeps alone
        timing(2000, 11, 300, [16, 32, 64, 128, 256], scheds=("launch",))
    if what == "ta":      # the default schedule (what bench.py times)
        timing(2000, 11, 300, [1, 8, 16, 32, 36, 64], scheds=("auto",))
    if what == "tx":      # the persistent sweeps alone, the widths that matter
        timing(2000, 11, 300, [1, 16, 32, 64], scheds=("xcd",))
    if what in ("all", "time"):
        timing(2000, 11, 300, [1, 8, 16, 32, 64, 128, 256])
        timing(500, 4, 300, [1, 32])
