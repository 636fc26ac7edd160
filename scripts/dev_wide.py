#!/usr/bin/env python
"""dev: the on-chip wide sweeps (hank_wide.h) against the oracle and the per-period launches (parity), and their time.

    python scripts/dev_wide.py parity          small shapes, both value-function families
    python scripts/dev_wide.py time [N ...]    2000x11, T=300: wide against the default schedule
"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()       # before libhank_hip loads its HIP runtime

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402
from oracle.oracle import pad_N  # noqa: E402


def block(m, schedule, **env):
    old = {k: os.environ.get(k) for k in ("HANK_SCHEDULE", *env)}
    if schedule != "auto":
        os.environ["HANK_SCHEDULE"] = schedule
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
        hb = h.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return hb


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def parity(n_a, n_e, T, N, shock=0.05):
    m, ss, orc = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, "x1", shock)
    y = np.random.default_rng(0).standard_normal((2, P, N))
    k = min(N, 32)
    Nc = pad_N(k)
    xr = np.zeros((P, 1 + Nc)); xw = np.zeros((P, 1 + Nc))
    xr[:, 0], xw[:, 0] = x[2], x[3]
    xr[:, 1:1 + k], xw[:, 1:1 + k] = y[0][:, :k], y[1][:, :k]
    st, oagg, opol = orc.household_block(xr, xw, ss.value, ss.D, Nc)
    out = {}
    for sched in ("wide", "launch"):
        hb = block(m, sched)
        hb.set_boundary(ss.value, ss.D)
        agg, dagg = hb.primal_jvp(x[2:4], y)
        dpol = hb.dpolicy_seq(N)
        d2 = hb.jvp(y)
        out[sched] = (agg, dagg, dpol)
        print(f"{sched:6s} {n_a}x{n_e} T={T} N={N}: vs oracle agg {rel(agg, oagg[:, 0]):.2e} dagg {rel(dagg[:, :k], oagg[:, 1:1 + k]):.2e} "
              f"dpol {rel(dpol.transpose(2, 0, 1, 3)[..., :k], opol[..., 1:1 + k]):.2e} repeat {np.array_equal(d2, dagg)} family {hb.info()['last_tangent_family_name']}", flush=True)
        hb.close()
    w, l = out["wide"], out["launch"]
    print(f"   wide vs launch: dagg {rel(w[1], l[1]):.2e} dpol {rel(w[2], l[2]):.2e}", flush=True)


def timing(Ns, n_a=2000, n_e=11, T=300):
    m, ss, _ = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, "x1", 0.01)
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    res = {}
    variants = (("wide", {}), ("wide", {"HANK_WIDE_R": 4}), ("auto", {"HANK_WIDE_MIN": 100000}))
    if os.environ.get("DEV_WIDE_ONLY"):
        variants = variants[:2]
    for sched, env in variants:
        sched_name = sched + ("_r4" if env.get("HANK_WIDE_R") == 4 else "")
        hb = block(m, sched, **env)
        hb.set_boundary(ss.value, ss.D)
        for N in Ns:
            d_dx = torch.from_numpy(np.random.default_rng(N).standard_normal(2 * P * N)).to(dev)
            d_agg = torch.empty(P, dtype=torch.float64, device=dev); d_out = torch.empty(P * N, dtype=torch.float64, device=dev)
            for _ in range(2):
                hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
            hb.check()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_out.data_ptr())
            hb.sync()
            el = (time.perf_counter() - t0) / reps
            tm = hb.last_timings()
            hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
            hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr()); hb.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                hb.jvp_dev(d_dx.data_ptr(), N, d_out.data_ptr())
            hb.sync()
            elj = (time.perf_counter() - t0) / reps
            res[(sched_name, N)] = d_out.cpu().numpy().copy()
            print(f"{sched_name:7s} N={N:4d}: primal_jvp {1e3 * el:7.2f} ms = {N / el:8.0f} JVPs/s | jvp at recorded primal {1e3 * elj:7.2f} ms = {N / elj:8.0f} JVPs/s | "
                  + " ".join(f"{k2}={v['ms']:.2f}" for k2, v in tm.items() if v['ms'] > 0) + f" | {hb.info()['last_tangent_family_name']}", flush=True)
        hb.close()
    for N in Ns:
        for other in ("wide_r4", "auto"):
            if (other, N) in res:
                print(f"   N={N}: wide vs {other} dagg rel {rel(res[('wide', N)], res[(other, N)]):.2e}")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if mode == "parity":
        for shape in ((50, 2, 20, 4), (130, 3, 20, 5), (30, 3, 25, 1), (70, 7, 16, 3), (500, 4, 300, 8), (2000, 11, 12, 2)):
            parity(*shape)
    else:
        timing([int(a) for a in sys.argv[2:]] or [64, 128, 256])
