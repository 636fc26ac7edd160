#!/usr/bin/env python
"""bench.py — sequence-space JVPs/sec of the MI355X household block (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: the primal sweep at x
(BackwardIteration + ForwardIteration on Float64, recorded once) followed by ONE batched JVP of N
tangent directions through the same two sweeps. Inputs (x, tangents, boundary) are resident in HBM
before the timed region starts. Default workload = BASELINE.json configs[2]: Krusell–Smith
2000x11 grid, T=300, 32-wide tangent batch on 1 GPU (the configuration the north star's target is
quoted on; configs[1] — 500x4, single tangent — is latency-bound and is reported in `extra`).

N>1 GPUs: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; tangent
columns shard across ranks (32 per GPU, weak scaling), every rank runs the primal redundantly, one
RCCL all-gather per step assembles the P x (32·N) aggregate-tangent block on every rank.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    "ks_2000x11_T300_N32": dict(n_a=2000, n_e=11, T=300, N=32),   # BASELINE.json configs[2]
    "ks_500x4_T300_N1": dict(n_a=500, n_e=4, T=300, N=1),          # configs[1]
    "ks_50x2_T100_N1": dict(n_a=50, n_e=2, T=100, N=1),            # configs[0] (plumbing)
}


def load_or_solve_ss(n_a, n_e, T):
    """model + steady state; the 2000x11 steady state (64 s of host Newton) is cached as a fixture."""
    import hank_amd as h
    from conftest import ks_setup
    fx = ROOT / "examples" / "fixtures" / f"ks_ss_{n_a}x{n_e}.npz"
    if fx.exists():
        ov = {"T": T, "dimensions": {"wealth": {"n": n_a}, "productivity": {"n": n_e}}}
        m = h.build_model_from_yaml(str(ROOT / "examples" / "krusell_smith.yaml"), overrides=ov)
        g = np.load(fx)
        if np.array_equal(g["a_grid"], m.heterogeneity["wealth"].grid) and np.array_equal(g["Pi"], m.heterogeneity["productivity"].transition):
            ss = h.SteadyState({k: float(g[f"var_{k}"]) for k in m.variables}, {"KD": g["policy"]}, None, g["D"], g["value"])
            return m, ss
    m, ss, _ = ks_setup(n_a, n_e, T)
    return m, ss


def check_timed_output(m, ss, x, dx_host, agg_gpu, dagg_gpu):
    """the output of the LAST timed step against the CPU oracle (value and the first tangent column; oracle = checker only)."""
    from oracle.oracle import Oracle
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    orc = Oracle(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons)
    P = m.compspec.T - 1
    xr = np.zeros((P, 2)); xw = np.zeros((P, 2))
    xr[:, 0], xw[:, 0] = x[2], x[3]
    xr[:, 1], xw[:, 1] = dx_host[0, :, 0], dx_host[1, :, 0]
    st, oagg, _ = orc.household_block(xr, xw, ss.value, ss.D, 1)
    return {"oracle_status": int(st),
            "agg_max_rel_err": float(np.max(np.abs(agg_gpu - oagg[:, 0])) / np.abs(oagg[:, 0]).max()),
            "dagg_col0_max_rel_err": float(np.max(np.abs(dagg_gpu[:, 0] - oagg[:, 1])) / np.abs(oagg[:, 1]).max()),
            "tolerance": 1e-10}


def cpu_baseline(m, ss, x, Z, budget_s=6.0):
    """the reference-style CPU path (oracle: dual numbers, primal recomputed on every JVP,
    NewtonRaphson.jl:95) timed on a bounded sample of the same workload: on ONE host core (the
    reference is single-threaded) and, beside it, on all the host cores this process may use, one
    single-tangent JVP per thread at a time (tangent directions are independent)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    P = m.compspec.T - 1
    shape = f"{wd.n}x{pd_.n} grid, T={m.compspec.T}"

    def worker(seed, deadline, cap):
        orc = Oracle(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons)
        rng = np.random.default_rng(seed)
        n = 0
        while n < cap and (n == 0 or time.perf_counter() < deadline):
            y = rng.standard_normal((4, P, 1))
            orc.ks_jvp(x, y, Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)   # ctypes call: releases the GIL
            n += 1
        return n

    t0 = time.perf_counter()
    n1 = worker(1, t0 + budget_s, 64)
    el1 = time.perf_counter() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # the GPU box gives one GPU's job a 16-CPU share
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        nall = sum(ex.map(lambda k: worker(100 + k, t0 + budget_s, 64), range(cores)))
    elall = time.perf_counter() - t0
    return {"value": n1 / el1, "unit": "JVPs/s", "cores": 1, "kind": "port",
            "sample": f"{n1} single-tangent JVPs (dual-number pipeline incl. primal, {shape}) in {el1:.1f} s on 1 core",
            "all_cores": {"value": nall / elall, "unit": "JVPs/s", "cores": cores, "kind": "port",
                          "sample": f"{nall} single-tangent JVPs, one per thread at a time, in {elall:.1f} s on {cores} threads"}}


def model_ceiling(G, N, n_e):
    """What this arithmetic can reach on this chip per period and sweep ("the path is fp64-issue-bound before
    it is HBM-bound"): the larger of the HBM time of the algorithmic bytes and the issue time of the fp64 work — about 80 lane-
    instructions per point and direction at n_e = 11 (10 Y half, 6 + 2 n_e X half / mixing, the rest index, LDS and select
    instructions; counters: profiles/r03n256_*), 1 024 SIMDs issuing one wave-instruction (64 lanes) per 4 clocks at 2.4 GHz —
    plus, at narrow batches, the floor of a dependent period (2.5 us: one barrier + one L2 round trip + the drain of the state
    stores, the persistent sweeps' stamps; a launch per period has 6.5-7 us)."""
    bytes_per_period = G * 8 * (1 + N)
    t_hbm = bytes_per_period / (HBM_PEAK_GBS * 1e9)
    lane_instr = (58 + 2 * n_e) * G * N
    t_issue = lane_instr / (1024 * 64 * 2.4e9 / 4)
    t_floor = 2.5e-6
    t = max(t_hbm, t_issue, t_floor)
    return {"GBs": bytes_per_period / t / 1e9, "frac_of_hbm_peak": bytes_per_period / t / 1e9 / HBM_PEAK_GBS,
            "bound": "hbm" if t == t_hbm else ("fp64 issue" if t == t_issue else "dependent-period latency"),
            "inputs": {"hbm_us": 1e6 * t_hbm, "fp64_issue_us": 1e6 * t_issue, "period_floor_us": 1e6 * t_floor,
                       "lane_instructions_per_point_direction": 58 + 2 * n_e}}


def kernel_source_sha16():
    """hash of the device sources the library is built from: profiles/pmc_latest.json records the one its counters were
    collected on (scripts/update_pmc_latest.py), `roofline.traffic_stale` says when they no longer match."""
    import hashlib
    h_ = hashlib.sha256()
    for f in sorted((ROOT / "julia-newtonraphsonhank_amd" / "csrc").glob("*.h*")):
        h_.update(f.read_bytes())
    return h_.hexdigest()[:16]


def extra_measurements(hb, d_x, P, N, dev, torch_stream=None):
    """second half of the BASELINE metric ("wall-clock to converged path, Krusell-Smith T=300") on
    configs[1] (500x4), and the JVP rate of a wider tangent batch on the headline grid."""
    import torch
    from examples.solve_transition import solve
    extra = {"converged_path": []}
    # two independent N-wide batches in flight (two contexts, two streams): what a Jacobian assembly, whose column
    # batches do not depend on each other, gets out of the latency-bound per-period launches. Measured first: HIP maps
    # streams onto a few hardware queues in creation order, and with the contexts the solves below create the two
    # streams would share one (and serialise).
    # (Both on the launch schedule: persistent sweeps own the chip one at a time and are ordered behind one another.)
    torch.cuda.synchronize()
    prev = os.environ.get("HANK_SCHEDULE")
    os.environ["HANK_SCHEDULE"] = "launch"
    try:
        hbs = (hb.clone(), hb.clone())
    finally:
        if prev is None:
            os.environ.pop("HANK_SCHEDULE", None)
        else:
            os.environ["HANK_SCHEDULE"] = prev
    try:
        bufs = [(torch.randn(2 * P * N, dtype=torch.float64, device=dev), torch.empty(P, dtype=torch.float64, device=dev),
                 torch.empty(P * N, dtype=torch.float64, device=dev)) for _ in range(2)]
        def both():
            for h_, (dx_, ag_, out_) in zip(hbs, bufs):
                h_.primal_jvp_dev(d_x.data_ptr(), dx_.data_ptr(), N, ag_.data_ptr(), out_.data_ptr())
        both(); hbs[0].sync(); hbs[1].sync()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            both()
        hbs[0].sync(); hbs[1].sync()
        el = (time.perf_counter() - t0) / reps
        extra["two_batches_in_flight"] = {"tangents": 2 * N, "JVPs_per_s": 2 * N / el, "ms_per_pair": 1e3 * el, "schedule": "launch-per-period"}
    finally:
        for h_ in hbs:
            h_.sync(); h_.close()
    # wider batch on the same context: N = 256 tangents in one dual-sweep pass
    Nw = 256
    d_dx = torch.randn(2 * P * Nw, dtype=torch.float64, device=dev)
    d_out = torch.empty(P * Nw, dtype=torch.float64, device=dev)
    d_agg = torch.empty(P, dtype=torch.float64, device=dev)
    hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), Nw, d_agg.data_ptr(), d_out.data_ptr())
    hb.sync()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), Nw, d_agg.data_ptr(), d_out.data_ptr())
    hb.sync()
    el = (time.perf_counter() - t0) / reps
    tm = hb.last_timings()
    G8 = hb.G * 8
    fam = hb.info()["last_tangent_family_name"]
    if fam == "on-chip-wide":
        # one workgroup per direction, ONE launch per sweep (csrc/hank_wide.h): a launch moves the policy partials of every period
        # and direction once, P*G*8*N bytes; the Float64 sweeps (hank_primal, XCD-local persistent) run before them and are timed
        # in ms_per_step
        kb, kf, np_, per, kpre = "tangent_backward", "tangent_forward", 1, P * Nw, "k_wide_"
    elif hb.stats()["schedule"] == 1:
        kb, kf, np_ = "tangent_backward", "tangent_forward", tm["tangent_backward"]["launches"]
        per, kpre = P * Nw, "k_xtan_"
    else:
        kb, kf, np_ = "dual_backward", "dual_forward", 1
        per, kpre = tm["dual_backward"]["launches"] * (1 + Nw), "k_fused_"
    b_alg_w = 2 * P * G8 * (1 + Nw)
    slow = kb if tm[kb]["ms"] >= tm[kf]["ms"] else kf
    ach_w = G8 * per / (1e-3 * tm[slow]["ms"]) / 1e9
    # fabric traffic of that kernel from the counter passes of scripts/dev_pmc_wide.sh (same hash rule as the headline's)
    traffic_w, stale_w, pmc_w = None, None, None
    try:
        rec = json.loads((ROOT / "profiles" / "pmc_latest.json").read_text()).get("ks_2000x11_T300_N256", {})
        pmc_w = rec.get(kpre + ("back" if slow == kb else "fwd"))
        if pmc_w:
            traffic_w, stale_w = pmc_w["hbm_bytes"], rec.get("kernel_source_sha16") != kernel_source_sha16()
    except (OSError, ValueError, KeyError):
        pass
    # the y-iteration's pattern at this width: JVP batches at the recorded primal
    hb.jvp_dev(d_dx.data_ptr(), Nw, d_out.data_ptr()); hb.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        hb.jvp_dev(d_dx.data_ptr(), Nw, d_out.data_ptr())
    hb.sync()
    el_j = (time.perf_counter() - t0) / reps
    extra["wide_batch"] = {"tangents": Nw, "JVPs_per_s": Nw / el, "ms_per_step": 1e3 * el, "passes": np_, "family": fam,
                           "jvp_at_recorded_primal": {"JVPs_per_s": Nw / el_j, "ms_per_batch": 1e3 * el_j},
                           "sweeps_ms": {k: round(v["ms"], 4) for k, v in tm.items() if v["ms"] > 0},
                           "backward_sweep_GBs": G8 * per / (1e-3 * tm[kb]["ms"]) / 1e9,
                           "forward_sweep_GBs": G8 * per / (1e-3 * tm[kf]["ms"]) / 1e9,
                           # the same object as the headline's, for the slower sweep of the wide batch (HIP events on the library's stream)
                           "roofline": {"bound": "hbm", "kernel": kpre + ("back" if slow == kb else "fwd"),
                                        "achieved": ach_w, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_w / HBM_PEAK_GBS, "traffic": traffic_w,
                                        "traffic_stale": stale_w, "traffic_detail": pmc_w, "bytes_per_launch": G8 * per,
                                        "sweep_ms": tm[slow]["ms"], "launches": tm[slow]["launches"], "model_ceiling": model_ceiling(hb.G, Nw, hb.n_e)},
                           "whole_batch": {"B_alg_bytes": b_alg_w, "achieved_GBs": b_alg_w / el / 1e9, "frac_of_hbm_peak": b_alg_w / el / 1e9 / HBM_PEAK_GBS}}
    # the y-iteration's access pattern (NewtonRaphson.jl:91-111): ONE primal, then JVP batches at that record
    hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
    Nj = N
    d_dxj = torch.randn(2 * P * Nj, dtype=torch.float64, device=dev)
    d_outj = torch.empty(P * Nj, dtype=torch.float64, device=dev)
    hb.jvp_dev(d_dxj.data_ptr(), Nj, d_outj.data_ptr()); hb.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        hb.jvp_dev(d_dxj.data_ptr(), Nj, d_outj.data_ptr())
    hb.sync()
    el = (time.perf_counter() - t0) / reps
    extra["jvp_at_recorded_primal"] = {"tangents": Nj, "JVPs_per_s": Nj / el, "ms_per_batch": 1e3 * el}
    d_dx1 = torch.randn(2 * P, dtype=torch.float64, device=dev)
    hb.jvp_dev(d_dx1.data_ptr(), 1, d_outj.data_ptr()); hb.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        hb.jvp_dev(d_dx1.data_ptr(), 1, d_outj.data_ptr())
    hb.sync()
    extra["single_tangent_jvp_ms"] = 1e3 * (time.perf_counter() - t0) / reps
    # BASELINE configs[1]: 500x4, T=300, ONE tangent at a recorded primal (its 9.6 MB policy sequence lives in the Infinity Cache:
    # the figure is an effective bandwidth from the algorithmic bytes)
    try:
        from conftest import ks_paths as _kp, ks_setup as _ks
        m1, ss1, _ = _ks(500, 4, 300)
        wd1, pd1 = m1.heterogeneity["wealth"], m1.heterogeneity["productivity"]
        hb1 = type(hb)(wd1.grid, pd1.grid, pd1.transition, m1.params.β, m1.params.γ, m1.params.borrow_cons, 300)
        hb1.set_boundary(ss1.value, ss1.D)
        x1, _ = _kp(m1, ss1, "x1", 0.01)
        P1 = 299
        hb1.primal(x1[2:4])
        d1 = torch.randn(2 * P1, dtype=torch.float64, device=dev)
        o1 = torch.empty(P1, dtype=torch.float64, device=dev)
        hb1.jvp_dev(d1.data_ptr(), 1, o1.data_ptr()); hb1.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            hb1.jvp_dev(d1.data_ptr(), 1, o1.data_ptr())
        hb1.sync()
        ms1 = 1e3 * (time.perf_counter() - t0) / 20
        tm1 = hb1.last_timings()
        slow1 = max(("tangent_backward", "tangent_forward"), key=lambda k: tm1[k]["ms"])
        bytes1 = P1 * 2000 * 8            # one sweep moves the policy partials of one direction: P G 8 bytes
        extra["config1_single_tangent"] = {
            "workload": "ks_500x4_T300_N1", "single_tangent_jvp_ms": ms1, "JVPs_per_s": 1e3 / ms1,
            "roofline": {"bound": "hbm", "kernel": "k_xtan_back" if slow1 == "tangent_backward" else "k_xfwd", "achieved": bytes1 / (1e-3 * tm1[slow1]["ms"]) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes1 / (1e-3 * tm1[slow1]["ms"]) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "sweep_ms": tm1[slow1]["ms"], "launches": tm1[slow1]["launches"],
                         "note": "ONE persistent launch per sweep; the 9.6 MB policy-partials sequence stays in the 256 MB Infinity Cache",
                         "model_ceiling": model_ceiling(2000, 1, 4)}}
        hb1.close()
    except Exception as e:              # noqa: BLE001
        extra["config1_single_tangent"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # BASELINE configs[4]: one-asset HANK 1000x7, T=500, Newton to convergence, both inner loops (examples/solve_hank.py)
    extra["config4_one_asset_hank"] = []
    for inner in ("fixed_point", "krylov"):
        try:
            from examples.solve_hank import solve as solve_hank
            res = solve_hank(1000, 7, 500, inner=inner)[0]
            res.pop("impact", None)
        except Exception as e:          # noqa: BLE001
            res = {"model": "one-asset HANK", "grid": "1000x7", "T": 500, "inner": inner, "error": f"{type(e).__name__}: {e}"[:300]}
        extra["config4_one_asset_hank"].append(res)
    # configs[1]'s grid with a mild shock and with RunMain.jl's Z_t = 1 + 0.8^t, then the headline grid
    # (the last two: the headline grid with the reference's damped fixed point, then with the opt-in Krylov inner loop)
    for n_a, n_e, shock, inner in ((500, 4, 0.01, "fixed_point"), (500, 4, 0.8, "fixed_point"), (2000, 11, 0.01, "fixed_point"), (2000, 11, 0.01, "krylov")):
        try:
            res, _ = solve(n_a, n_e, 300, shock, inner=inner)
        except Exception as e:          # noqa: BLE001 - report, do not kill the bench line
            res = {"grid": f"{n_a}x{n_e}", "T": 300, "shock": shock, "inner": inner, "error": str(e)[:200]}
        extra["converged_path"].append(res)
    # the whole RunMain sequence at the headline grid with NOTHING cached: steady state from the YAML guesses (value iteration
    # and stationary distribution as persistent launches on the device), then J̅ and Newton
    try:
        res, _ = solve(2000, 11, 300, 0.01, cold=True)
        res["yaml_to_converged_path_s"] = round(res["steady_state_s"] + res["wall_to_converged_path_s"], 3)
    except Exception as e:              # noqa: BLE001
        res = {"grid": "2000x11", "T": 300, "steady_state": "cold start", "error": str(e)[:200]}
    extra["converged_path"].append(res)
    return extra


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ks_2000x11_T300_N32", choices=sorted(WORKLOADS))
    ap.add_argument("--tangents", type=int, default=None, help="override the per-GPU tangent batch width")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--split", action="store_true", help="hank_primal + hank_jvp as two calls instead of the dual pass hank_primal_jvp")
    ap.add_argument("--mode", default="ranks", choices=("ranks", "devicegroup"),
                    help="ranks: one process per GPU + one RCCL all-gather per step (the driver's contract); "
                         "devicegroup: ONE process, one context per GPU (hank_create_on), no collective")
    ap.add_argument("--stand-in", action="store_true",
                    help="CPU test of the launcher and of the distributed plumbing: gloo, a stand-in step, no GPU, no product code")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N fresh ranks — one process per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would give them —
    BEFORE this process has made any GPU call (it never makes one), relay rank 0's JSON line, and fail if any rank fails.
    Children are started, never exec'ed into: no process that has touched a GPU is replaced."""
    import subprocess
    import tempfile
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")        # rank 0's stdout (a file, not a pipe: nobody reads it while the ranks run)
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=env, cwd=str(ROOT),
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others in a collective: watch all of them, end the rest when one fails
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.05)
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:      # noqa: BLE001
                p.kill()
        print(f"bench.py: rank {failed[0]} of {args.gpus} exited with status {failed[1]}", file=sys.stderr)
        return 1
    out0.seek(0)
    lines = [l for l in out0.read().splitlines() if l.startswith("{")]
    if not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def stand_in_main(args):
    """the launcher / rendezvous / barrier / max-over-ranks / all-gather plumbing of main() on CPU tensors over gloo with a
    stand-in step (tests/test_bench_launcher.py). Nothing of the product runs and nothing here is a measurement."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("BENCH_STANDIN_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    if use_dist:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_seen = dist.get_world_size() if use_dist else 1
    if n_seen != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {n_seen} rank(s)")
    P, N = 7, 4
    local = torch.full((P * N,), float(rank + 1), dtype=torch.float64)
    every = torch.empty(world * P * N, dtype=torch.float64) if use_dist else None

    def step():
        local.mul_(1.0)
        if use_dist:
            dist.all_gather_into_tensor(every, local)

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if use_dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        assert [float(every[r * P * N]) for r in range(world)] == [float(r + 1) for r in range(world)]
    if rank == 0:
        print(json.dumps({"metric": "stand-in (launcher test, not a measurement)", "value": world * N * args.steps / el, "unit": "JVPs/s",
                          "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "stand-in",
                          "config": {"workload": "stand-in", "parallelism": f"tangent-sharded x{world}", "backend": "gloo"}}), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def devicegroup_main(args):
    """--mode devicegroup: ONE process drives one context per GPU (hank_create_on; the form a single-process host such as
    the Julia reference uses). Every GPU runs the dual pass on its own 32 tangent columns
    through the asynchronous _dev entries; the results stay in each GPU's HBM — no collective."""
    import torch
    import hank_amd as h
    from conftest import ks_paths
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    ndev = torch.cuda.device_count()
    if args.gpus > ndev:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process sees {ndev} GPU(s)")
    wl = dict(WORKLOADS[args.workload])
    if args.tangents:
        wl["N"] = args.tangents
    n_a, n_e, T, N = wl["n_a"], wl["n_e"], wl["T"], wl["N"]
    P = T - 1
    m, ss = load_or_solve_ss(n_a, n_e, T)
    x, _ = ks_paths(m, ss, "x1", 0.01)
    xf = np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    first = h.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T, device=0)
    first.set_boundary(ss.value, ss.D)
    blocks = [first] + [first.clone(device=d) for d in range(1, args.gpus)]
    bufs = []
    for d, hb in enumerate(blocks):
        dev = torch.device("cuda", d)
        rng = np.random.default_rng(1000 + d)
        st = torch.cuda.Stream(device=dev)
        hb.set_stream(st.cuda_stream)
        bufs.append((torch.from_numpy(xf).to(dev), torch.from_numpy(rng.standard_normal(2 * P * N)).to(dev),
                     torch.empty(P, dtype=torch.float64, device=dev), torch.empty(P * N, dtype=torch.float64, device=dev), st))

    from hank_amd import hip as _hip
    d_all = torch.empty(len(blocks) * P * N, dtype=torch.float64, device=torch.device("cuda", 0))

    def step():
        for hb, (dx_, dy_, ag_, out_, _) in zip(blocks, bufs):
            hb.primal_jvp_dev(dx_.data_ptr(), dy_.data_ptr(), N, ag_.data_ptr(), out_.data_ptr())
        # the (P, N) blocks of all GPUs assembled in GPU 0's memory: peer copies over xGMI behind each context's sweeps (hank_gather_columns)
        _hip.gather_columns(blocks, [b[3].data_ptr() for b in bufs], [N] * len(blocks), d_all.data_ptr())

    def fence():
        for hb in blocks:
            hb.sync()

    for _ in range(args.warmup):
        step()
    for hb in blocks:
        hb.check()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    for hb in blocks:
        hb.check()
    out = {"metric": "sequence-space JVPs/sec (household block: BackwardIteration+ForwardIteration+aggregation, Krusell-Smith T=300)",
           "value": args.gpus * N * args.steps / el, "unit": "JVPs/s", "n_gpus": len(blocks), "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": args.workload, "grid": f"{n_a}x{n_e}", "T": T, "tangents_per_gpu": N,
                      "step": "1 dual pass per GPU: primal + N tangents (hank_primal_jvp_dev), the column blocks gathered on GPU 0 over xGMI (hank_gather_columns)",
                      "parallelism": f"one process, one context per GPU (hank_create_on) x{len(blocks)}"},
           # (a context that lost its persistent schedule while other processes still held the GPUs would show here, not as a slower number)
           "hank_stats": [{k: hb.stats()[k] for k in ("schedule", "fallbacks")} for hb in blocks]}
    print(json.dumps(out), flush=True)
    for hb in blocks[1:]:
        hb.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    launcher_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.mode == "ranks" and args.gpus > 1 and not launcher_env:
        raise SystemExit(self_launch(args, argv))        # before anything here touches a GPU
    if args.stand_in:
        return stand_in_main(args)
    if args.mode == "devicegroup":
        return devicegroup_main(args)

    import torch
    import torch.distributed as dist
    import hank_amd as h
    from conftest import ks_paths

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # under torch.distributed.run even at N=1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_seen = dist.get_world_size() if use_dist else 1     # the ranks RCCL actually has
    if n_seen != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {n_seen} rank(s): run `python bench.py --gpus N` "
                         "(it starts its own ranks) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")

    wl = dict(WORKLOADS[args.workload])
    if args.tangents:
        wl["N"] = args.tangents
    n_a, n_e, T, N = wl["n_a"], wl["n_e"], wl["T"], wl["N"]
    P, G = T - 1, n_a * n_e
    m, ss = load_or_solve_ss(n_a, n_e, T)
    x, Z = ks_paths(m, ss, "x1", 0.01)

    hb = h.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    # synthetic inputs, resident in HBM: household inputs (r_t, w_t) and this rank's tangent columns
    rng = np.random.default_rng(1000 + rank)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    dx_flat = rng.standard_normal(2 * P * N)
    d_dx = torch.from_numpy(dx_flat).to(dev)     # (2, P, N) column-major
    d_agg = torch.empty(P, dtype=torch.float64, device=dev)
    d_dagg = torch.empty(P * N, dtype=torch.float64, device=dev)        # (P, N) column-major
    d_all = torch.empty(world * P * N, dtype=torch.float64, device=dev) if use_dist else None

    # one HIP stream for the library's graphs AND for RCCL's stream hand-over: the all-gather is ordered
    # behind the sweeps without a host synchronisation
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    hb.set_stream(stream.cuda_stream)

    def step():
        if args.split:      # two calls: primal sweep, then the batched JVP at the cached primal
            hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
            hb.jvp_dev(d_dx.data_ptr(), N, d_dagg.data_ptr())
        else:               # one call, dual-sweep launches (what JVP(fullFunction, x, y) does in the reference)
            hb.primal_jvp_dev(d_x.data_ptr(), d_dx.data_ptr(), N, d_agg.data_ptr(), d_dagg.data_ptr())
        if use_dist:
            dist.all_gather_into_tensor(d_all, d_dagg)  # the only exchange on the path

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    hb.check()
    fence()
    st_before = hb.stats()
    primal_before = st_before["primal_sweeps"]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    hb.check()
    # every timed step ran its primal sweep (the primal memo of the host-pointer entry must never reach the timed region)
    st_timed = hb.stats()
    assert st_timed["primal_sweeps"] - primal_before == args.steps and st_timed["primal_memo_hits"] == 0, st_timed
    # ... on ONE schedule: a persistent sweep that could not form its groups moves the context to the launches silently (the host
    # entries fall back by themselves, hank_check for the _dev entries) — a number measured across such a switch is not a measurement
    if st_timed["fallbacks"] != st_before["fallbacks"] or st_timed["schedule"] != st_before["schedule"]:
        raise SystemExit(f"bench.py: the context changed schedule during the timed region ({st_before} -> {st_timed}): {hb._lib.hank_last_error(hb._ctx).decode()}")
    tm = hb.last_timings()          # HIP events on the library's stream around each sweep (last step)
    # which sweeps the timed call ran: the dual-sweep launches (dual_*), or persistent sweeps (the Float64 backward sweep — with the
    # partials in it when the batch is one pass: k_xdual_back — then the tangent sweeps; the forward one carries D_t: k_xfwd<D,true>)
    split_keys = tm["dual_backward"]["ms"] <= 0.0
    sweeps = {k: 0.0 for k in tm if tm[k]["ms"] > 0.0 and (k.startswith("dual") != split_keys)}
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # live per-sweep device times over a few extra (untimed) steps, for the roofline object
    reps = 5
    acc = {k: 0.0 for k in sweeps}
    for _ in range(reps):
        step()
        tmi = hb.last_timings()
        for k in acc:
            acc[k] += tmi[k]["ms"] / reps
    launches = {k: tm[k]["launches"] for k in tm}
    schedule = hb.stats()["schedule"]

    if rank == 0:
        total_jvps = world * N * args.steps
        ms_per_step = 1e3 * el / args.steps
        # dominant kernel: the per-period tangent kernels (k_tan_back / k_tan_fwd). One launch moves
        # the policy partials of ONE period for N directions: G*8*N algorithmic bytes
        # (B_alg = 2*P*G*8*(1+N) per batch = G*8 bytes per (sweep, period, direction)).
        persistent = split_keys and launches.get("tangent_forward", P) < P
        wide = hb.info()["last_tangent_family_name"] == "on-chip-wide"
        if wide:
            # the on-chip wide sweeps (csrc/hank_wide.h): one workgroup per direction, ONE launch per sweep; a launch moves the policy
            # partials of every period and direction once (written by k_wide_back, read by k_wide_fwd): P*G*8*N bytes. The Float64
            # sweeps run before them as XCD-local persistent sweeps and are part of ms_per_step.
            dom = max(("tangent_backward", "tangent_forward"), key=lambda k: acc[k])
            kname = {"tangent_backward": "k_wide_back", "tangent_forward": "k_wide_fwd"}[dom]
            bytes_per_launch = P * G * 8 * N
        elif persistent and not args.split and acc.get("tangent_backward", 0.0) < 0.05:
            # the persistent Dual pass: ONE launch per sweep, value and partials together in both (k_xdual_back writes the policy
            # and the policy partials of every period, k_xfwd<D, true> reads them): P*G*8*(1+N) algorithmic bytes per launch
            dom = max(("primal_backward", "tangent_forward"), key=lambda k: acc[k])
            kname = {"primal_backward": "k_xdual_back", "tangent_forward": "k_xfwd"}[dom]
            bytes_per_launch = P * G * 8 * (1 + N)
        elif persistent:
            # XCD-local persistent sweeps at a recorded primal: ONE launch carries a whole sweep (P periods). The kernels that
            # move the algorithmic bytes are the tangent sweeps (the policy-partials sequence: written once by k_xtan_back, read
            # once by k_xfwd): P*G*8*N_pass per launch; a batch wider than 32 runs as ceil(N/32) passes. The Float64
            # sweeps move P*G*8 bytes each and are latency-bound: reported beside, not hidden.
            dom = max(("tangent_backward", "tangent_forward"), key=lambda k: acc[k])
            kname = {"tangent_backward": "k_xtan_back", "tangent_forward": "k_xfwd"}[dom]
            bytes_per_launch = P * G * 8 * N / launches[dom]
        elif split_keys:
            dom = max(("tangent_backward", "tangent_forward"), key=lambda k: acc[k])
            kname = {"tangent_backward": "k_tan_back", "tangent_forward": "k_tan_fwd"}[dom]
            bytes_per_launch = G * 8 * N
        else:   # a dual-sweep launch advances the primal and the N tangents by one period: G*8*(1+N)
            dom = max(("dual_backward", "dual_forward"), key=lambda k: acc[k])
            kname = {"dual_backward": "k_fused_back", "dual_forward": "k_fused_fwd"}[dom]
            bytes_per_launch = G * 8 * (1 + N)
        avg_launch_s = 1e-3 * acc[dom] / launches[dom]
        achieved = bytes_per_launch / avg_launch_s / 1e9
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
        # WRITE_SIZE in separate runs, profiles/r01b_pmc_*): a profile artefact, not measured live
        pmc, traffic, traffic_stale = None, None, None
        pmc_file = ROOT / "profiles" / "pmc_latest.json"
        if pmc_file.exists() and N == WORKLOADS[args.workload]["N"]:
            try:
                rec = json.loads(pmc_file.read_text()).get(args.workload, {})
                pmc = rec.get(kname)
                traffic = pmc["hbm_bytes"] if pmc else None
                # the counters belong to the kernel sources they were collected on: a different hash = re-profile
                traffic_stale = (rec.get("kernel_source_sha16") != kernel_source_sha16()) if pmc else None
            except Exception:
                pmc, traffic, traffic_stale = None, None, None
        b_alg_batch = 2 * P * G * 8 * (1 + N)
        out = {
            "metric": "sequence-space JVPs/sec (household block: BackwardIteration+ForwardIteration+aggregation, Krusell-Smith T=300)",
            "value": total_jvps / el, "unit": "JVPs/s", "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "grid": f"{n_a}x{n_e}", "T": T, "tangents_per_gpu": N,
                       "step": ("1 primal sweep + 1 batched JVP of N tangents (hank_primal + hank_jvp)" if args.split else
                                "1 dual-sweep pass: primal + N tangents (hank_primal_jvp)") + (" + RCCL all-gather" if use_dist else ""),
                       "parallelism": f"tangent-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": traffic_stale, "traffic_detail": pmc,
                         "bytes_per_launch": bytes_per_launch, "avg_launch_us": 1e6 * avg_launch_s,
                         "model_ceiling": model_ceiling(G, N, n_e),
                         "schedule": "on-chip-wide" if wide else ("xcd-persistent" if persistent else "launch-per-period"),
                         # measured per-wave issue and wait shares of that kernel (a separate rocprofv3 --pmc pass, same hash rule as traffic)
                         "counters": (pmc or {}).get("counters"),
                         "note": "avg launch = HIP-event time of the sweep's kernels on the library's stream / launches"},
            # hank_stats of the timed context: schedule in use (2 = auto, 1 = persistent forced, 0 = launches), fallbacks from the
            # persistent sweeps to the launches (the run FAILS if one happens inside the timed region), primal sweeps the memo skipped
            "hank_stats": {k: st_timed[k] for k in ("schedule", "fallbacks", "primal_memo_hits", "primal_sweeps", "sweep_launches")},
            "tangent_family": hb.info()["last_tangent_family_name"],
            "whole_batch": {"B_alg_bytes": b_alg_batch, "achieved_GBs": b_alg_batch / (1e-3 * ms_per_step) / 1e9,
                            "frac_of_hbm_peak": b_alg_batch / (1e-3 * ms_per_step) / 1e9 / HBM_PEAK_GBS},
            "sweeps_ms": {k: round(acc[k], 4) for k in acc}, "launches": launches,
        }
        if not args.no_cpu_baseline and world == 1:
            step(); fence()          # the benched call once more, its outputs against the CPU oracle
            out["output_check"] = check_timed_output(m, ss, x, dx_flat.reshape((2, P, N), order="F"), d_agg.cpu().numpy(),
                                                     d_dagg.cpu().numpy().reshape((P, N), order="F"))
            out["cpu_baseline"] = cpu_baseline(m, ss, x, Z)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if not args.no_extra and world == 1:
            try:
                out["extra"] = extra_measurements(hb, d_x, P, N, dev, stream.cuda_stream)
            except Exception as e:      # noqa: BLE001 - the headline line must survive a failure of the side measurements
                out["extra"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if use_dist:
        # every rank gives its GPU back BEFORE the last barrier: the device-group child below runs persistent sweeps on all of
        # them, and a persistent sweep must find its XCD's CUs free (a rank still running kernels there could cost the child a
        # group, i.e. a fallback to the launches or a failed check; the child reports hank_stats of every context next to its number)
        hb.set_stream(None)
        hb.sync()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not args.no_extra:
            # the same partition with ONE process driving one context per GPU (no collective), measured in a fresh child once
            # the ranks are done with the GPUs (idle behind the barrier above): a failure there must not cost the headline line
            import subprocess
            try:
                cmd = [sys.executable, str(Path(__file__).resolve()), "--mode", "devicegroup", "--gpus", str(world), "--steps", str(args.steps),
                       "--warmup", str(args.warmup), "--workload", args.workload] + (["--tangents", str(args.tangents)] if args.tangents else [])
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
                r = subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                dg = json.loads(line[-1]) if (r.returncode == 0 and line) else {"error": (r.stderr or r.stdout)[-300:]}
                out.setdefault("extra", {})["devicegroup"] = {k: dg[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "config", "hank_stats", "error") if k in dg}
            except Exception as e:      # noqa: BLE001
                out.setdefault("extra", {})["devicegroup"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
