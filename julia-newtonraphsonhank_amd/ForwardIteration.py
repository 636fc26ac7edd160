"""ForwardIteration — same call surface as ForwardIteration.jl:253-311, body on the GPU.

    ForwardIteration(policy_seqs, model, ss_initial) -> {het_var: length-(T-1) vector}

D_t = Λ_exog · Λ_endog(policy_t) · D_{t-1} with Young's (2010) lottery, aggregate
dot(vec(policy_t), D_t) on the POST-transition distribution (ForwardIteration.jl:297-308).
Policy sequences that come straight from this package's BackwardIteration are still resident in
HBM and take the fused sweep; arbitrary user-supplied sequences are pushed through the granular
device step (hank_forward_step[_dual]) period by period.
"""
from __future__ import annotations

import numpy as np

from .BackwardIteration import PolicySequences, household_block
from .dual import Dual
from .GeneralStructures import SequenceModel, vars_of_type


def make_endogenous_transition(policy_mat, dim, n_exog: int):
    """Young's (2010) block-diagonal lottery matrix as scipy CSC (ForwardIteration.jl:37-78).
    Host helper for the steady-state solver (SteadyState.jl stays on the host)."""
    import scipy.sparse as sp

    policy_mat = np.asarray(policy_mat, dtype=np.float64)
    n_a, grid = dim.n, dim.grid
    p = policy_mat.reshape(-1, order="F")
    cols = np.arange(n_a * n_exog)
    eoff = (cols // n_a) * n_a
    m = np.searchsorted(grid, p, side="left")          # searchsortedfirst, 0-based
    lo_clamp, hi_clamp = m == 0, m >= n_a
    mi = np.clip(m, 1, n_a - 1)
    w = (p - grid[mi - 1]) / (grid[mi] - grid[mi - 1])
    rows = np.concatenate([eoff + np.where(lo_clamp, 0, np.where(hi_clamp, n_a - 1, mi - 1)), (eoff + mi)[~(lo_clamp | hi_clamp)]])
    vals = np.concatenate([np.where(lo_clamp | hi_clamp, 1.0, 1.0 - w), w[~(lo_clamp | hi_clamp)]])
    cc = np.concatenate([cols, cols[~(lo_clamp | hi_clamp)]])
    return sp.csc_matrix((vals, (rows, cc)), shape=(n_a * n_exog, n_a * n_exog))


def transition_step(policy_mat, D_prev, model: SequenceModel):
    """one period of distribution evolution on the GPU (ForwardIteration.jl:95-99); returns D_new
    as a length-G vector (a `Dual` when either input is one)."""
    hb = household_block(model)
    if isinstance(policy_mat, Dual) or isinstance(D_prev, Dual):
        N = policy_mat.N if isinstance(policy_mat, Dual) else D_prev.N
        pm = policy_mat if isinstance(policy_mat, Dual) else Dual.constant(policy_mat, N)
        dp = D_prev if isinstance(D_prev, Dual) else Dual.constant(np.asarray(D_prev, dtype=np.float64), N)
        Do, dDo, _, _ = hb.forward_step_dual(pm.v, pm.p, dp.v, dp.p.reshape(hb.n_a, hb.n_e, N, order="F"))
        return Dual(Do.reshape(-1, order="F"), dDo.reshape(-1, N, order="F"))
    Do, _ = hb.forward_step(policy_mat, D_prev)
    return Do.reshape(-1, order="F")


def ForwardIteration(policy_seqs, model: SequenceModel, ss_initial):
    T = model.compspec.T
    P = T - 1
    endog = [(n, d) for n, d in model.heterogeneity.items() if d.dim_type == "endogenous"]
    if len(endog) != 1:
        raise ValueError(f"ForwardIteration: exactly one endogenous dimension is currently supported (got {len(endog)})")
    het_keys = vars_of_type(model, "heterogeneous")
    hb = household_block(model)
    D0 = np.asarray(ss_initial.D, dtype=np.float64)

    pol_key = endog[0][1].policy_var
    outs = tuple(model.value_fn.outputs)
    fused = pol_key in het_keys and all(k in outs for k in het_keys)     # (the variables the device sweeps reduce: hank_get_het_outputs)
    if isinstance(policy_seqs, PolicySequences) and policy_seqs._pending is not None and fused:
        policy_seqs._run(D0)         # the deferred sweep of a 4-argument BackwardIteration: ONE pass, with the right D_0
    if isinstance(policy_seqs, PolicySequences) and getattr(hb, "_generation", None) == policy_seqs._generation \
            and fused and getattr(hb, "_last", None) is not None:
        # fused path: the device sweeps aggregate every heterogeneous variable of the family with the D_t they hold
        # (ForwardIteration.jl:303-307: each variable with ITS policy, all with the same D_t)
        last = hb._last
        if not np.array_equal(last["D0"], D0):
            # the backward sweep did not know ss_initial: redo the fused sweep with the right D_0
            hb.set_boundary(last["value"], D0)
            agg = hb.primal(last["xhh"])
            dagg = hb.jvp(last["dxhh"]) if last["dxhh"] is not None else None
            last.update(agg=agg, dagg=dagg, D0=D0)
            if "het" in last:
                last["het"] = hb.het_outputs(len(outs), last["dxhh"])
        agg, dagg = last["agg"], last["dagg"]
        if len(het_keys) == 1:
            return {pol_key: Dual(agg, dagg) if dagg is not None else agg}
        aggs, daggs = last["het"]
        return {k: (Dual(aggs[:, outs.index(k)].copy(), np.ascontiguousarray(daggs[:, outs.index(k), :])) if daggs is not None
                    else aggs[:, outs.index(k)].copy()) for k in het_keys}

    # generic path: explicit policy matrices, one granular device step per period
    seqs = {k: policy_seqs[k] for k in het_keys}
    first = seqs[het_keys[0]][0]
    is_dual = isinstance(first, Dual)
    N = first.N if is_dual else 0
    D = Dual.constant(D0, N) if is_dual else D0
    aggs = {k: (np.empty(P), np.zeros((P, N))) for k in het_keys}
    for t in range(P):
        pm = seqs[pol_key][t]
        if is_dual:
            Do, dDo, a, da = hb.forward_step_dual(pm.v, pm.p, D.v, D.p.reshape(hb.n_a, hb.n_e, N, order="F"))
            D = Dual(Do.reshape(-1, order="F"), dDo.reshape(-1, N, order="F"))
        else:
            Do, a = hb.forward_step(pm, D)
            D = Do.reshape(-1, order="F")
        for k in het_keys:
            if k == pol_key:
                aggs[k][0][t] = a
                if is_dual:
                    aggs[k][1][t] = da
            else:  # another heterogeneous variable aggregated against the same D_t
                pk = seqs[k][t]
                if is_dual:
                    pv, pp = pk.v.reshape(-1, order="F"), pk.p.reshape(-1, N, order="F")
                    aggs[k][0][t] = pv @ D.v
                    aggs[k][1][t] = pp.T @ D.v + D.p.T @ pv
                else:
                    aggs[k][0][t] = np.asarray(pk).reshape(-1, order="F") @ D
    return {k: (Dual(v, p) if is_dual else v) for k, (v, p) in aggs.items()}
