"""Sequence-space Jacobian at the steady state — the J̅ that y_Iteration uses as GMRES operator.

The reference assembles it with Boehl's block-Toeplitz recursion from n_endog ForwardDiff JVPs
through BackwardIteration plus n_endog Zygote pullbacks through ForwardIteration
(SteadyStateJacobian.jl:41-410) and validates columns against full-pipeline JVPs
(test_SteadyState.jl:194-231, abs tol 1e-5). With a batched native JVP the same matrix is obtained
from the same Toeplitz structure (method="toeplitz": n_hh backward tangent sweeps + one forward push + the
expectation vectors on the GPU, `hank_fake_news`) or, as a check and for non-stationary paths, directly as
n = n_endog·(T-1) unit-tangent JVPs of the full pipeline (method="columns") — no reverse mode either way.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .GeneralStructures import SequenceModel, vars_of_type
from .NewtonRaphson import LinearizedFunction
from ._threads import host_algebra


class DenseJacobian(np.ndarray):
    """J̅ as the dense matrix it is (the reference wraps it in `sparse(...)`, SteadyStateJacobian.jl:64: at n ≈ 1.2k–3.5k with
    full household blocks the conversion costs more than the assembly). `.toarray()` keeps the sparse-matrix surface."""

    def toarray(self):
        return np.asarray(self)


def household_jacobian(F, Dv):
    """d agg_t / d xhh_{k,s} at the steady state from the fake-news matrix F (P, P, n_hh) and the direct term Dv (P, n_hh)
    (hank_fake_news): the Toeplitz recursion J[t, s] = J[t-1, s-1] + F[t, s] of SteadyStateJacobian.jl:363-371, with
    J[0, s] = Dv[s] + F[0, s]. Returns (n_hh, P, P)."""
    P, _, n_hh = F.shape
    J = np.empty((n_hh, P, P))
    for k in range(n_hh):
        Jk = J[k]
        Jk[0, :] = Dv[:, k] + F[0, :, k]
        for t in range(1, P):
            Jk[t, 0] = F[t, 0, k]
            Jk[t, 1:] = Jk[t - 1, :-1] + F[t, 1:, k]
    return J


def direct_blocks(model: SequenceModel, ss):
    """d R_t / d xMat[:, t + o] at the steady state for every offset o in -max_lag..max_lead: the `blocks` of
    SteadyStateJacobian.jl:124-145, from ONE evaluation of the compiled equations on a (1 + max_lag + max_lead)-column
    steady-state matrix with a unit tangent in every entry. Returns {o: (n_eq, n_v)}."""
    from .Aggregation import Residuals
    from .dual import Dual
    from .GeneralStructures import var_names
    cs = model.compspec
    keys = var_names(model)
    W = 1 + cs.max_lag + cs.max_lead
    xv = np.tile(np.array([float(ss.vars[k]) for k in keys])[:, None], (1, W))
    n_v = len(keys)
    xp = np.zeros((n_v, W, n_v * W))
    for r in range(n_v):
        for c_ in range(W):
            xp[r, c_, r * W + c_] = 1.0
    res = Residuals(Dual(xv, xp), model)        # n_eq residuals of the one middle period
    p = np.asarray(res.p).reshape(len(model.equations), n_v, W)
    return {c_ - cs.max_lag: p[:, :, c_] for c_ in range(W)}


@host_algebra
def getSteadyStateJacobian(ss, model: SequenceModel, chunk: int = 512, drop_tol: float = 0.0, device_batch: int = 256, group=None,
                           method: str = "toeplitz", sparse: bool = False):
    """n x n Jacobian of F at the constant steady-state path (SteadyStateJacobian.jl:41-65); `sparse=True` returns the
    reference's CSC form, the default a dense matrix with a `.toarray()`.

    method="toeplitz" (default): the reference's own structure — direct blocks of the equations (:124-145) plus the
    household block's Jacobian from its Toeplitz recursion (:187-256, :293-323, :358-387), obtained on the device from
    n_hh backward tangent sweeps seeded at the last period (`hank_fake_news`) instead of n unit tangents.
    method="columns": n unit-tangent JVPs of the full pipeline (`chunk` per call; only those that move the household
    inputs reach the GPU, in device batches padded to `device_batch`), sharded over the ranks of `group` — exact at ANY
    primal path, and the check of the other branch (they agree to ~1e-10 at a converged steady state)."""
    cs = model.compspec
    if len(model.equations) != cs.n_endog:
        raise AssertionError(f"System is not square: {len(model.equations)} equations but {cs.n_endog} endogenous "
                             "variables. Newton-Raphson requires n_eq == n_endog.")
    P = cs.T - 1
    n = cs.n_endog * P
    endog_keys = vars_of_type(model, "endogenous")
    exog_keys = vars_of_type(model, "exogenous")
    x_ss = np.tile(np.array([ss.vars[k] for k in endog_keys]), P)
    exog_ss = {k: np.full(P, float(ss.vars[k])) for k in exog_keys}
    lin = LinearizedFunction(x_ss, exog_ss, model, ss, ss)
    if method == "toeplitz" and len(lin.het) > 1:
        # hank_fake_news carries the expectation vectors of the policy variable's aggregate only: a model with a second
        # heterogeneous variable takes the unit-tangent columns (exact at any path; n JVPs in device batches)
        method = "columns"
    if method == "toeplitz":
        from .GeneralStructures import var_names
        keys = var_names(model)
        n_eq, n_endog = len(model.equations), cs.n_endog
        Jhh = household_jacobian(*lin.hb.fake_news())                  # (n_hh, P, P)
        B = direct_blocks(model, ss)
        J4 = np.zeros((P, n_eq, P, n_endog))                           # [t, eq, s, j]  ->  row eq + n_eq t, column j + n_endog s
        tt = np.arange(P)
        for o, Bo in B.items():
            ok = (tt + o >= 0) & (tt + o < P)
            # the equations' own dependence on x_{t+o}
            for j, k in enumerate(endog_keys):
                col = Bo[:, keys.index(k)]
                if np.any(col != 0.0):
                    J4[tt[ok], :, tt[ok] + o, j] += col[None, :]
            # through the aggregates: d R_t / d agg_{t+o} * d agg_{t+o} / d xhh_{k, s}
            for h in lin.het:
                colh = Bo[:, keys.index(h)]
                if not np.any(colh != 0.0):
                    continue
                for kk, name in enumerate(model.value_fn.household_inputs):
                    if name not in endog_keys:
                        continue                                       # an exogenous household input: no column of J̅
                    M = np.zeros((P, P))
                    M[tt[ok]] = Jhh[kk][tt[ok] + o]
                    jcol = endog_keys.index(name)
                    for q in np.flatnonzero(colh):                      # (the few equations the aggregate enters)
                        J4[:, q, :, jcol] += colh[q] * M
        J = J4.reshape(P * n_eq, P * n_endog)
    elif method == "columns":
        # unit tangents in chunks; under torch.distributed with W > 1 ranks the chunks of a pass are shared out over the
        # ranks (one GPU each) and all-gathered (parallel.assemble_columns) — every rank ends with the whole matrix
        from .parallel import assemble_columns
        J = assemble_columns(lambda E: lin.jvp(E, pad_to=device_batch), n, chunk, group)
    else:
        raise ValueError(f"unknown method {method!r} (toeplitz | columns)")
    if drop_tol > 0:
        J[np.abs(J) < drop_tol] = 0.0
    return sp.csc_matrix(J) if sparse else np.ascontiguousarray(J).view(DenseJacobian)
