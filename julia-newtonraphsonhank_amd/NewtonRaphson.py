"""Newton–Raphson for the perfect-foresight path — Python mirror of NewtonRaphson.jl (Boehl 2021).

Same call surface (`NewtonRaphsonHANK(x_0, J̅, exog_paths, mod, ss_initial, ss_ending; ε)`,
`y_Iteration(...)`); the hot path inside `fullFunction` — BackwardIteration → ForwardIteration —
runs on the GPU. One difference in *cost*, not in results: the reference re-runs the whole primal
pipeline inside every JVP (NewtonRaphson.jl:95, GeneralStructures.jl:546-547); here the primal
sweep at x is recorded once by the Float64 call `fullFunction(x)` (:91) and every JVP of the inner
loop reuses that linearisation (hank_jvp).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse.linalg as spla

from .Aggregation import Residuals
from .BackwardIteration import BackwardIteration, household_block, household_inputs
from .dual import Dual
from .ForwardIteration import ForwardIteration
from .GeneralStructures import JVP, SequenceModel, assemble_full_xMat, vars_of_type
from ._threads import host_algebra


def make_fullFunction(exog_paths, mod: SequenceModel, ss_initial, ss_ending):
    """the closure of y_Iteration (NewtonRaphson.jl:77-83)."""

    def fullFunction(x_Vec):
        policy_seqs = BackwardIteration(x_Vec, exog_paths, mod, ss_ending, ss_initial=ss_initial)
        agg_seqs = ForwardIteration(policy_seqs, mod, ss_initial)
        padded_xMat = assemble_full_xMat(x_Vec, agg_seqs, exog_paths, mod, ss_initial, ss_ending)
        return Residuals(padded_xMat, mod)

    return fullFunction


class LinearizedFunction:
    """F(x) evaluated once + cheap J(x)·y products at that fixed x (the y-iteration's access pattern,
    NewtonRaphson.jl:91-105). Tangent batches (n, N) go through ONE hank_jvp."""

    exact_residual_layer = False        # True: J(x)·y through the compiled equations under a Dual at every call (tests compare the two)

    def __init__(self, x, exog_paths, mod: SequenceModel, ss_initial, ss_ending):
        self.x = np.asarray(x, dtype=np.float64)
        self.mod, self.exog_paths, self.ss_initial, self.ss_ending = mod, exog_paths, ss_initial, ss_ending
        self.hb = household_block(mod)
        xhh, _ = household_inputs(self.x, exog_paths, mod)
        self._xhh = xhh
        het = vars_of_type(mod, "heterogeneous")
        self.het = het
        outs = tuple(mod.value_fn.outputs)
        self._out_idx = [outs.index(k) for k in het]        # device output of each heterogeneous variable (hank_get_het_outputs)
        self._n_out = len(outs) if len(het) > 1 else 1
        self._record_primal()
        self._xMat = assemble_full_xMat(self.x, {k: self.aggs[:, j] for k, j in zip(het, self._out_idx)}, exog_paths, mod, ss_initial, ss_ending)
        self.Fx = Residuals(self._xMat, mod)
        self._Rx = self._Ragg = None        # the residual layer's linearisation at this x, built at the first jvp

    def _linearise_residuals(self):
        """∂R/∂x and ∂R/∂agg at this x as sparse maps, from ONE evaluation of the compiled equations under a Dual: x — and with
        it the residual layer's linearisation — is fixed for the whole y-iteration (NewtonRaphson.jl:91-111), so the ≈ 21 inner
        iterations need not re-seed a Dual and re-evaluate the equations under it each time (Aggregation.jl:20-22,
        GeneralStructures.jl:329-377: the padded matrix is a selection of x, the aggregates and constants). Direction
        (row r, colour k) perturbs variable r in every padded column c with c mod W == k; the W = 1 + max_lag + max_lead columns
        a period's equations read have distinct colours, so each direction is one entry of that period's Jacobian block."""
        import scipy.sparse as sp
        from .GeneralStructures import var_names
        mod, cs = self.mod, self.mod.compspec
        P, n_v, n_endog, max_lag, max_lead = cs.T - 1, cs.n_v, cs.n_endog, cs.max_lag, cs.max_lead
        W = 1 + max_lag + max_lead
        T_pad = P + max_lag + max_lead
        n_eq = len(mod.equations)
        xp = np.zeros((n_v, T_pad, n_v * W))
        cols = np.arange(T_pad)
        for r in range(n_v):
            xp[r, cols, r * W + cols % W] = 1.0
        res = Residuals(Dual(self._xMat, xp), mod)
        Pm = np.asarray(res.p).reshape(P, n_eq, n_v, W)                   # [t, q, r, colour]: rows are ordered q + n_eq t, directions r W + colour
        keys = var_names(mod)
        endog = {keys.index(k): j for j, k in enumerate(vars_of_type(mod, "endogenous"))}
        hetr = {keys.index(k): j for j, k in enumerate(self.het)}
        rx, cx, vx, ra, ca, va = [], [], [], [], [], []
        t = np.arange(P)
        for k in range(W):
            c = t + (k - t) % W                   # the padded column of colour k inside period t's window [t, t + W)
            s_ = c - max_lag                      # ... as a transition period (boundary columns are constants)
            ok = (s_ >= 0) & (s_ < P)
            for r in range(n_v):
                if r not in endog and r not in hetr:
                    continue                      # exogenous paths carry no tangent
                for q in range(n_eq):
                    v = Pm[:, q, r, k]
                    nz = ok & (v != 0.0)
                    if not nz.any():
                        continue
                    rows = q + n_eq * t[nz]
                    if r in endog:
                        rx.append(rows); cx.append(endog[r] + n_endog * s_[nz]); vx.append(v[nz])
                    else:
                        ra.append(rows); ca.append(hetr[r] * P + s_[nz]); va.append(v[nz])
        cat = lambda L, dt: np.concatenate(L) if L else np.zeros(0, dtype=dt)
        self._Rx = sp.csr_matrix((cat(vx, float), (cat(rx, int), cat(cx, int))), shape=(n_eq * P, n_endog * P))
        self._Ragg = sp.csr_matrix((cat(va, float), (cat(ra, int), cat(ca, int))), shape=(n_eq * P, len(self.het) * P))

    def _record_primal(self):
        """(re-)run the Float64 sweep at x on the model's device context and take ownership of its record: the
        context is shared by everything that touches this model, so the generation counter says whose primal
        it currently holds (BackwardIteration and other linearisations bump it too)."""
        hb = self.hb
        hb.set_boundary(self.ss_ending.value, self.ss_initial.D)
        self.agg = hb.primal(self._xhh)
        self.aggs = self.agg[:, None] if self._n_out == 1 else hb.het_outputs(self._n_out)[0]      # (P, outputs)
        hb._generation = getattr(hb, "_generation", 0) + 1
        hb._last = None          # an older PolicySequences must not take ForwardIteration's fused shortcut
        self._generation = hb._generation

    def jvp(self, y, pad_to: int | None = None):
        """J(x)·y for one tangent (n,) or a batch (n, N). Directions that do not move the household inputs
        (unit tangents in Y or KS, say) are not sent to the GPU — their household partials are zero; `pad_to`
        rounds the device batch up to a multiple (zero columns) so that repeated calls reuse one workspace."""
        y = np.asarray(y, dtype=np.float64)
        single = y.ndim == 1
        xd = Dual.seed(self.x, y[:, None] if single else y)
        _, dxhh = household_inputs(xd, self.exog_paths, self.mod)
        dxhh = np.asarray(dxhh, dtype=np.float64)
        if dxhh.ndim == 2:
            dxhh = dxhh[:, :, None]
        N = dxhh.shape[2]
        nz = np.flatnonzero(np.any(dxhh != 0.0, axis=(0, 1)))
        dagg = np.zeros((dxhh.shape[1], N))
        daggs = dagg[:, None, :] if self._n_out == 1 else np.zeros((dxhh.shape[1], self._n_out, N))      # (P, outputs, N)
        if len(nz):
            if getattr(self.hb, "_generation", None) != self._generation:
                # another linearisation / BackwardIteration used the model's context since: its record is not
                # this x's any more — restore ours (bit-identical: same kernels, same inputs) instead of
                # silently returning J(x_other)·y
                self._record_primal()
            sub = dxhh[:, :, nz]
            if pad_to and len(nz) % pad_to:
                padded = np.zeros(sub.shape[:2] + (-(-len(nz) // pad_to) * pad_to,))
                padded[:, :, :len(nz)] = sub
                sub = padded
            dagg[:, nz] = self.hb.jvp(sub)[:, :len(nz)]
            if self._n_out > 1:
                daggs[:, :, nz] = self.hb.het_outputs(self._n_out, sub)[1][:, :, :len(nz)]
        if self.exact_residual_layer:           # the reference's way: re-evaluate the equations under the Dual every time
            agg = {k: Dual(self.aggs[:, j], np.ascontiguousarray(daggs[:, j, :])) for k, j in zip(self.het, self._out_idx)}
            res = Residuals(assemble_full_xMat(xd, agg, self.exog_paths, self.mod, self.ss_initial, self.ss_ending), self.mod)
            return res.p[:, 0].copy() if single else res.p.copy()
        if self._Rx is None:
            self._linearise_residuals()
        Y = y[:, None] if single else y
        out = self._Rx @ Y + self._Ragg @ np.concatenate([daggs[:, j, :] for j in self._out_idx], axis=0)     # rows: het variable, period
        return out[:, 0].copy() if single else out


def _gmres(J, b, x0):
    """IterativeSolvers.gmres!(x, A, b) defaults: restart = min(20, n), reltol = sqrt(eps),
    warm start from x (NewtonRaphson.jl:97-98). `maxiter` there counts Krylov steps, scipy's counts
    restart cycles."""
    n = len(b)
    restart = min(20, n)
    x, _ = spla.gmres(J, b, x0=x0, rtol=np.sqrt(np.finfo(float).eps), atol=0.0, restart=restart,
                      maxiter=max(1, n // restart))
    return x


_LU_CACHE = []          # [(J, lu)]: holds the factored matrix itself — an id() alone can be recycled by a new J̅


def _lu_solver(J):
    """J̅ obtained from batched JVPs is a dense n x n matrix (n ≈ 1.2k): one LU factorisation, reused
    by every inner and outer iteration, replaces the two GMRES solves per inner iteration — same
    J̅⁻¹·b up to rounding, milliseconds instead of seconds on the host. (A deviation from
    NewtonRaphson.jl:97-98, which runs two warm-started gmres! solves: `linear_solver="gmres"` is that branch.)"""
    import scipy.linalg as sla
    if not (_LU_CACHE and _LU_CACHE[0][0] is J):
        A = J.toarray() if hasattr(J, "toarray") else np.asarray(J)
        _LU_CACHE[:] = [(J, sla.lu_factor(A))]
    lu = _LU_CACHE[0][1]
    return lambda b: sla.lu_solve(lu, b)


_INV_CACHE = []         # [(J, apply)]
_WARM = {"done": False}


def warm_linear_solver(n_unknowns: int):
    """The device linear-algebra libraries behind `_device_inverse_solver` (rocSOLVER / rocBLAS through torch) take 0.15-0.35 s to
    start the first time a process uses them (scripts/dev_inv_cost.py): shared objects and code objects being loaded. A model with
    >= 2 000 unknowns will need them, so `find_ss` calls this BEFORE its own solve begins: one small inverse and one product on the
    device, synchronously. (Round 4 ran it in a background thread beside the solve. The steady state's inner fixed points are
    persistent sweeps that assume the process has the GPU to itself — a library kernel that lands between their workgroups can
    keep a group from forming, and the context then moves to the per-period launches for good, silently: hank_stats' `fallbacks`.
    The start-up is paid once per process either way; here it can no longer cost a schedule.) Idempotent; a no-op without a GPU."""
    if n_unknowns < 2000 or _WARM["done"]:
        return
    _WARM["done"] = True
    try:
        import torch
        if torch.cuda.is_available():
            A = torch.eye(512, dtype=torch.float64, device="cuda") + 0.001
            (torch.linalg.inv(A) @ A[:, :1]).cpu()
            torch.cuda.synchronize()
    except Exception:       # noqa: BLE001 - a warm-up must never fail a solve
        pass


def release_linear_solver():
    """drop the factorisations kept for a J̅ (the host LU and the device inverse: n² fp64 in HBM, 97 MB at n = 3 493).
    NewtonRaphsonHANK calls it when it returns; a caller that drives y_Iteration itself calls it when done with a J̅."""
    _LU_CACHE[:] = []
    _INV_CACHE[:] = []


def _device_inverse_solver(J, device=None):
    """J̅⁻¹·b as ONE dense matrix-vector product on the GPU (a library GEMV: torch → rocBLAS) with the explicit inverse, formed once
    per J̅ (torch.linalg.inv on the device): at the one-asset HANK size (n = 3 493) the host's two triangular solves read 97 MB per
    inner iteration and took as long as the device's tangent sweeps (2.1 ms against 2.9 ms, scripts/dev_profile_newton.py). J̅⁻¹
    is a preconditioner here — the fixed point of the y-iteration is J(x)⁻¹F(x) whatever its accuracy. Returns None without a GPU."""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
    except Exception:       # noqa: BLE001
        return None
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))       # the MODEL's GPU (hank_create_on), not torch's current one
    if not (_INV_CACHE and _INV_CACHE[0][0] is J and _INV_CACHE[0][2] == dev):
        import time
        t0 = time.perf_counter()
        A = J.toarray() if hasattr(J, "toarray") else np.asarray(J)
        Ainv = torch.linalg.inv(torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(dev))
        (Ainv[:1] @ Ainv[:, :1]).cpu()      # the first product of a process also loads rocBLAS: count it with the set-up, not with an iteration
        # once per J̅ (and, the first time in a process, the linear-algebra libraries' own start-up: 0.15-0.35 s against 40 ms warm at
        # n = 3 493, scripts/dev_inv_cost.py): reported next to the solve times, examples/solve_hank.py
        y_Iteration.setup_s = getattr(y_Iteration, "setup_s", 0.0) + (time.perf_counter() - t0)

        def apply(b, Ainv=Ainv, dev=dev):
            return (Ainv @ torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).to(dev)).cpu().numpy()

        _INV_CACHE[:] = [(J, apply, dev)]
    return _INV_CACHE[0][1]


@host_algebra
def y_Iteration(J̅, x, y0, exog_paths, mod: SequenceModel, ss_initial, ss_ending, *, precond=None,
                α: float | None = None, γ: float = 1.5, ε: float = 1e-9, verbose: bool = False, max_inner: int = 10_000,
                linear_solver: str = "auto", inner: str = "fixed_point"):
    """inner iteration for the search direction y with J(x)·y = F(x) (NewtonRaphson.jl:65-114).

    inner="fixed_point" (default, the reference): y ← y + α·J̅⁻¹(F(x) − J(x)·y). The reference accepts α (:72) and
    overrides it with 0.5 (:102): `α=None` is that; a value passed here is honoured (α = 1 is the undamped iteration).
    inner="krylov" (opt-in, not in the reference): GMRES on J(x) with J̅⁻¹ as right preconditioner — the same fixed point
    (the exact Newton step) in a handful of JVPs instead of the ≈ 21 that α = 0.5 needs to contract 1e-9 by halves."""
    lin = LinearizedFunction(x, exog_paths, mod, ss_initial, ss_ending)
    y = np.asarray(y0, dtype=np.float64)
    n = len(y)
    y_old, M, R = np.ones(n), np.ones(n), np.ones(n)
    Fx = lin.Fx
    i = 1
    solve = None
    if linear_solver in ("auto", "device") and len(y) >= 2000:      # (below that the host's LU solve takes a tenth of a millisecond)
        solve = _device_inverse_solver(J̅, getattr(lin.hb, "device", None))
    if solve is None and linear_solver in ("auto", "lu", "device"):
        solve = _lu_solver(J̅)
    if inner == "krylov":
        count = [0]

        def matvec(v):
            count[0] += 1
            return lin.jvp(np.asarray(v, dtype=np.float64).ravel())

        Pinv = spla.LinearOperator((n, n), matvec=(solve if solve is not None else (lambda b: _gmres(J̅, b, np.zeros(n)))), dtype=np.float64)
        A = spla.LinearOperator((n, n), matvec=matvec, dtype=np.float64)
        # right-preconditioned: J(x) J̅⁻¹ z = F(x), y = J̅⁻¹ z; J(x) J̅⁻¹ ≈ I near the steady state. Stop when the linear residual is
        # below ε relative to ‖F(x)‖ and below 1e-3 ε absolutely (the outer loop's ‖y‖ ≤ ε test then sees a converged step)
        AP = spla.LinearOperator((n, n), matvec=lambda z: matvec(Pinv.matvec(z)), dtype=np.float64)
        z, info = spla.gmres(AP, Fx, rtol=min(1e-10, ε), atol=1e-3 * ε, restart=min(30, n), maxiter=max(1, max_inner // 30))
        y = Pinv.matvec(z)
        if info != 0:       # not converged within maxiter (or a breakdown): the step is still a descent direction of the preconditioned
            import warnings  # system, but the caller must know it is not the Newton step
            warnings.warn(f"y_Iteration(inner='krylov'): GMRES returned info={info} after {count[0]} JVPs; "
                          f"‖F − J y‖ = {np.linalg.norm(Fx - lin.jvp(y)):.3e}", RuntimeWarning, stacklevel=2)
        y_Iteration.last_jvp_count = count[0]
        y_Iteration.total_jvps = getattr(y_Iteration, "total_jvps", 0) + count[0]
        if verbose:
            print(f"y_Iteration (krylov): {count[0]} JVPs, info={info}, ‖F − J y‖={np.linalg.norm(Fx - lin.jvp(y))}")
        return y
    if inner != "fixed_point":
        raise ValueError(f"unknown inner iteration {inner!r} (fixed_point | krylov)")
    step = 0.5 if α is None else float(α)
    while ε < np.linalg.norm(y - y_old) and i < max_inner:
        Λxy = lin.jvp(y)
        if solve is not None:
            R = solve(Fx - Λxy)
            M = solve(Λxy) if verbose else M       # only feeds the printed Rayleigh quotient (:101, :108-110)
        else:
            R = _gmres(J̅, Fx - Λxy, R)
            M = _gmres(J̅, Λxy, M)
        with np.errstate(divide="ignore", invalid="ignore"):      # printed only (:101, :108-110); NaN at y = 0 like Julia
            ray = np.divide(y @ M, y @ y) if verbose else np.nan
        y_old = y
        y = y_old + step * R
        i += 1
        if verbose and i % 10 == 0:
            print(f"y_Iteration {i}: α={step}  ‖y−y_old‖={np.linalg.norm(y - y_old)}  ray={ray}")
    y_Iteration.last_jvp_count = i - 1
    y_Iteration.total_jvps = getattr(y_Iteration, "total_jvps", 0) + i - 1
    return y


@host_algebra
def NewtonRaphsonHANK(x_0, J̅, exog_paths, mod: SequenceModel, ss_initial, ss_ending, *, ε: float = 1e-9,
                      verbose: bool = False, linear_solver: str = "auto", inner: str = "fixed_point", α: float | None = None):
    """outer Newton loop (NewtonRaphson.jl:27-46): x ← x − y until ‖y‖ ≤ ε or 100 iterations. `inner` / `α`: see y_Iteration
    (defaults = the reference's damped fixed point)."""
    x = np.asarray(x_0, dtype=np.float64)
    y = x.copy()
    i = 1
    try:
        while ε < np.linalg.norm(y) and i < 100:
            y = y_Iteration(J̅, x, y, exog_paths, mod, ss_initial, ss_ending, verbose=verbose, linear_solver=linear_solver, inner=inner, α=α)
            x = x - y
            i += 1
            if verbose:
                print(f"Iteration: {i}, norm(y): {np.linalg.norm(y)}")
    finally:
        release_linear_solver()         # (the factorisations of J̅ live for one solve, not for the life of the process)
    NewtonRaphsonHANK.iterations = i - 1
    return x
