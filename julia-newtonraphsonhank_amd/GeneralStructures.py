"""Model-agnostic host structures of the sequence-space solver — the Python mirror of the
reference's GeneralStructures.jl (same names, argument meaning and error behaviour).

Host-only: these are the boundary types the MI355X household block is called through
(SURVEY.md §8b); nothing here is on the GPU hot path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np

from .dual import Dual


# ── heterogeneity / steady-state specs (GeneralStructures.jl:43-49, :73-76) ──────────────────
@dataclass(frozen=True)
class HeterogeneityDimension:
    dim_type: str                      # "endogenous" | "exogenous"
    n: int
    grid: np.ndarray
    transition: Optional[np.ndarray]   # n x n row-stochastic for exogenous dims, else None
    policy_var: Optional[str]          # e.g. "KD" for the wealth dimension


def n_total(heterogeneity: Dict[str, HeterogeneityDimension]) -> int:
    """product of all dimension sizes (GeneralStructures.jl:59)."""
    out = 1
    for d in heterogeneity.values():
        out *= d.n
    return out


@dataclass(frozen=True)
class SteadyStateSpec:
    fixed: Dict[str, float]
    guesses: Dict[str, float]


# ── model structures (GeneralStructures.jl:106-111, :166-174, :216-226) ──────────────────────
@dataclass(frozen=True)
class Variable:
    name: str
    var_type: str                      # "endogenous" | "exogenous" | "heterogeneous"
    description: str = ""
    seq_fn: Optional[Callable] = None


@dataclass(frozen=True)
class ComputationalSpec:
    T: int
    ε: float
    dx: float
    n_v: int
    n_endog: int
    max_lag: int
    max_lead: int


@dataclass
class SequenceModel:
    variables: Dict[str, Variable]     # insertion order = row order of xMat
    equations: Tuple[str, ...]
    compspec: ComputationalSpec
    params: SimpleNamespace
    residuals_fn: Callable
    ss_initial: SteadyStateSpec
    ss_ending: SteadyStateSpec
    heterogeneity: Dict[str, HeterogeneityDimension]
    value_fn: Any                      # a plugin object (KrusellSmith.ValueFunction): name -> native kernel family
    _hip_block: Any = field(default=None, repr=False)


def var_names(model) -> Tuple[str, ...]:
    return tuple(model.variables.keys())


def vars_of_type(model, t: str) -> Tuple[str, ...]:
    return tuple(k for k, v in model.variables.items() if v.var_type == t)


# ── out-of-the-box grid functions (GeneralStructures.jl:242-261) ─────────────────────────────
def double_exponential(*, n, grid_min, grid_max) -> np.ndarray:
    return np.array(make_DoubleExponentialGrid(float(grid_min), float(grid_max), int(n)), dtype=np.float64)


def rouwenhorst_discretization(*, n, ρ, σ):
    Π, _, z = get_RouwenhorstDiscretization(int(n), float(ρ), float(σ))
    return z, Π


# ── sequence-space assembly helpers ──────────────────────────────────────────────────────────
def generate_exog_paths(model: SequenceModel, T: int) -> Dict[str, np.ndarray]:
    """call every exogenous variable's seq_fn (GeneralStructures.jl:279-289)."""
    paths = {}
    for key in vars_of_type(model, "exogenous"):
        var = model.variables[key]
        if var.seq_fn is None:
            raise ValueError(f"Exogenous variable '{key}' has no seq_fn. Specify a seq_function in the YAML.")
        paths[key] = np.asarray(var.seq_fn(T), dtype=np.float64)
    return paths


def assemble_full_xMat(xVec_endog, agg_seqs, exog_paths, model: SequenceModel, ss_start, ss_end):
    """padded n_v x T_pad matrix for the compiled residuals (GeneralStructures.jl:329-377).

    Columns 1:max_lag = ss_start.vars, max_lag+1 : max_lag+T-1 = transition path, the rest =
    ss_end.vars. Works on float arrays and on `Dual` (boundary columns get zero partials)."""
    cs = model.compspec
    T, n_v, n_endog, max_lag, max_lead = cs.T, cs.n_v, cs.n_endog, cs.max_lag, cs.max_lead
    P = T - 1
    T_pad = P + max_lag + max_lead
    all_keys = var_names(model)
    endog_keys = vars_of_type(model, "endogenous")
    het_keys = vars_of_type(model, "heterogeneous")
    exog_keys = vars_of_type(model, "exogenous")

    is_dual = isinstance(xVec_endog, Dual) or any(isinstance(v, Dual) for v in agg_seqs.values())
    N = None
    if is_dual:
        N = xVec_endog.N if isinstance(xVec_endog, Dual) else next(v.N for v in agg_seqs.values() if isinstance(v, Dual))
    xv = np.zeros((n_v, T_pad))
    xp = np.zeros((n_v, T_pad, N)) if is_dual else None

    for col in range(max_lag):
        for row, k in enumerate(all_keys):
            xv[row, col] = ss_start.vars[k]
    for col in range(max_lag + P, T_pad):
        for row, k in enumerate(all_keys):
            xv[row, col] = ss_end.vars[k]

    def put(row, seq):
        if isinstance(seq, Dual):
            xv[row, max_lag:max_lag + P] = seq.v
            xp[row, max_lag:max_lag + P, :] = seq.p
        else:
            xv[row, max_lag:max_lag + P] = np.asarray(seq, dtype=np.float64)

    if isinstance(xVec_endog, Dual):
        xe = Dual(xVec_endog.v.reshape(n_endog, P, order="F"), xVec_endog.p.reshape(n_endog, P, N, order="F"))
    else:
        xe = np.asarray(xVec_endog, dtype=np.float64).reshape(n_endog, P, order="F")
    for j, k in enumerate(endog_keys):
        put(all_keys.index(k), xe[j])
    for k in het_keys:
        put(all_keys.index(k), agg_seqs[k])
    for k in exog_keys:
        put(all_keys.index(k), exog_paths[k])
    return Dual(xv, xp) if is_dual else xv


# ── time-shift operators (GeneralStructures.jl:441-455) ──────────────────────────────────────
def shift_lag(x, i: int):
    if isinstance(x, Dual):
        return Dual(shift_lag(x.v, i), np.concatenate([np.repeat(x.p[:1], i, axis=0), x.p[:len(x.v) - i]], axis=0))
    x = np.asarray(x)
    return np.concatenate([np.full(i, x[0]), x[:len(x) - i]])


def shift_lead(x, i: int):
    if isinstance(x, Dual):
        return Dual(shift_lead(x.v, i), np.concatenate([x.p[i:], np.repeat(x.p[-1:], i, axis=0)], axis=0))
    x = np.asarray(x)
    return np.concatenate([x[i:], np.full(i, x[-1])])


# ── grid construction primitives (GeneralStructures.jl:474-525) ──────────────────────────────
def make_DoubleExponentialGrid(amin: float, amax: float, n_a: int) -> np.ndarray:
    """a = amin + exp(exp(u) - 1) - 1, u uniform on [0, log(1 + log(1 + amax - amin))]."""
    U = np.log(1 + np.log(1 + amax - amin))
    # Julia's range(0, U, n_a): start + (i-1)*step with the endpoint hit exactly
    ugrid = np.linspace(0.0, U, n_a)
    return amin + np.exp(np.exp(ugrid) - 1) - 1


def invariant_dist(Π, D0=None, direct_max: int = 4000) -> np.ndarray:
    """stationary distribution of a row-stochastic Π.

    Chains up to `direct_max` states use the reference's linear-system trick
    (ForwardIteration.jl:436-442): solve (I - Πᵀ[2:,2:]) y = Πᵀ[2:,1], D = [1; y]/sum. Larger chains
    (the 2000x11 benchmark grid: a sparse LU there costs ~15 s per call) and chains where state 1 is
    transient (the trick's normalisation D[1] = 1 is then singular — Julia's `\` would throw) use
    the power method, warm-started from `D0` when given. Same fixed point either way."""
    import warnings
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    n = Π.shape[0]
    if n <= direct_max:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                if sp.issparse(Π):
                    ΠT = Π.T.tocsc()
                    M = sp.identity(n - 1, format="csc") - ΠT[1:, 1:]
                    b = np.asarray(ΠT[1:, 0].todense()).ravel()
                    y = spla.spsolve(M.tocsc(), b)
                else:
                    ΠT = np.asarray(Π, dtype=np.float64).T
                    y = np.linalg.solve(np.eye(n - 1) - ΠT[1:, 1:], ΠT[1:, 0])
            except (np.linalg.LinAlgError, RuntimeError):
                y = np.array([np.nan])
        D = np.concatenate([[1.0], np.atleast_1d(y)])
        if D.size == n and np.all(np.isfinite(D)) and D.min() > -1e-12 and D.sum() > 0:
            return D / D.sum()
    A = (Π.T.tocsr() if sp.issparse(Π) else np.asarray(Π, dtype=np.float64).T)
    D = np.full(n, 1.0 / n) if D0 is None else np.asarray(D0, dtype=np.float64) / np.sum(D0)
    for k in range(500_000):
        Dn = A @ D
        if k % 25 == 0:
            Dn /= Dn.sum()
            if np.max(np.abs(Dn - D)) < 1e-15:
                D = Dn
                break
        D = Dn
    return D / D.sum()


def get_RouwenhorstDiscretization(n: int, ρ: float, σ: float):
    """Rouwenhorst (1995) discretisation (GeneralStructures.jl:500-525): returns (Π, D, z)."""
    p = (1 + ρ) / 2
    Π = np.array([[p, 1 - p], [1 - p, p]])
    for i in range(3, n + 1):
        Π_old = Π
        Π = np.zeros((i, i))
        Π[:i - 1, :i - 1] += p * Π_old
        Π[:i - 1, 1:] += (1 - p) * Π_old
        Π[1:, :i - 1] += (1 - p) * Π_old
        Π[1:, 1:] += p * Π_old
        Π[1:i - 1, :] /= 2
    D = invariant_dist(Π)
    α = 2 * (σ / np.sqrt(n - 1))
    z = np.exp(α * np.arange(n))
    z = z / np.sum(z * D)
    return Π, D, z


# ── linear-algebra utilities (GeneralStructures.jl:542-561) ──────────────────────────────────
def JVP(func: Callable, primal, tangent):
    """J(primal)·tangent by forward mode: func(primal + t·tangent), t = Dual(0, 1)
    (GeneralStructures.jl:542-550; ForwardDiff.derivative, derivative.jl:12-15).

    `tangent` may be a vector (one direction, as in the reference) or an (n, N) matrix — a batch of
    N directions carried as N partials (ForwardDiff's chunk mode); the household block inside
    `func` then runs ONE batched hank_jvp on the GPU. Returns shape (m,) or (m, N)."""
    primal = np.asarray(primal, dtype=np.float64)
    tangent = np.asarray(tangent, dtype=np.float64)
    single = tangent.ndim == 1
    res = func(Dual.seed(primal, tangent[:, None] if single else tangent))
    if not isinstance(res, Dual):
        raise TypeError("JVP: func did not propagate dual numbers")
    return res.p[:, 0].copy() if single else res.p.copy()


def RayleighQuotient(M, z):
    z = np.asarray(z, dtype=np.float64)
    return float(z @ (M @ z)) / float(z @ z)
