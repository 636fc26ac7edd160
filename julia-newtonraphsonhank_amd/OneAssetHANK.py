"""One-asset HANK household block — a SECOND value-function family behind the reference's plugin contract
`value_fn(value_next, xVals, model) -> (Value, <het vars>...)` (GeneralStructures.jl:210-214).

NOT in the reference (BASELINE.json configs[4] "EconPizza-style spec"): there is no Julia
value function, YAML or test to pin it to, so parity is UNPINNED by construction; the oracle carries the same
family (`orc_value_function_tr`) and the GPU path is checked against it and against finite differences.

Household: CRRA consumption-savings with idiosyncratic productivity z_e, one asset a >= borrow_cons and the budget

    c + a' = (1 + r_t) a + om_t z_e + Tr_t

where om_t is after-tax labour income per efficiency unit (hours are demand-determined and rationed equally) and
Tr_t a lump-sum transfer (dividends + net government transfers). This is the Krusell-Smith EGM step
(KrusellSmith.jl:59-80) with a lump sum added to cash on hand, so the native kernel family
HANK_VF_ONE_ASSET_HANK reuses egm_X / egm_Y and their tangent forms with a third household input.

TWO heterogeneous variables: the savings policy `A` and consumption `C` (the budget residual, the c_grid of
KrusellSmith.jl:79 returned as a second policy). A model lists the ones its equations use under `heterogeneous:`;
ForwardIteration aggregates each with the same D_t (ForwardIteration.jl:303-307) — on the device, inside the fused
sweeps (hank_get_het_outputs).
"""
from __future__ import annotations

import numpy as np

from .ModelParser import register_function

HANK_VF_ONE_ASSET_HANK = 1


def monetary_shock(T: int, size: float = 0.0025, ρ: float = 0.6) -> np.ndarray:
    """deterministic AR(1) innovation path ei_t = size * ρ^(t-1) to the Taylor rule (25 bp per quarter by default)."""
    return size * ρ ** np.arange(T)


class _HANKValueFunction:
    name = "HANKValueFunction"
    value_fn_id = HANK_VF_ONE_ASSET_HANK
    household_inputs = ("r", "om", "Tr")   # rows of xVals the household block reads
    outputs = ("A", "C")                    # device output index = position here (hank_get_het_outputs)
    endogenous_dim, exogenous_dim = "wealth", "productivity"

    def derived_policy(self, key: str, policy, xVals: dict, model):
        """a heterogeneous variable other than the policy variable, from the savings policy (n_a, n_e) — Float64 or `Dual` —
        and the period's inputs: consumption, the budget residual."""
        if key != "C":
            raise KeyError(key)
        grid = model.heterogeneity["wealth"].grid
        z = model.heterogeneity["productivity"].grid
        return (1.0 + xVals["r"]) * grid[:, None] + (xVals["om"] * z[None, :] + xVals["Tr"]) - policy

    def host_steady_state_step(self, value_next: np.ndarray, xVals: dict, model) -> dict:
        """one Float64 EGM step for the host steady-state VFI (the KS step with the transfer in cash on hand)."""
        grid = model.heterogeneity["wealth"].grid
        z = model.heterogeneity["productivity"].grid
        Π = model.heterogeneity["productivity"].transition
        β, γ, bc = model.params.β, model.params.γ, model.params.borrow_cons
        r, om, tr = xVals["r"], xVals["om"], xVals["Tr"]
        inc = om * z[None, :] + tr
        cmat = (β * (value_next @ Π.T)) ** (-1.0 / γ)
        s = (cmat - inc + grid[:, None]) / (1.0 + r)
        if not np.all(np.diff(s, axis=0) > 0):
            raise ValueError("knot-vectors must be unique and sorted in increasing order")
        g = np.empty_like(s)
        for e in range(z.size):
            g[:, e] = np.interp(grid, s[:, e], grid)   # flat outside [s_1, s_n]
        g = np.maximum(g, bc)
        c = (1.0 + r) * grid[:, None] + inc - g
        return {"Value": (1.0 + r) * c ** (-γ), "A": g, "C": c}

    def __call__(self, *a, **k):
        raise RuntimeError("HANKValueFunction is a native kernel family (libhank_hip); on the transition path it is "
                           "invoked by BackwardIteration on the GPU, not called from Python.")


HANKValueFunction = _HANKValueFunction()
register_function("HANKValueFunction", HANKValueFunction)
register_function("monetary_shock", monetary_shock)


def household_asset_demand(model, r: float, om: float, tr: float, tol: float = 1e-10, max_iter: int = 20_000):
    """stationary asset demand A(r, om, Tr) of the household block on the host (value-function iteration + invariant
    distribution); used by `calibrate_bond_supply`."""
    from .ForwardIteration import make_endogenous_transition
    from .GeneralStructures import invariant_dist
    import scipy.sparse as sp
    wd = model.heterogeneity["wealth"]
    pdm = model.heterogeneity["productivity"]
    xv = {"r": r, "om": om, "Tr": tr}
    V = np.ones((wd.n, pdm.n))
    for _ in range(max_iter):
        res = HANKValueFunction.host_steady_state_step(V, xv, model)
        if np.max(np.abs(res["Value"] - V)) < tol:
            V = res["Value"]
            break
        V = res["Value"]
    pol = res["A"]
    Λe = make_endogenous_transition(pol, wd, pdm.n)
    Λx = sp.kron(sp.csc_matrix(pdm.transition.T), sp.identity(wd.n, format="csc"), format="csc")
    D = invariant_dist((Λx @ Λe).tocsc().T)
    return float(np.dot(pol.reshape(-1, order="F"), D)), V, pol, D


def calibrate_bond_supply(model, tol: float = 1e-10):
    """the real bond supply B that clears the asset market at the model's target real rate `rstar`:
    B = A(rstar, om_ss, Tr_ss(B)) (the transfer depends on B through the government's interest bill). Secant on B."""
    p = model.params
    om = (1.0 - p.τ) / p.μ                       # w = 1/μ and Y = 1 at zero inflation
    def excess(B):
        tr = 1.0 - 1.0 / p.μ + p.τ / p.μ - p.rstar * B
        return household_asset_demand(model, p.rstar, om, tr)[0] - B
    b0, b1 = 1.0, 2.0
    f0, f1 = excess(b0), excess(b1)
    for _ in range(60):
        if abs(f1) < tol:
            break
        b0, b1, f0 = b1, b1 - f1 * (b1 - b0) / (f1 - f0), f1
        f1 = excess(b1)
    return b1
