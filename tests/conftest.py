import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hank():
    try:                                # torch's HIP runtime must be up before libhank_hip loads its own (a GPU test that moves a
        import torch                    # tensor to the device after the library has loaded finds "No HIP GPUs" otherwise)
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:                   # noqa: BLE001
        pass
    import hank_amd
    return hank_amd


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


_SS_CACHE = {}


def ks_setup(n_a, n_e, T):
    """model + steady state + oracle for a Krusell-Smith economy of the given size (cached)."""
    key = (n_a, n_e, T)
    if key not in _SS_CACHE:
        import hank_amd as h
        from oracle.oracle import Oracle
        ov = {"T": T, "dimensions": {"wealth": {"n": n_a}, "productivity": {"n": n_e}}}
        m = h.build_model_from_yaml(str(ROOT / "examples" / "krusell_smith.yaml"), overrides=ov)
        fx = ROOT / "examples" / "fixtures" / f"ks_ss_{n_a}x{n_e}.npz"
        ss = None
        if fx.exists() and not os.environ.get("HANK_NO_SS_FIXTURE"):
            g = np.load(fx)
            if np.array_equal(g["a_grid"], m.heterogeneity["wealth"].grid) and \
                    np.array_equal(g["Pi"], m.heterogeneity["productivity"].transition):
                ss = h.SteadyState({k: float(g[f"var_{k}"]) for k in m.variables}, {"KD": g["policy"]}, None, g["D"], g["value"])
        if ss is None:
            # host VFI: the committed goldens were generated without a GPU, and the steady state must be the same
            # bit for bit here and on the GPU box (the device VFI has its own tests: test_gpu_steady_state.py)
            ss, _ = h.get_SteadyStates(m, vfi="host")
        wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
        orc = Oracle(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons)
        _SS_CACHE[key] = (m, ss, orc)
    return _SS_CACHE[key]


def ks_paths(m, ss, kind="x1", shock=0.01):
    """primal points of BASELINE.md §3: x0 = SS repeated; x1 = firm conditions at fixed SS capital
    under Z_t = 1 + shock*0.8^t (non-stationary, exercises moving brackets)."""
    P = m.compspec.T - 1
    α, δ = m.params.α, m.params.δ
    t = np.arange(1, P + 1)
    Z = 1.0 + shock * 0.8 ** t
    K = ss.vars["KS"]
    if kind == "x0":
        x = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")])[:, None], (1, P))
    else:
        x = np.stack([Z * K ** α, np.full(P, K), α * Z * K ** (α - 1) - δ, (1 - α) * Z * K ** α])
    return x, Z


@pytest.fixture(scope="session")
def ks_small():
    return ks_setup(50, 2, 100)
