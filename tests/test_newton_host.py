"""Host-side pieces of the Newton driver that need no GPU: the LU cache of the J̅ solves, the lazy
BackwardIteration/ForwardIteration pairing, and the reference-side Julia shim's source forms."""
import re
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent


def test_lu_cache_is_keyed_by_the_matrix_itself(hank):
    """two different J̅ of equal size, created and dropped back to back (CPython recycles their id()): each must be
    solved with ITS factors (round-1 bug: the cache was keyed by id(J) and returned the first matrix's LU)."""
    from hank_amd.NewtonRaphson import _lu_solver
    rng = np.random.default_rng(0)
    n = 40
    b = rng.standard_normal(n)

    def solve_fresh(seed):
        A = np.random.default_rng(seed).standard_normal((n, n)) + n * np.eye(n)
        J = sp.csc_matrix(A)
        return _lu_solver(J)(b), np.linalg.solve(A, b)

    for seed in (1, 2, 3, 4):
        got, want = solve_fresh(seed)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    # same object twice: factored once
    A = rng.standard_normal((n, n)) + n * np.eye(n)
    from hank_amd import NewtonRaphson as nr
    _lu_solver(A); first = nr._LU_CACHE[0][1]
    _lu_solver(A)
    assert nr._LU_CACHE[0][1] is first


def test_y_iteration_at_zero_direction_does_not_divide_by_zero(hank):
    """ray = (y·M)/(y·y) is NaN at y = 0 in the reference (printed only): no ZeroDivisionError here."""
    with np.errstate(divide="ignore", invalid="ignore"):
        assert np.isnan(np.divide(np.zeros(3) @ np.ones(3), np.zeros(3) @ np.zeros(3)))
    src = (ROOT / "julia-newtonraphsonhank_amd" / "NewtonRaphson.py").read_text()
    assert "float(y @ M) / float(y @ y)" not in src


def test_julia_shim_uses_constructors_the_reference_has():
    """julia/HankHIP.jl cannot run here (no Julia): pin its source forms instead. Duals are built with
    Dual{Tag}(value, Partials(tuple)) (ForwardDiff.jl/src/dual.jl:59-66) — there is no TF(value, p1, ..., pN)
    method for TF = Dual{T,V,N} (:14-21 takes a Partials) — and the four-argument BackwardIteration
    (NewtonRaphson.jl:78) does no device work."""
    src = (ROOT / "julia" / "HankHIP.jl").read_text()
    code = "\n".join(l.split("#")[0] for l in src.splitlines())
    assert not re.search(r"\bTF\(", code), "TF(value, partials...) has no method in the reference's ForwardDiff"
    assert "Dual{tagtype(TF)}(v, Partials(p))" in code
    assert "using ForwardDiff: Dual, Partials" in code
    body = code[code.index("function BackwardIteration("):code.index("function Base.getproperty")]
    assert "ccall" not in body
    assert body.count("_run_block!") == 1 and "ss_initial === nothing || _run_block!" in body
    fwd = code[code.index("function ForwardIteration("):]
    assert "_run_block!(seqs, D0)" in fwd
    # one context per (model, HIP device): hank_create_on when a device is named, and the column shard helper drives them from tasks
    assert "hank_context(model::SequenceModel; device" in code and "(:hank_create_on, LIBHANK)" in code
    shard = code[code.index("function sharded_jvp_columns("):]
    assert "Threads.@spawn" in shard and "hank_context(model; device = dev)" in shard and "(:hank_primal_jvp, LIBHANK)" in shard
    # the household block of getSteadyStateJacobian: one hank_fake_news + the reference's recursion (SteadyStateJacobian.jl:363-371)
    toe = code[code.index("function household_jacobian_toeplitz("):]
    assert "(:hank_fake_news, LIBHANK)" in toe and "J[t-1, 1:P-1, k] .+ F[t, 2:P, k]" in toe and "Dv[:, k] .+ F[1, :, k]" in toe
    # every ccall target is a symbol the C ABI exports
    from hank_amd.hip import ABI_SYMBOLS
    for sym in set(re.findall(r"ccall\(\(:(\w+), LIBHANK\)", src)):
        assert sym in ABI_SYMBOLS, sym


class _StubBlock:
    """stands in for the device context so that the lazy pairing logic can run without a GPU."""

    def __init__(self, P, G, n_a, n_e):
        self.P, self.G, self.n_a, self.n_e, self.n_hh = P, G, n_a, n_e, 2
        self.calls = {"primal": 0, "jvp": 0, "primal_jvp": 0}
        self.boundaries = []

    def set_boundary(self, v, D):
        self.boundaries.append(np.array(D, copy=True))

    def primal(self, xhh):
        self.calls["primal"] += 1
        return np.full(self.P, float(np.sum(self.boundaries[-1][:3])))

    def primal_jvp(self, xhh, dxhh):
        self.calls["primal_jvp"] += 1
        return np.zeros(self.P), np.zeros((self.P, dxhh.shape[2]))

    def policy_seq(self):
        return np.zeros((self.n_a, self.n_e, self.P), order="F")


def test_four_argument_backward_iteration_is_deferred_host_logic(hank):
    from types import SimpleNamespace
    m = hank.build_model_from_yaml(str(ROOT / "examples" / "krusell_smith.yaml"),
                                   overrides={"T": 6, "dimensions": {"wealth": {"n": 8}, "productivity": {"n": 2}}})
    P, G = 5, 16
    stub = _StubBlock(P, G, 8, 2)
    m._hip_block = stub
    ss = SimpleNamespace(value=np.ones((8, 2)), D=np.arange(G, dtype=float) / np.arange(G).sum(), vars={})
    x = np.tile(np.array([1.0, 3.0, 0.03, 1.2]), P)
    seqs = hank.BackwardIteration(x, {"Z": np.ones(P)}, m, ss)
    assert stub.calls["primal"] == 0 and not stub.boundaries            # deferred
    out = hank.ForwardIteration(seqs, m, ss)["KD"]
    assert stub.calls["primal"] == 1 and np.array_equal(stub.boundaries[-1], ss.D)
    assert out[0] == float(np.sum(ss.D[:3]))
    hank.ForwardIteration(seqs, m, ss)
    assert stub.calls["primal"] == 1                                     # nothing re-run
    # five-argument form runs immediately
    hank.BackwardIteration(x, {"Z": np.ones(P)}, m, ss, ss_initial=ss)
    assert stub.calls["primal"] == 2
    # reading a policy first: placeholder D_0, then ForwardIteration redoes with the right one
    s2 = hank.BackwardIteration(x, {"Z": np.ones(P)}, m, ss)
    _ = s2["KD"]
    assert stub.calls["primal"] == 3 and np.allclose(stub.boundaries[-1], 1.0 / G)
    hank.ForwardIteration(s2, m, ss)
    assert stub.calls["primal"] == 4 and np.array_equal(stub.boundaries[-1], ss.D)
    # Dual pass: one primal_jvp
    xd = hank.Dual.seed(x, np.ones((x.size, 2)))
    hank.ForwardIteration(hank.BackwardIteration(xd, {"Z": np.ones(P)}, m, ss), m, ss)
    assert stub.calls["primal_jvp"] == 1
