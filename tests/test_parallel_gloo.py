"""N>1 path on CPU: tangent columns sharded over 2 gloo ranks, one all-gather assembles J·Y."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, N, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import ks_paths, ks_setup
        from hank_amd.parallel import shard_bounds, sharded_jvp
        m, ss, orc = ks_setup(20, 2, 12)
        P = m.compspec.T - 1
        x, Z = ks_paths(m, ss, "x1", 0.05)
        Y = torch.from_numpy(np.random.default_rng(0).standard_normal((4 * P, N)))

        def jvp_fn(block):      # oracle-backed stand-in for the GPU block product (tests only)
            yb = block.numpy().reshape(4, P, -1, order="F")
            if yb.shape[2] == 0:
                return torch.empty((4 * P, 0), dtype=torch.float64)
            return torch.from_numpy(orc.ks_jvp(x, yb, Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)[1])

        full = sharded_jvp(jvp_fn, Y)
        ref = jvp_fn(Y)
        ok = full.shape == ref.shape and torch.equal(full, ref)
        lo, hi = shard_bounds(N, world, rank)
        ret[rank] = (bool(ok), lo, hi)
    finally:
        dist.destroy_process_group()


def _worker_cols(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hank_amd.parallel import assemble_columns
        n = 13
        A = np.random.default_rng(7).standard_normal((n, n))
        calls = []

        def jvp_block(E):       # a linear stand-in for LinearizedFunction.jvp
            calls.append(E.shape[1])
            return A @ E

        J = assemble_columns(jvp_block, n, chunk=3)
        ret[rank] = (bool(np.array_equal(J, A)), sum(calls))
    finally:
        dist.destroy_process_group()


def test_jacobian_columns_assembled_over_two_ranks():
    """getSteadyStateJacobian's column assembly under torch.distributed: each rank pushes its share of every pass,
    one all-gather per pass, every rank ends with the whole matrix (ragged last pass included)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_cols, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] + ret[1][1] == 13 and abs(ret[0][1] - ret[1][1]) <= 2      # the columns were shared out


@pytest.mark.parametrize("N", [8, 5])
def test_sharded_jvp_two_ranks(N):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() + N) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == 0 and ret[0][2] == ret[1][1] and ret[1][2] == N   # contiguous cover


def test_shard_bounds_cover(hank):
    from hank_amd.parallel import shard_bounds
    for N in (1, 7, 32, 256):
        for W in (1, 2, 3, 8):
            edges = [shard_bounds(N, W, r) for r in range(W)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
