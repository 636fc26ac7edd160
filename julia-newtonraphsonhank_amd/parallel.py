"""Tangent-batch sharding across the GPUs of one node.

Tangent directions are independent given the primal (a JVP is linear in the tangent,
GeneralStructures.jl:546), so rank g takes columns [g·N/W, (g+1)·N/W) of the tangent matrix, every
rank runs the (cheap) primal sweep redundantly, and ONE all-gather of the n x N/W result blocks
assembles J·Y on every rank (RCCL over xGMI when the process group is "nccl"; "gloo" in CPU tests).
There is no other collective on the path.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def shard_bounds(N: int, world_size: int, rank: int):
    """contiguous near-equal column ranges; the first N % W ranks get one extra column."""
    base, extra = divmod(N, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_jvp(jvp_fn: Callable[[torch.Tensor], torch.Tensor], tangents: torch.Tensor, group=None) -> torch.Tensor:
    """jvp_fn maps an (n, k) tangent block to the (m, k) block J·Y_k on this rank's device.
    Returns the full (m, N) product on every rank."""
    W = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    N = tangents.shape[1]
    lo, hi = shard_bounds(N, W, rank)
    local = jvp_fn(tangents[:, lo:hi])
    if W == 1:
        return local
    m = local.shape[0]
    if N % W == 0:
        # equal shards: one all_gather_into_tensor of (k, m) row blocks (columns of J·Y are rows here)
        out = torch.empty((N, m), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.t().contiguous(), group=group)
        return out.t()
    # ragged shards: pad every block to the widest shard, gather once, trim
    kmax = -(-N // W)
    padded = torch.zeros((kmax, m), dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local.t()
    out = torch.empty((W * kmax, m), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    rows = [out[r * kmax: r * kmax + (shard_bounds(N, W, r)[1] - shard_bounds(N, W, r)[0])] for r in range(W)]
    return torch.cat(rows, dim=0).t()


def assemble_columns(jvp_block: Callable, n: int, chunk: int = 512, group=None, device=None):
    """J = [J·e_1 ... J·e_n] from unit tangents (the Jacobian assembly of SURVEY.md §8e): rank g owns the contiguous
    column range `shard_bounds(n, W, g)`, pushes it through `jvp_block` ((n, k) numpy -> (m, k) numpy: household block on
    the GPU, residual layer on the host) `chunk` columns per call, and ONE all-gather at the end puts the finished
    matrix on every rank — no collective per pass, nothing but the finished block crosses to the device for it.
    `device`: where the gathered blocks live ("cuda" for the nccl backend, CPU for gloo)."""
    import numpy as np
    W = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if W > 1 else 0
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (W > 1 and dist.get_backend(group) == "nccl") else torch.device("cpu")
    lo, hi = shard_bounds(n, W, rank)
    blocks = []
    for c0 in range(lo, hi, chunk):
        c1 = min(hi, c0 + chunk)
        E = np.zeros((n, c1 - c0))
        E[np.arange(c0, c1), np.arange(c1 - c0)] = 1.0
        blocks.append(np.ascontiguousarray(jvp_block(E)))
    local = np.concatenate(blocks, axis=1) if blocks else np.empty((n, 0))      # (the system is square: m = n rows)
    if W == 1:
        return local
    kmax = -(-n // W)
    padded = torch.zeros((kmax, local.shape[0]), dtype=torch.float64)
    padded[: hi - lo] = torch.from_numpy(local.T.copy())
    padded = padded.to(device)
    out = torch.empty((W * kmax, local.shape[0]), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, padded, group=group)
    out = out.cpu()
    rows = [out[r * kmax: r * kmax + (shard_bounds(n, W, r)[1] - shard_bounds(n, W, r)[0])] for r in range(W)]
    return torch.cat(rows, dim=0).t().contiguous().numpy()


class DeviceGroup:
    """ONE host process driving one context per GPU of the node — the multi-GPU form behind the C ABI (hank_create_on;
    INTEGRATION.md section 6): the model, boundary and primal are replicated, GPU g takes the tangent columns
    `shard_bounds(N, len(devices), g)`, and the results land in ONE host array — no collective, no second process.
    Every context is driven from its own host thread (the library releases the GIL inside a call and a context makes
    its device current for the call), so the sweeps of all GPUs overlap. `block` is the context of devices[0]."""

    def __init__(self, block, devices):
        self.devices = list(devices)
        if not self.devices:
            raise ValueError("DeviceGroup needs at least one device")
        if block.device is not None and block.device != self.devices[0]:
            raise ValueError(f"DeviceGroup: `block` lives on device {block.device}, devices[0] is {self.devices[0]}")
        self.blocks = [block] + [block.clone(device=d) for d in self.devices[1:]]
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=len(self.blocks))

    def _each(self, fn, args_per_block=None):
        futs = [self._pool.submit(fn, b, *(args_per_block[k] if args_per_block else ())) for k, b in enumerate(self.blocks)]
        return [f.result() for f in futs]

    def set_boundary(self, value, D):
        self._each(lambda b: b.set_boundary(value, D))

    def primal(self, x):
        """the Float64 sweep on every GPU (redundant and cheap); returns the aggregates of the first."""
        return self._each(lambda b: b.primal(x))[0]

    def jvp(self, y):
        """y: (n_hh, P, N) -> (P, N): columns sharded over the GPUs, assembled on the host."""
        import numpy as np
        y = np.asarray(y, dtype=np.float64)
        N, W = y.shape[2], len(self.blocks)
        bounds = [shard_bounds(N, W, g) for g in range(W)]
        parts = self._each(lambda b, lo, hi: b.jvp(y[:, :, lo:hi]) if hi > lo else np.empty((y.shape[1], 0)), bounds)
        return np.concatenate(parts, axis=1)

    def jvp_dev(self, d_y_blocks, N_k):
        """the same partition with everything on the devices: GPU g's tangent columns are already in its memory (`d_y_blocks[g]`: a
        torch tensor of (n_hh, P, N_k[g]) column-major values on that device), every context runs its sweeps asynchronously, and
        the (P, N_k[g]) results are assembled in devices[0]'s memory by hank_gather_columns — peer copies over xGMI, no host buffer,
        no second process. Returns the (P, sum N_k) column-major matrix as a flat torch tensor on devices[0] (synchronised)."""
        import torch
        from . import hip
        P = self.blocks[0].P
        outs = []
        for b, d, dy, nk in zip(self.blocks, self.devices, d_y_blocks, N_k):
            o = torch.empty(P * int(nk), dtype=torch.float64, device=torch.device("cuda", d))
            if nk:
                b.jvp_dev(dy.data_ptr(), int(nk), o.data_ptr())
            outs.append(o)
        full = torch.empty(P * int(sum(N_k)), dtype=torch.float64, device=torch.device("cuda", self.devices[0]))
        hip.gather_columns(self.blocks, [o.data_ptr() for o in outs], N_k, full.data_ptr())
        self.blocks[0].sync()
        self._keep = outs                    # (the blocks must outlive the copies)
        return full

    def close(self):
        for b in self.blocks[1:]:
            b.close()
        self._pool.shutdown(wait=True)
