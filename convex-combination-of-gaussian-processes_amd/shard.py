"""One-process-per-GPU sharding of the independent evaluations.

Every (grid row x Halton node) likelihood and every (posterior draw x test point)
prediction depends only on the shared read-only (X, y) and its own parameter row, so
the path shards with NO data-path collective: rank r evaluates a contiguous block of
rows and a single all-gather (RCCL over xGMI when the backend is "nccl") collects the
scalar results.  The grid is sharded by grid ROW so that each row's mean over its N
Halton nodes stays rank-local (HX:574).

`evaluate` is injected so that the CPU test-suite can drive the same code with the
oracle over gloo; the product path passes a libccgp-backed evaluator (`hip_evaluator`).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of `total` items for `rank`; sizes differ by at most 1."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(total, world):
    return [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]


def gather_device(backend, tensor_device):
    """Where the all-gather buffers must live: RCCL ("nccl") only moves device memory, so host arrays (what the
    C ABI's host-pointer entry points return) are staged through this rank's GPU; gloo gathers in place."""
    import torch
    if backend == "nccl" and tensor_device.type != "cuda":
        return torch.device("cuda", torch.cuda.current_device())
    return tensor_device


def all_gather_rows(local, total, group=None):
    """Gather variable-length per-rank slices of a [rows, ...] float64 array into the full
    array on every rank (one all_gather of equal-size padded buffers).  numpy in -> numpy out; a tensor
    comes back on the device it came from."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    sizes = shard_sizes(total, world)
    is_t = isinstance(local, torch.Tensor)
    t = local if is_t else torch.from_numpy(np.ascontiguousarray(local))
    home = t.device
    dev = gather_device(dist.get_backend(group), home)
    if dev != home:
        t = t.to(dev)
    mx = max(sizes)
    pad_shape = (mx,) + tuple(t.shape[1:])
    buf = torch.zeros(pad_shape, dtype=t.dtype, device=dev)
    buf[: t.shape[0]] = t
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    full = torch.cat([o[: sizes[r]] for r, o in enumerate(outs)], dim=0)
    if dev != home:
        full = full.to(home)
    return full if is_t else full.numpy()


def sharded_loglik(evaluate, params, group=None):
    """params: [B, P] on every rank.  evaluate(params_slice) -> loglik[b] for the slice.
    Returns loglik[B] on every rank."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    B = params.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    local = np.asarray(evaluate(params[lo:hi]), dtype=np.float64).reshape(hi - lo)
    return all_gather_rows(local, B, group)


def sharded_grid(evaluate_rows, hyper, group=None):
    """hyper: [G, 4].  evaluate_rows(hyper_slice) -> per-row marginal likelihood.  Rows are
    sharded so each row's quadrature stays rank-local.  Returns (values[G], argmax)."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    G = hyper.shape[0]
    lo, hi = shard_bounds(G, rank, world)
    local = np.asarray(evaluate_rows(hyper[lo:hi]), dtype=np.float64).reshape(hi - lo)
    vals = all_gather_rows(local, G, group)
    finite = np.where(np.isnan(vals), -np.inf, vals)
    return vals, int(np.argmax(finite))


def sharded_predict(evaluate, draws, group=None):
    """draws: [S, P].  evaluate(draw_slice) -> (mean[s, m], var[s, m]).  Sharded over the
    posterior draws; returns the full (S, m) tables on every rank."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S = draws.shape[0]
    lo, hi = shard_bounds(S, rank, world)
    mean, var = evaluate(draws[lo:hi])
    both = np.stack([np.asarray(mean), np.asarray(var)], axis=1)  # [s, 2, m]
    full = all_gather_rows(both, S, group)
    return full[:, 0], full[:, 1]


def hip_evaluator(handle, X, y, K, sigma2, mean_mode=0, tau2=0.0):
    """The product evaluator: libccgp on this rank's GPU."""
    def evaluate(params_slice):
        if params_slice.shape[0] == 0:
            return np.empty(0)
        ll, _, _ = handle.loglik_batch(X, y, K, params_slice, sigma2, mean_mode, tau2)
        return ll
    return evaluate
