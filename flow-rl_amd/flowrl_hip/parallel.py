"""Batch sharding of a sampler call over the GPUs of one node (one process per GPU, RCCL over xGMI).

The path shards embarrassingly: chains are independent and the 0.6-3.3 MB of
weights are replicated, so rank r owns the contiguous block of STATES [lo, hi) — a state's particles
stay on one GPU, its condition row is read once and best-of-N selection stays local.  The only
collective is the final all-gather of x0 (and of the histories when requested); the reference itself
has no distributed path (it runs one independent job per GPU, scripts/mujoco/dacer.sh:28-52).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_states: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first (n_states % world) ranks own one extra state."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} / world {world} invalid")
    q, r = divmod(n_states, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_ragged(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """All-gather along dim 0 when ranks own different row counts (pads to the largest shard; one
    collective, sized for the point-to-point xGMI mesh: every rank sends its shard once)."""
    world = dist.get_world_size(group)
    mx = max(counts)
    if all(c == mx for c in counts):
        out = torch.empty((world * mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world, mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(world * mx, *local.shape[1:]), pad, group=group)
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)


def sample_sharded(sample_fn: Callable, xT: torch.Tensor, condition: Optional[torch.Tensor],
                   noise: Optional[torch.Tensor], particles: int = 1, group=None, gather: bool = True, keyed: bool = False):
    """Run ``sample_fn(xT_shard, cond_shard, noise_shard) -> x0_shard`` on this rank's block of states and
    all-gather the result.  Global layouts: xT (n_states*particles, A), condition (n_states, C) un-repeated,
    noise (T, n_states*particles, A).  Every rank passes the same global tensors (or at least its own block
    at the right offsets) and receives the full x0 when ``gather`` is set, else its own shard.  ``keyed``: the third
    argument handed to ``sample_fn`` is ``rng_shard`` = (first chain, total chains) for ``model.sample(key, ..., rng_shard=)``
    instead of a noise block."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_states = xT.shape[0] // particles
    lo, hi = shard_bounds(n_states, rank, world)
    clo, chi = lo * particles, hi * particles
    if keyed:
        # rng-in / rng-out sharding: every rank passes the SAME key; its draws are the elements [clo, chi) of the unsharded
        # normal(key_j, (B, A)) (FrlFwdBufs.rng_chain_offset / rng_total_chains), so the gathered x0 is the single-GPU call's
        x0 = sample_fn(xT[clo:chi], None if condition is None else condition[lo:hi], (clo, xT.shape[0]))
    else:
        x0 = sample_fn(xT[clo:chi], None if condition is None else condition[lo:hi],
                       None if noise is None else noise[:, clo:chi])
    if world == 1 or not gather:
        return x0
    counts = [(shard_bounds(n_states, r, world)[1] - shard_bounds(n_states, r, world)[0]) * particles for r in range(world)]
    return all_gather_ragged(x0, counts, group)


def _leaves(tree, prefix=()):
    for k in sorted(tree):
        v = tree[k]
        if isinstance(v, dict):
            yield from _leaves(v, prefix + (k,))
        else:
            yield prefix + (k,), v


def all_reduce_grads(grads: dict, group=None) -> dict:
    """Sum a gradient tree over the ranks with ONE all-reduce of the flat f32 buffer (VJP
    collective: 152,579 + 8,320 parameters = 0.64 MB for DQL -> latency-bound; one bucket, never one collective per
    leaf).  The loss of the reference's actor update is a mean over the global batch (dql.py:112-118): the caller
    folds 1/global_batch into the cotangent, so a plain sum is what is needed here.  In place; returns ``grads``."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return grads
    flat = getattr(grads, "flat", None)
    if flat is not None:
        # the library's VJP wrote its leaves into one flat buffer (_sampler.GradTree): the collective runs on it in place
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return grads
    items = list(_leaves(grads))
    flat = torch.cat([v.reshape(-1) for _, v in items])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for _, v in items:
        n = v.numel()
        v.copy_(flat[o:o + n].view_as(v))
        o += n
    return grads
