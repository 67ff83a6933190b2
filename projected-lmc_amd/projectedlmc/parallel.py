"""Latent sharding across GPUs: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The q latent GPs of the projected model are independent given the projection, so each rank
factorises only its own latents -- there is no collective on the n x n data path.  Per training
step the only exchange is ONE small all-reduce (sum) of a fused buffer holding every parameter
gradient and the scalar loss share (O(q d + p^2) numbers: latency-bound, so a single fused
message instead of per-tensor calls); at prediction one all-reduce of the (2, n*, p) partial
mean / variance sums (projected_lmc.py:1144,1152 is where the cross-latent sum happens).
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_of(rank=None, world=None):
    if rank is None:
        if not is_distributed():
            return None
        rank, world = dist.get_rank(), dist.get_world_size()
    return (rank, world)


def all_reduce_sum(t):
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def sync_loss_and_grads(loss, params):
    """After `loss.backward()` on every rank: all-reduce (sum) loss share and all gradients in one
    fused buffer.  Returns the global loss (detached).  Every rank ends with identical gradients,
    so identical optimiser steps keep the replicated parameters in sync."""
    params = [p for p in params if p.requires_grad]
    if not is_distributed():
        return loss.detach()
    bdt = torch.float64 if any(p.dtype == torch.float64 for p in params) else torch.float32
    flat = [loss.detach().reshape(1).to(bdt)]
    for p in params:
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        flat.append(g.reshape(-1).to(bdt))
    buf = torch.cat(flat)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    off = 1
    for p in params:
        k = p.numel()
        g = buf[off:off + k].reshape(p.shape).to(p.dtype)
        if p.grad is None:
            p.grad = g
        else:
            p.grad.copy_(g)
        off += k
    return buf[0].to(loss.dtype)
