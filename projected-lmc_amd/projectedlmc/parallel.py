"""Latent sharding across GPUs: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The q latent GPs of the projected model are independent given the projection, so each rank
factorises only its own latents -- there is no collective on the n x n data path.  Per training
step the only exchange is ONE small all-reduce (sum) of a fused buffer holding every parameter
gradient and the scalar loss share (O(q d + p^2) numbers: latency-bound, so a single fused
message instead of per-tensor calls); at prediction one all-reduce of the (2, n*, p) partial
mean / variance sums (projected_lmc.py:1144,1152 is where the cross-latent sum happens).
"""
import ctypes
import os

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


_direct = {"tried": False, "on": False, "device": None}


def _direct_rccl():
    """PLMC_COMM=rccl: the all-reduces go through the library's own RCCL communicator (plmc_comm_*, include/plmc.h) instead of
    torch.distributed, which is then used for the bootstrap only: the 128-byte unique id travels from rank 0 to the others
    together with a STATUS byte, and a second all-reduce collects every rank's `plmc_comm_init` result -- either every rank ends
    on the direct path or none does (a rank that failed alone would otherwise fall back to dist.all_reduce while the others wait
    in ncclAllReduce).  Decided once per process.  Default: off."""
    if _direct["tried"]:
        return _direct["on"]
    if os.environ.get("PLMC_COMM", "") != "rccl" or not is_distributed() or not torch.cuda.is_available():
        _direct["tried"] = True
        return False
    from . import _hip
    L = _hip.lib().cdll
    dev = torch.device("cuda", torch.cuda.current_device())
    on_dev = dist.get_backend() == "nccl"
    msg = torch.zeros(129, dtype=torch.uint8)                 # [status | unique id]
    if dist.get_rank() == 0:
        buf = (ctypes.c_char * 128)()
        if L.plmc_comm_unique_id(buf) == 0:
            msg[0] = 1
            msg[1:] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
    msg = msg.to(dev) if on_dev else msg
    dist.broadcast(msg, src=0)                                 # (every rank takes part, whatever rank 0's call returned)
    msg = msg.cpu()
    ok = 0
    if int(msg[0]) == 1:
        raw = bytes(msg[1:].numpy().tobytes())
        ok = int(L.plmc_comm_init(ctypes.c_char_p(raw), dist.get_rank(), dist.get_world_size()) == 0)
    agree = torch.tensor([ok], dtype=torch.int32)
    agree = agree.to(dev) if on_dev else agree
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)
    _direct["tried"] = True
    if int(agree.cpu()[0]) != 1:
        if ok:
            L.plmc_comm_destroy()
        import warnings
        warnings.warn("PLMC_COMM=rccl: the direct RCCL communicator could not be set up on every rank (%s); using torch.distributed"
                      % L.plmc_last_error().decode(), RuntimeWarning)
        return False
    _direct["on"], _direct["device"] = True, dev
    return True


def all_reduce_sum(t):
    if is_distributed():
        if t.is_cuda and t.dtype in (torch.float32, torch.float64) and t.is_contiguous() and _direct_rccl() and t.device == _direct["device"]:
            # (the communicator is bound to the device that was current at its creation: other tensors go through torch.distributed)
            from . import _hip
            _hip.lib().call("plmc_comm_allreduce_sum", t.dtype, _hip.ptr(t), t.numel(), _hip.stream_ptr(t.device))
        else:
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
    buf = all_reduce_sum(torch.cat(flat))
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
