"""Direct RCCL entry points of the C ABI (plmc_comm_*, include/plmc.h): the exchange of the sharded path for a host without
torch.distributed.  One GPU is what a test box has, so this is the world-size-1 case: RCCL is opened at run time, a
communicator is built from a unique id, the in-place sum all-reduce returns the buffer unchanged on the caller's stream, and the
error paths answer with a code and a message instead of crashing.  (World size 2 of the same sharded step: the gloo CPU test,
tests/test_sharding_gloo.py; the 8-GPU run is the driver's.)"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_direct_rccl_allreduce_world_one():
    from projectedlmc import _hip
    L = _hip.lib()
    C = L.cdll
    dev = torch.device("cuda:0")
    assert C.plmc_comm_world() == 0 and C.plmc_comm_rank() == -1
    x = torch.arange(1000, dtype=torch.float64, device=dev)
    with pytest.raises(RuntimeError, match="plmc_comm_init has not been called"):
        L.call("plmc_comm_allreduce_sum", torch.float64, _hip.ptr(x), x.numel(), _hip.stream_ptr(dev))
    uid = (ctypes.c_char * 128)()
    assert C.plmc_comm_unique_id(uid) == 0, C.plmc_last_error()
    assert any(b != 0 for b in uid.raw)
    assert C.plmc_comm_init(uid, 1, 1) != 0                     # rank outside the world
    assert C.plmc_comm_init(uid, 0, 1) == 0, C.plmc_last_error()
    try:
        assert C.plmc_comm_world() == 1 and C.plmc_comm_rank() == 0
        assert C.plmc_comm_init(uid, 0, 1) != 0                 # one communicator per process
        for dt in (torch.float64, torch.float32):
            x = torch.linspace(-3, 5, 4097, dtype=dt, device=dev)
            ref = x.clone()
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                L.call("plmc_comm_allreduce_sum", dt, _hip.ptr(x), x.numel(), _hip.stream_ptr(dev))
            s.synchronize()
            assert torch.equal(x, ref)
    finally:
        assert C.plmc_comm_destroy() == 0
    assert C.plmc_comm_world() == 0
