"""ng_comm_* / ng_allreduce_grads: the C-ABI gradient exchange.  One GPU per box here, so what
can be checked is the binding: RCCL is found and loaded at first use, a communicator of ONE rank is created through
ncclCommInitRank, the in-place all-reduce runs on the caller's stream and is the identity, errors come back as codes.
The multi-rank exchange itself has never run (DESIGN §6 says so); the Python host side uses torch.distributed."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_allreduce_in_a_world_of_one_through_rccl(gpu_device):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    ctx = _lib.Context(0)                     # its own context: the communicator belongs to it
    lib = ctx.lib
    try:
        assert lib.ng_comm_world(ctx.handle) == 1
        g = torch.arange(114605, dtype=torch.float32, device=gpu_device) * 0.5 - 7.0     # the F=64 gradient bucket size
        ref = g.clone()
        st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
        # no communicator: identity, no RCCL needed
        ctx.check(lib.ng_allreduce_grads(ctx.handle, st, ptr(g), g.numel()), "allreduce (no comm)")
        torch.cuda.synchronize()
        assert torch.equal(g, ref)
        uid = (C.c_char * 128)()
        ctx.check(lib.ng_comm_unique_id(uid), "unique id")
        assert any(b != b"\x00" for b in uid)
        ctx.check(lib.ng_comm_init(ctx.handle, 0, 1, uid), "comm init")
        assert lib.ng_comm_world(ctx.handle) == 1
        for _ in range(3):
            ctx.check(lib.ng_allreduce_grads(ctx.handle, st, ptr(g), g.numel()), "allreduce")
        torch.cuda.synchronize()
        assert torch.equal(g, ref)
        # a second communicator on the same context is refused with a message
        assert lib.ng_comm_init(ctx.handle, 0, 1, uid) == -1
        assert b"already" in lib.ng_last_error(ctx.handle)
        ctx.check(lib.ng_comm_destroy(ctx.handle), "destroy")
        # bad rank
        assert lib.ng_comm_init(ctx.handle, 3, 2, uid) == -1
    finally:
        ctx.close()
