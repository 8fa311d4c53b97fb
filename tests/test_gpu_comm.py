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


def test_rccl_is_found_by_a_process_that_never_imports_torch(gpu_device):
    """the stated purpose of ng_comm_*: a caller WITHOUT torch.  On this image RCCL exists under /opt/rocm and inside the torch
    wheel; a fresh interpreter that loads only libnmrgnn_hip.so must get a unique id and a one-rank communicator (comm.hip:
    rccl_load walks NG_RCCL_PATH, already-mapped copies, the loader path, /opt/rocm, the wheel's torch/lib)."""
    import os
    import subprocess
    import sys
    from nmrgnn_amd import _lib
    code = r"""
import ctypes as C, sys
assert 'torch' not in sys.modules
lib = C.CDLL(sys.argv[1])
uid = (C.c_char * 128)()
rc = lib.ng_comm_unique_id(uid)
assert rc == 0, rc
h = C.c_void_p()
assert lib.ng_ctx_create(0, C.byref(h)) == 0
assert lib.ng_comm_init(h, 0, 1, uid) == 0
assert lib.ng_comm_world(h) == 1
assert lib.ng_comm_destroy(h) == 0
lib.ng_ctx_destroy(h)
assert 'torch' not in sys.modules
print('ok')
"""
    for extra in ({}, {"NG_RCCL_PATH": "/opt/rocm/lib"}):
        env = dict(os.environ, **extra)
        res = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], capture_output=True, text=True, timeout=300, env=env)
        assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-1500:]
