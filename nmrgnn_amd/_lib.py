"""ctypes binding of libnmrgnn_hip.so (the C ABI declared in include/nmrgnn_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMRGNN_HIP_LIB: another build of the same library (A/B measurements of kernel variants on one GPU box)
LIB_PATH = os.environ.get("NMRGNN_HIP_LIB") or os.path.join(_HERE, "csrc", "libnmrgnn_hip.so")

NG_ACT_NONE, NG_ACT_SOFTPLUS, NG_ACT_RELU, NG_ACT_TANH = 0, 1, 2, 3
ACT_CODES = {None: NG_ACT_NONE, "linear": NG_ACT_NONE, "softplus": NG_ACT_SOFTPLUS, "relu": NG_ACT_RELU,
             "tanh": NG_ACT_TANH}

_c_float_p = C.POINTER(C.c_float)
_c_int32_p = C.POINTER(C.c_int32)
_vp = C.c_void_p
_i64 = C.c_int64
_u64 = C.c_uint64
_int = C.c_int
_f = C.c_float

# name -> (restype, argtypes); must list every symbol include/nmrgnn_hip.h declares
SIGNATURES = {
    "ng_abi_version": (_int, []),
    "ng_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "ng_ctx_destroy": (None, [_vp]),
    "ng_last_error": (C.c_char_p, [_vp]),
    "ng_ctx_reserve": (_int, [_vp, _u64]),
    "ng_replay_arm": (_int, [_vp, _int]),
    "ng_replay_stage": (_int, [_vp, _vp, _u64, _f, _f, _f, _i64, _int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u64)]),
    "ng_replay_token": (_int, [_vp, C.POINTER(_u64)]),
    "ng_replay_commit": (_int, [_vp, _u64]),
    "ng_reload_env": (_int, []),
    "ng_weights_frozen": (_int, [_vp, _int]),
    "ng_weights_changed": (_int, [_vp]),
    "ng_defer_reductions": (_int, [_vp, _vp, _int]),
    "ng_flush_reductions": (_int, [_vp, _vp]),
    "ng_ctx_set_graph_span": (_int, [_vp, _i64]),
    "ng_prof_enable": (_int, [_vp, _int]),
    "ng_prof_reset": (_int, [_vp]),
    "ng_prof_read": (_int, [_vp, _int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "ng_randn": (_int, [_vp, _vp, _u64, _u64, _vp, _i64]),
    "ng_dropout_mask": (_int, [_vp, _vp, _u64, _u64, _f, _vp, _i64]),
    "ng_add_scaled": (_int, [_vp, _vp, _i64, _vp, _vp, _f, _vp]),
    "ng_add_noise": (_int, [_vp, _vp, _u64, _u64, _i64, _vp, _f, _vp]),
    "ng_rbf_expand": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _f, _vp]),
    "ng_edge_mlp_fwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _f,
                               C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "ng_edge_tape_layout": (_int, [_int, _int, _int, _int, _i64]),
    "ng_edge_table_range": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _f, _vp]),
    "ng_edge_table_points": (_int, [_vp, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "ng_edge_table_check": (_int, [_vp, _vp, _int, _int, _vp, _f, _vp, _vp, _vp, _int, _vp, _vp]),
    "ng_edge_table_interp": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_edge_table_scatter": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_edge_mlp_bwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _f,
                               C.POINTER(_vp), _vp, _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "ng_edge_mlp_bwd_tape": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _f,
                                    C.POINTER(_vp), _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _int]),
    "ng_build_live_edges": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ng_build_graph_lists": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_graph_lists_one_launch": (_int, [_i64, _int]),
    "ng_add_noise_live": (_int, [_vp, _vp, _u64, _u64, _i64, _vp, _vp, _f, _vp, _vp]),
    "ng_edge_live_supported": (_int, [_int, _int, _int, _int]),
    "ng_edge_mlp_fwd_live": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _f,
                                    C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "ng_edge_mlp_bwd_live": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _f,
                                    C.POINTER(_vp), _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _int]),
    "ng_embed_fwd": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp]),
    "ng_embed_bwd": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp]),
    "ng_mp_aggregate": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "ng_mp_layer_fwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp]),
    "ng_mp_layer_bwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "ng_mp_layer_bwd_rec": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp]),
    "ng_mp_edge_records": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp]),
    "ng_mp_aggregate_csr": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "ng_mp_layer_fwd_csr": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp]),
    "ng_mp_layer_bwd_csr": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "ng_build_incoming_lists": (_int, [_vp, _vp, _i64, _int, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ng_incoming_lists_scratch_bytes": (C.c_size_t, [_i64, _i64]),
    "ng_cutoff_count": (_int, [_vp, _vp, _int, _int, _f, _vp, _vp]),
    "ng_cutoff_fill": (_int, [_vp, _vp, _int, _int, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ng_cutoff_fill_rows": (_int, [_vp, _vp, _int, _int, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_exclusive_scan_i32": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "ng_dense_fwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "ng_dense_bwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp]),
    "ng_head_fwd": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_head_fwd_dropout": (_int, [_vp, _vp, _i64, _int, _int, _vp, _u64, _u64, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_head_bwd": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_head_loss_blocks": (_int, [_vp, _int, _int, _int, _i64]),
    "ng_head_loss_bwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _i64, _vp, _u64, _u64, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _f, _vp, _vp, _vp]),
    "ng_head_loss_reduce": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "ng_mp_layer_wants_aggregate": (_int, [_int, _int, _int]),
    "ng_fc_block_fwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "ng_fc_block_scratch_floats": (_i64, [_i64, _int, _int]),
    "ng_fc_block_bwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_mp_layer_fwd_short": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_mp_layer_fwd_short_csr": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_mp_layer_short_ok": (_int, [_i64, _int, _int, _int]),
    "ng_fc_head_ok": (_int, [_i64, _int, _int, _int, _int]),
    "ng_fc_head_fwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _vp,
                              _vp, _vp, _vp]),
    "ng_knn_graph": (_int, [_vp, _vp, _int, _int, _int, _f, _vp, _vp, _vp, _vp]),
    "ng_amp_attend": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_amp_attend_bwd": (_int, [_vp, _vp, _i64, _int, _int, _int] + [_vp] * 13),
    "ng_loss_l2": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ng_loss_name": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "ng_adam_step": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i64, _f]),
    "ng_comm_unique_id": (_int, [_vp]),
    "ng_comm_init": (_int, [_vp, _int, _int, _vp]),
    "ng_comm_destroy": (_int, [_vp]),
    "ng_comm_world": (_int, [_vp]),
    "ng_allreduce_grads": (_int, [_vp, _vp, _vp, _i64]),
}

_lib = None
_lock = threading.Lock()


class NGError(RuntimeError):
    pass


def load():
    """dlopen the engine; raises NGError loudly when it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NGError(
                f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C nmrgnn_amd/csrc`. nmrgnn_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.ng_abi_version() != 9:
            raise NGError("libnmrgnn_hip.so ABI version mismatch")
        _lib = lib
        return lib


def reload_env():
    """make the library parse its NG_* path switches again (it reads them once per process)"""
    load().ng_reload_env()


class Context:
    """Owns one ng_ctx (scratch workspace + profiling state) for one device."""

    def __init__(self, device_index: int):
        self.lib = load()
        h = _vp()
        rc = self.lib.ng_ctx_create(int(device_index), C.byref(h))
        if rc != 0 or not h:
            raise NGError(f"ng_ctx_create(device={device_index}) failed with code {rc} "
                          "(no HIP device? nmrgnn_amd needs an AMD GPU)")
        self.handle = h
        self.device_index = device_index

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.ng_last_error(self.handle)
            raise NGError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ng_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- profiling ------------------------------------------------------------------
    def prof_enable(self, on=True):
        self.check(self.lib.ng_prof_enable(self.handle, 1 if on else 0), "ng_prof_enable")

    def prof_reset(self):
        self.check(self.lib.ng_prof_reset(self.handle), "ng_prof_reset")

    def prof_read(self):
        cap = 64
        names = (C.c_char_p * cap)()
        tot = (C.c_double * cap)()
        cnt = (_i64 * cap)()
        n = self.lib.ng_prof_read(self.handle, cap, names, tot, cnt)
        if n < 0:
            self.check(n, "ng_prof_read")
        return {names[i].decode(): (tot[i], cnt[i]) for i in range(n)}


_contexts = {}


def get_context(device_index: int) -> Context:
    ctx = _contexts.get(device_index)
    if ctx is None:
        ctx = Context(device_index)
        _contexts[device_index] = ctx
    return ctx


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return None
    return _vp(t.data_ptr())


def ptr_array(tensors):
    arr = (_vp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
