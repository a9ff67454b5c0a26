"""ctypes binding of libplaner_hip.so (C ABI: include/planer_hip.h).

There is deliberately NO fallback: if the shared library is missing or no
MI355X is visible, every device operation raises.  The numpy path lives in
the reference itself (and in oracle/ for tests), never in this package.
"""
import ctypes
import os
from ctypes import (POINTER, byref, c_char_p, c_double, c_float, c_int, c_longlong,
                    c_size_t, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libplaner_hip.so")

PL_OK, PL_EINVAL, PL_EUNSUPPORTED, PL_ENOMEM, PL_EHIP, PL_ERCCL = range(6)
NONZERO_BLOCK = 2048          # include/planer_hip.h PL_NONZERO_BLOCK
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
ACT_RES_AFTER = 16
UNIQUE_ID_BYTES = 128

_P = c_void_p
_I = c_int
_Z = c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "pl_last_error": [],
    "pl_version": [],
    "pl_device_count": [POINTER(c_int)],
    "pl_ctx_create": [_I, POINTER(_P)],
    "pl_ctx_destroy": [_P],
    "pl_ctx_info": [_P, POINTER(c_int), POINTER(c_int), POINTER(c_size_t), c_char_p, _Z],
    "pl_ctx_pci_bus_id": [_P, c_char_p, _Z],
    "pl_sync": [_P],
    "pl_alloc": [_P, _Z, POINTER(_P)],
    "pl_free": [_P, _P],
    "pl_pool_stats": [_P, POINTER(c_size_t), POINTER(c_size_t)],
    "pl_pool_block": [_P, _P, POINTER(_P), POINTER(c_size_t)],
    "pl_pool_trim": [_P],
    "pl_h2d": [_P, _P, _P, _Z],
    "pl_d2h": [_P, _P, _P, _Z],
    "pl_d2d": [_P, _P, _P, _Z],
    "pl_h2d_staged": [_P, _P, _P, _P, _Z],
    "pl_h2d_direct": [_P, _P, _P, _Z],
    "pl_d2h_begin": [_P, _P, _P, _Z, POINTER(c_int)],
    "pl_d2h_finish": [_P, _I, _P],
    "pl_host_alloc": [_Z, POINTER(_P)],
    "pl_host_free": [_P],
    "pl_copy_threads": [POINTER(c_int)],
    "pl_memset": [_P, _P, _I, _Z],
    "pl_event_create": [_P, POINTER(_P)],
    "pl_event_record": [_P, _P],
    "pl_event_sync": [_P],
    "pl_event_elapsed_ms": [_P, _P, POINTER(c_float)],
    "pl_event_destroy": [_P],
    "pl_stream_wait": [_P, _P],
    "pl_stream_wait_event": [_P, _P],
    "pl_ctx_swap_streams": [_P, _P],
    "pl_capture_begin": [_P],
    "pl_capture_end": [_P, POINTER(_P)],
    "pl_graph_launch": [_P],
    "pl_graph_destroy": [_P],
    "pl_plan_build": [_P, _P, _Z, POINTER(_P)],
    "pl_plan_info": [_P, POINTER(c_int), POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t), POINTER(c_int)],
    "pl_plan_tensor": [_P, _I, _I, POINTER(_P), POINTER(c_size_t), POINTER(c_int), POINTER(c_int), POINTER(c_int)],
    "pl_plan_run": [_P, POINTER(_P), POINTER(_P)],
    "pl_plan_destroy": [_P],
    "pl_conv2d_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P] + [_I] * 9,
    "pl_conv2d_fused_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P] + [_I] * 9
                           + [_P, _P, _P, _I, c_double, _I],
    "pl_conv2d_prepare_weights_f32": [_P, _P, _I, _I, _I, _I, _P],
    "pl_conv2d_prepare_winograd_f32": [_P, _P, _I, _I, _P],
    "pl_nchw_to_q4_f32": [_P, _P, _P, _I, _I, _I],
    "pl_q4_to_nchw_f32": [_P, _P, _P, _I, _I, _I],
    "pl_conv2d_q4_filter_elems": [_I, _I, _I, _I, _I, POINTER(c_size_t)],
    "pl_conv2d_prepare_q4_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "pl_conv2d_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P] + [_I] * 9
                        + [_P, _P, _P, _I, c_double],
    "pl_conv2d_winograd_q4_filter_elems": [_I, _I, POINTER(c_size_t)],
    "pl_conv2d_prepare_winograd_q4_f32": [_P, _P, _I, _I, _P],
    "pl_conv2d_winograd_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, c_double],
    "pl_conv2d_winograd4_q4_filter_elems": [_I, _I, POINTER(c_size_t)],
    "pl_conv2d_prepare_winograd4_q4_f32": [_P, _P, _I, _I, _P],
    "pl_conv2d_winograd4_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, c_double],
    "pl_conv2d_q4_pair_f32": [_P, _P, _I, _I, _I, _I] + 2 * ([_P] + [_I] * 7 + [_P, _P, _P, _I, c_double, _P]),
    "pl_conv2d_wf4_filter_elems": [_I, _I, POINTER(c_size_t)],
    "pl_conv2d_prepare_wf4_f32": [_P, _P, _I, _I, _P],
    "pl_conv2d_wf4_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, c_double],
    "pl_wino4_elems": [_I, _I, _I, _I, POINTER(c_size_t)],
    "pl_wino4_chain_supported": [_P, _I, _I, _I, _I, POINTER(c_int)],
    "pl_wino4_input_q4_f32": [_P, _P, _I, _I, _I, _I, _P],
    "pl_wino4_gemm_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P],
    "pl_wino4_output_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, c_double, _P],
    "pl_wino4_chain_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, c_double, _P, _P],
    "pl_wino43_supported": [_I, _I, POINTER(c_int)],
    "pl_wino43_elems": [_I, _I, _I, _I, POINTER(c_size_t)],
    "pl_conv2d_winograd43_q4_filter_elems": [_I, _I, POINTER(c_size_t)],
    "pl_conv2d_prepare_winograd43_q4_f32": [_P, _P, _I, _I, _P],
    "pl_wino43_input_q4_f32": [_P, _P, _I, _I, _I, _I, _P],
    "pl_wino43_gemm_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P],
    "pl_wino43_output_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, c_double, _P],
    "pl_wino43_chain_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, c_double, _P, _P],
    "pl_conv2d_winograd43_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, c_double],
    "pl_conv1x1_wino_in_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, c_double, _I, _P],
    "pl_conv2d_rowpack_filter_elems": [_I, _I, _I, _I, POINTER(c_size_t)],
    "pl_conv2d_prepare_rowpack_f32": [_P, _P, _I, _I, _I, _I, _P],
    "pl_conv2d_rowpack_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, c_double],
    "pl_rowpack_input_elems": [_I, _I, _I, _I, _I, _I, _I, _I, POINTER(c_size_t)],
    "pl_rowpack_input_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I],
    "pl_conv2d_rowpacked_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, c_double],
    "pl_conv2d_rowpacked_pool_supported": [_I] * 10 + [POINTER(c_int)],
    "pl_conv2d_rowpacked_pool_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, c_double],
    "pl_conv2d_stem_pool_nchw_supported": [_I] * 10 + [POINTER(c_int)],
    "pl_conv2d_stem_nchw_filter_elems": [_I, POINTER(c_size_t)],
    "pl_conv2d_prepare_stem_nchw_f32": [_P, _P, _I, _P],
    "pl_conv2d_stem_pool_nchw_q4_f32": [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I, c_double, _I],
    "pl_conv2d_w1d4_q4_filter_elems": [_I, _I, POINTER(c_size_t)],
    "pl_conv2d_prepare_w1d4_q4_f32": [_P, _P, _I, _I, _P],
    "pl_conv2d_w1d4_q4_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, c_double],
    "pl_pool2d_q4_f32": [_P, _P, _P] + [_I] * 13,
    "pl_upsample_nearest_q4_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I],
    "pl_gap_q4_f32": [_P, _P, _P, _I, _I, _I],
    "pl_concat2_q4_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "pl_scale_shift_q4_f32": [_P, _P, _P, _P, _P, _I, _I, _I],
    "pl_set_autotune": [_P, _I],
    "pl_tune_cache_save": [_P, c_char_p],
    "pl_tune_cache_load": [_P, c_char_p, POINTER(c_int)],
    "pl_conv2d_set_config": [_P, _I, _I],
    "pl_conv2d_set_plan": [_P, _I, _I, _I, _I],
    "pl_conv2d_num_configs": [],
    "pl_conv2d_config_name": [_I, c_char_p, _Z],
    "pl_conv2d_last_plan": [_P, c_char_p, _Z],
    "pl_conv2d_last_extents": [_P, POINTER(c_longlong)],
    "pl_tune_stats": [_P, POINTER(c_int), POINTER(c_int)],
    "pl_gemm_f32": [_P, _P, _I, _I, _P, _I, _I, _P, _P],
    "pl_scale_shift_f32": [_P, _P, _P, _P, _P, _I, _I, _I],
    "pl_relu_f32": [_P, _P, _P, _Z],
    "pl_leakyrelu_f32": [_P, _P, _P, _Z, c_double],
    "pl_sigmoid_f32": [_P, _P, _P, _Z],
    "pl_add_f32": [_P, _P, _P, _P, _Z],
    "pl_add_channel_f32": [_P, _P, _P, _P, _I, _I, _I],
    "pl_pool2d_f32": [_P, _P, _P] + [_I] * 12,
    "pl_upsample_nearest_f32": [_P, _P, _P, _I, _I, _I, _I, _I],
    "pl_copy2d_f32": [_P, _P, _Z, _P, _Z, _Z, _Z],
    "pl_gap_f32": [_P, _P, _P, _I, _I],
    "pl_unary_f32": [_P, _P, _P, _Z, _I, c_double, c_double],
    "pl_binary_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "pl_binary_bcast_f32": [_P, _P, _P, _P, _I, POINTER(c_int), POINTER(ctypes.c_longlong), POINTER(ctypes.c_longlong), _I],
    "pl_upsample_linear_f32": [_P, _P, _P, _I, _I, _I, _I, _I, POINTER(ctypes.c_float)],
    "pl_resize_linear_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "pl_softmax_f32": [_P, _P, _P, _I, _I, _I],
    "pl_reduce_f32": [_P, _P, _P, _I, _I, _I],
    "pl_transpose_f32": [_P, _P, _P, _I, POINTER(c_int), POINTER(c_int)],
    "pl_strided_map_f32": [_P, _P, _P, _I, POINTER(c_int), POINTER(ctypes.c_longlong), POINTER(c_int), POINTER(c_int),
                           POINTER(c_int), POINTER(c_int), POINTER(c_int), c_double],
    "pl_compare_f32": [_P, _P, _P, _P, _Z, _I, _I, _I],
    "pl_where_f32": [_P, _P, _P, _P, _P, _Z, _I, _I],
    "pl_cast": [_P, _P, _P, _Z, _I, _I],
    "pl_gather_f32": [_P, _P, _P, _P, _I, _I, _I, _I],
    "pl_erf_lut_f32": [_P, _P, _P, _P, _Z],
    "pl_instancenorm_f32": [_P, _P, _P, _P, _I, _I, _I, c_double],
    "pl_scatter_rows_f32": [_P, _P, _P, _P, _P, _I, _I],
    "pl_nonzero_count": [_P, _P, _Z, _I, _P, _P],
    "pl_nonzero_write": [_P, _P, _Z, _I, _P, _P, _I, _P, c_longlong],
    "pl_topk_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "pl_lstm_cell_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I],
    "pl_resize_hwc_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "pl_tile_accumulate_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I],
    "pl_tile_normalise_f32": [_P, _P, _P, _I, _I, _I],
    "pl_splitk_reduce_f32": [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I, c_double],
    "pl_comm_unique_id": [_P],
    "pl_comm_init_rank": [_P, _I, _I, _P],
    "pl_comm_bcast": [_P, _P, _Z, _I],
    "pl_comm_allreduce_max_f32": [_P, _P, _Z],
    "pl_comm_allgather": [_P, _P, _P, _Z],
    "pl_comm_info": [_P, POINTER(c_int), POINTER(c_int)],
    "pl_comm_destroy": [_P],
}
_RESTYPE = {"pl_last_error": c_char_p}

_lib = None


class HipBackendError(RuntimeError):
    pass


class NotCapturable(ValueError):
    """An operator needed a host round trip (upload of a host-computed index list, a data-dependent output
    shape) while the forward pass was being captured into a hipGraph: the net runs eagerly instead."""


def load(path=None):
    """dlopen the library and declare every prototype; raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("PLANER_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise HipBackendError(
            "libplaner_hip.so not found at %s -- build it with "
            "`python -m planer_amd._build` (needs hipcc, gfx950). "
            "planer_amd has no CPU fallback." % path)
    # multi-process RCCL on this driver stack needs dmabuf IPC (the legacy path fails with hipIpcGetMemHandle: invalid
    # argument); must be in the environment before the HSA runtime comes up, i.e. before the first HIP call
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = ctypes.CDLL(path)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


def check(rc):
    if rc == PL_OK:
        return
    msg = load().pl_last_error()
    msg = msg.decode() if msg else "status %d" % rc
    if rc == PL_EINVAL:
        raise (NotCapturable if "during capture" in msg else ValueError)(msg)
    if rc == PL_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == PL_ENOMEM:
        raise MemoryError(msg)
    raise HipBackendError(msg)


_recorder = None        # planer_amd.export: sees every call that goes through `call` while a plan is being recorded


def call(name, *args):
    check(getattr(load(), name)(*args))
    if _recorder is not None:
        _recorder(name, args)


__all__ = ["load", "check", "call", "HipBackendError", "NotCapturable", "SIGNATURES", "LIB_PATH",
           "byref", "c_void_p", "c_int", "c_size_t", "c_float"]
