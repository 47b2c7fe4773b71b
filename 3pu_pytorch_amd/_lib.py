"""ctypes binding of lib3pu_hip.so (the C ABI declared in include/tpu3.h).

There is no CPU fallback anywhere in this package: if the HIP library is missing or a tensor
is not on a ROCm device, the call raises.  PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib3pu_hip.so")

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class KnnLayout(ctypes.Structure):
    """tpu3_knn_layout (include/tpu3.h)."""
    _fields_ = [("n_arr", _vp), ("m_arr", _vp), ("pts_of", _vp), ("grp", _vp),
                ("bp", _i), ("groups", _i), ("cand", _vp), ("cand_count", _vp),
                ("tile_pts", _vp), ("tile_idx", _vp), ("tile_box", _vp)]


# name -> (restype, argtypes); exactly the functions include/tpu3.h declares
SIGNATURES = {
    "tpu3_version": (ctypes.c_char_p, []),
    "tpu3_strerror": (ctypes.c_char_p, [_i]),
    "tpu3_fps_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "tpu3_fps_ragged_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "tpu3_fps_workspace_bytes": (_sz, [_i, _i]),
    "tpu3_gather_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tpu3_gather_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tpu3_ball_query": (_i, [_vp, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp]),
    "tpu3_nmdist_fwd_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tpu3_debug_nmdist_form": (_i, [_i]),
    "tpu3_debug_nmdist_grid_calls": (ctypes.c_long, [_i]),
    "tpu3_debug_nmdist_grid_stats": (_i, [_vp]),
    "tpu3_nmdist_bwd_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tpu3_chamfer_reduce_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "tpu3_knn_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, ctypes.POINTER(KnnLayout), _vp, _vp,
                          _vp, _i, _vp, _vp]),
    "tpu3_knn_unique_prepare_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, ctypes.POINTER(KnnLayout),
                                         _vp, _vp, _vp, _sz]),
    "tpu3_knn_unique_workspace_bytes": (_sz, [_i, _i]),
    "tpu3_interlevel_skip_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _i,
                                      _vp, _sz]),
    "tpu3_interlevel_skip_st_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _i,
                                         _vp, _sz, _i]),
    "tpu3_interlevel_skip_train_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp,
                                            _vp, _sz]),
    "tpu3_interlevel_skip_bwd_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp]),
    "tpu3_interlevel_skip_workspace_bytes": (_sz, [_i, _i, _i]),
    "tpu3_linear_small_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i]),
    "tpu3_linear_small_st_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i]),
    "tpu3_dec_train_fwd_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tpu3_dec_train_bwd_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i, _vp, _vp, _vp, _sz]),
    "tpu3_gather_rows_f32": (_i, [_vp, _i, _i, ctypes.c_long, _i, _vp, _vp, _i, _vp]),
    "tpu3_scatter_add_rows_f32": (_i, [_vp, _i, _i, ctypes.c_long, _i, _vp, _vp, _i, _vp]),
    "tpu3_linear_wide_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i]),
    "tpu3_split_bf16": (_i, [_i]),
    "tpu3_linear_wide_split_bytes": (ctypes.c_size_t, [_i]),
    "tpu3_linear_wide_split_bf16": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "tpu3_linear_wide_sb_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _vp, _vp, _i]),
    "tpu3_linear_lift_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i]),
    "tpu3_linear_wgrad_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _sz]),
    "tpu3_linear_wgrad_workspace_bytes": (_sz, [ctypes.c_long]),
    "tpu3_linear_wgrad_bias_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz]),
    "tpu3_linear_wgrad_bias_workspace_bytes": (_sz, [ctypes.c_long, _i, _i]),
    "tpu3_dec_train_wgrad_f32": (_i, [_vp, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "tpu3_dec_train_wgrad_workspace_bytes": (_sz, [ctypes.c_long]),
    "tpu3_knn_tiles_build_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "tpu3_knn_tiles_workspace_bytes": (_sz, [_i, _i]),
    "tpu3_knn_tiles_query_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "tpu3_linear_dgrad_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _i, _vp, _vp, _i]),
    "tpu3_regress_tail_f32": (_i, [_vp, ctypes.c_long, _i] + [_vp] * 10 + [_i]),
    "tpu3_knn_unique_compact_i32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "tpu3_knn_graph_self_f32": (_i, [_vp, _i, _i, _i, _i, _vp, ctypes.POINTER(KnnLayout), _vp, _vp, _vp, _vp, _sz]),
    "tpu3_knn_graph_self_optimistic_f32": (_i, [_vp, _i, _i, _i, _i, _vp, ctypes.POINTER(KnnLayout), _vp, _vp]),
    "tpu3_knn_graph_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, ctypes.POINTER(KnnLayout), _vp, _vp, _vp]),
    "tpu3_normalize_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "tpu3_normalize_cl_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "tpu3_repatch_filter_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tpu3_repatch_seeds_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tpu3_gather_xyz_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i]),
    "tpu3_denormalize_f32": (_i, [_vp, ctypes.c_long, _i, _vp, _vp, _vp, _vp]),
    "tpu3_fill_f32_i32": (_i, [_vp, _vp, ctypes.c_long, _f, _vp, ctypes.c_long, _i]),
    "tpu3_debug_fps_bucket_events": (_i, [_vp, _vp]),
    "tpu3_debug_fps_level_stats": (_i, [_vp]),
    "tpu3_debug_fps_tile_stats": (_i, [_vp]),
    "tpu3_debug_knn_tiles_stats": (_i, [_vp]),
    "tpu3_debug_knn_slab_launches": (ctypes.c_long, [_i]),
    "tpu3_debug_fps_cluster": (_i, [_i]),
    "tpu3_debug_dec_split": (_i, [_i]),
    "tpu3_debug_skip_fused": (_i, [_i]),
    "tpu3_debug_fps_plan": (_i, [_i, _i, _i, ctypes.POINTER(ctypes.c_int)]),
    "tpu3_fps_cluster_faults": (ctypes.c_long, [_i]),
    "tpu3_debug_fps_cluster_absent": (_i, [_i]),
    "tpu3_dense_edge_conv_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _i, _i]),
    "tpu3_dense_edge_conv_st_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _i, _i, _i]),
    "tpu3_dense_edge_conv_fold_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tpu3_dense_edge_conv_pack_floats": (_sz, [_i]),
    "tpu3_dense_edge_conv_pack_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "tpu3_dense_edge_conv_pk_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i]),
    "tpu3_dense_edge_conv_fold_pk_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i,
                                              _i, _vp]),
}

MFMA_F32, MFMA_F16 = 0, 1        # TPU3_MFMA_* of include/tpu3.h
STORE_F32, STORE_F16 = 0, 1      # TPU3_STORE_* of include/tpu3.h
EINVAL, ELIMIT = -1, -2          # TPU3_EINVAL / TPU3_ELIMIT

_lib = None


def lib():
    """Load lib3pu_hip.so once; raise (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "lib3pu_hip.so is missing at %s -- run `python -c \"import __graft_entry__ as g; "
                "g.build()\"` (there is no CPU fallback)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s (code %d)" % (what, lib().tpu3_strerror(code).decode(), code))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_of(t):
    """hipStream_t of torch's current stream on the tensor's device."""
    return torch.cuda.current_stream(t.device).cuda_stream


def require_device(t, name):
    # mirrors CHECK_CUDA / CHECK_CONTIGUOUS of sampling/sampling.cpp:20-24 (RuntimeError)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def require_dtype(t, dtype, name):
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
