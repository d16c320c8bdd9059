"""ctypes binding of libflairhip.so (the C ABI declared in include/flairhip.h).

Only plain pointers and integers cross this boundary; torch tensors are reduced to
``data_ptr()`` by ``flairhip.ops``.  There is no CPU fallback: if the shared library is missing
the import of the product path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLAIRHIP_LIB selects another build of the same library (kernel A/B runs); there is still no fallback
LIB_PATH = os.environ.get("FLAIRHIP_LIB") or os.path.join(_HERE, "libflairhip.so")

BF16 = 0
F32 = 1
BCO_RING = 0x1000  # FFA_BCO_RING
BCO_THIN = 0x2000  # FFA_BCO_THIN
BCO_STEM = 0x8000  # FFA_BCO_STEM
ERR_UNSUPPORTED = -2  # FFA_ERR_UNSUPPORTED: no kernel for the requested shape (callers may fall back to another op)


class FlairHipError(RuntimeError):
    pass


class Tile(C.Structure):
    _fields_ = [
        ("left", C.c_double), ("bottom", C.c_double), ("right", C.c_double), ("top", C.c_double),
        ("x0", C.c_double), ("y0", C.c_double), ("x1", C.c_double), ("y1", C.c_double),
        ("row", C.c_longlong), ("col", C.c_longlong),
    ]


class Window(C.Structure):
    _fields_ = [("col_off", C.c_int), ("row_off", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("skip", C.c_int)]


_i, _ll, _p, _f, _d = C.c_int, C.c_longlong, C.c_void_p, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/flairhip.h one to one
SIGNATURES = {
    "ffa_last_error": (C.c_char_p, []),
    "ffa_version": (_i, []),
    "ffa_target_arch": (C.c_char_p, []),
    "ffa_conv_block_co": (_i, [_i, _i, _i, _i]),
    "ffa_conv_row_group": (_i, [_i]),
    "ffa_pack_conv_weight_bytes": (_ll, [_i, _i, _i, _i, _i]),
    "ffa_pack_conv_weight": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "ffa_pack_desc_bytes": (_i, []),
    "ffa_pack_desc_fill": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "ffa_pack_desc_fill_cols": (_i, [_p, _p, _p, _p] + [_i] * 12),
    "ffa_pack_conv_weights_batched": (_i, [_i, _p, _i, _p]),
    "ffa_conv2d": (_i, [_i, _p, _p, _p, _p, _p] + [_i] * 15 + [_p]),
    "ffa_conv_stat_rows": (_ll, [_i, _i, _i, _i, _i]),
    "ffa_conv_plan": (_i, [_i] * 7),
    "ffa_ring_conv3x3": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 7 + [_p]),
    "ffa_ring_stat_rows": (_ll, [_i, _i, _i, _i]),
    "ffa_ring_pack": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_conv2d_pro": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 9 + [_p]),
    "ffa_conv_wgrad_pro": (_i, [_i, _p, _p, _p, _p, _p] + [_i] * 11 + [_p, _ll, _p]),
    "ffa_thin_eligible": (_i, [_i] * 6),
    "ffa_stem_eligible": (_i, [_i] * 6),
    "ffa_stem_pack_bytes": (_ll, []),
    "ffa_stem_stat_rows": (_ll, [_i, _i, _i]),
    "ffa_stem_pack": (_i, [_p, _p, _p, _i, _i, _p]),
    "ffa_stem_conv7x7": (_i, [_p, _p, _p, _p, _p] + [_i] * 8 + [_p]),
    "ffa_thin_stat_rows": (_ll, [_i, _i, _i, _i]),
    "ffa_thin_conv3x3": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 9 + [_p]),
    "ffa_thin_conv3x3_pro": (_i, [_p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 9 + [_p]),
    "ffa_thin_pack": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_thin_pack_bytes": (_ll, [_i, _i]),
    "ffa_thin_pack_desc_bytes": (_i, []),
    "ffa_thin_pack_desc_fill": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i]),
    "ffa_thin_pack_batched": (_i, [_p, _i, _p]),
    "ffa_ring_pack_desc_bytes": (_i, []),
    "ffa_ring_pack_desc_fill": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    "ffa_ring_pack_batched": (_i, [_i, _p, _i, _p]),
    "ffa_conv2d_stats": (_i, [_i, _p, _p, _p, _p, _p, _p] + [_i] * 15 + [_p]),
    "ffa_conv2d_bnbwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 14 + [_p]),
    "ffa_bn_bwd_partials": (_i, [_i, _p, _p, _p, _ll, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _p, _ll, _p]),
    "ffa_conv2d_dgrad_upcat": (_i, [_i, _p, _p, _p, _p] + [_i] * 8 + [_p]),
    "ffa_conv2d_upcat": (_i, [_i, _p, _p, _p, _p, _p, _p] + [_i] * 9 + [_p]),
    "ffa_conv_wgrad_workspace_bytes": (_ll, [_i] * 9),
    "ffa_conv_wgrad": (_i, [_i, _p, _p, _p] + [_i] * 14 + [_p, _ll, _p]),
    "ffa_conv_wgrad_upcat": (_i, [_i, _p, _p, _p, _p] + [_i] * 8 + [_p, _ll, _p]),
    "ffa_conv_is_persistent": (_i, [_i] * 11),
    "ffa_reflect_pad1": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "ffa_group_norm": (_i, [_i, _p, _p, _p, _p, _p, _ll, _i, _ll, _ll, _i, _ll, _i, _i, _f, _i, _p]),
    "ffa_positional_encoding": (_i, [_p, _p, _i, _i, _i, _f, _p]),
    "ffa_add_rowvec": (_i, [_i, _p, _p, _i, _i, _i, _p]),
    "ffa_ltae_attention": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ffa_temporal_aggregate": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ffa_detect_pad_images": (_i, [_p, _p, _i, _ll, _f, _p]),
    "ffa_mask_images": (_i, [_i, _p, _p, _i, _ll, _f, _p]),
    "ffa_linear": (_i, [_i, _p, _ll, _p, _p, _p, _ll, _p, _ll, _i, _i, _i, _i, _p]),
    "ffa_space_to_depth": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_linear_ex": (_i, [_i, _p, _ll, _p, _p, _p, _ll, _p, _ll, _i, _i, _i, _i, _p, _ll, _p, _i, _p]),
    "ffa_linear_wgrad_workspace_bytes": (_ll, [_i, _i, _i]),
    "ffa_linear_wgrad": (_i, [_i, _p, _ll, _p, _ll, _p, _p, _i, _i, _i, _i, _p, _ll, _p]),
    "ffa_layer_norm": (_i, [_i, _p, _p, _p, _p, _p, _ll, _i, _f, _p]),
    "ffa_patch_merge_norm": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "ffa_layer_norm_bwd_workspace_bytes": (_ll, [_ll, _i]),
    "ffa_layer_norm_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _p, _ll, _p]),
    "ffa_scale_rows": (_i, [_i, _p, _p, _p, _ll, _i, _i, _p]),
    "ffa_updown2x_slice": (_i, [_i, _p, _p] + [_i] * 8 + [_p]),
    "ffa_column_sums_workspace_bytes": (_ll, [_ll, _i]),
    "ffa_column_sums": (_i, [_i, _p, _p, _ll, _i, _p, _ll, _p]),
    "ffa_patch_merge_norm_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _ll, _p]),
    "ffa_window_attention_bwd_workspace_bytes": (_ll, [_i] * 6),
    "ffa_window_attention_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _ll, _p]),
    "ffa_bilinear_slice_bwd": (_i, [_i, _p, _p] + [_i] * 9 + [_p]),
    "ffa_adaptive_avg_pool_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_window_attention": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "ffa_gelu": (_i, [_i, _p, _p, _ll, _p]),
    "ffa_adaptive_avg_pool": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_bilinear_slice": (_i, [_i, _p, _p, _p] + [_i] * 9 + [_p]),
    "ffa_reflect_pad1_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p]),
    "ffa_group_norm_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _ll, _i, _ll, _ll, _i, _ll, _i, _i, _f, _i, _p]),
    "ffa_ltae_attention_train": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ffa_ltae_attention_bwd_blocks": (_i, [_i, _i, _i]),
    "ffa_ltae_attention_bwd": (_i, [_i] + [_p] * 11 + [_i] * 6 + [_p]),
    "ffa_temporal_aggregate_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ffa_mul": (_i, [_i, _p, _p, _p, _ll, _p]),
    "ffa_mean_stack": (_i, [_i, _p, _i, C.c_float, _p, _ll, _p]),
    "ffa_ktime_begin": (_i, [_i]),
    "ffa_ktime_end": (_i, [_p, _p, _i]),
    "ffa_adamw_multi": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_float, C.c_float, _i, _i, _p]),
    "ffa_bn_workspace_bytes": (_ll, [_i]),
    "ffa_bn_stats": (_i, [_i, _p, _ll, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _ll, _p]),
    "ffa_bn_finalize": (_i, [_p, _ll, _ll, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _ll, _p]),
    "ffa_bn_eval_params": (_i, [_i, _p, _p, _p, _p, _f, _p, _p, _p]),
    "ffa_bn_apply": (_i, [_i, _p, _p, _p, _p, _p, _ll, _i, _i, _p]),
    "ffa_bn_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p, _ll, _p]),
    "ffa_bn_bwd_stages": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p, _ll, _i, _p]),
    "ffa_bn_bwd_fused": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p, _ll, _p, _p]),
    "ffa_channel_sums": (_i, [_i, _p, _ll, _i, _p, _p, _p, _ll, _p]),
    "ffa_maxpool3x3s2_fwd": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ffa_maxpool3x3s2_bwd": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ffa_nchw_to_nhwc": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_u8_nchw_to_nhwc": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ffa_raw_nchw_to_nhwc": (_i, [_i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ffa_nhwc_to_nchw": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_upsample_nearest2x_concat_fwd": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_upsample_nearest2x_concat_bwd": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ffa_bilinear_fwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ffa_bilinear_bwd_workspace_bytes": (_ll, [_i, _i, _i, _i]),
    "ffa_bilinear_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _ll, _p]),
    "ffa_softmax_ce_workspace_bytes": (_ll, []),
    "ffa_softmax_ce": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p, _ll, _p]),
    "ffa_softmax_ce_sums": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _i, _p, _ll, _p]),
    "ffa_scale_inplace": (_i, [_i, _p, _ll, _p, _p]),
    "ffa_predict_u8": (_i, [_i, _i, _p, _p] + [_i] * 9 + [_p]),
    "ffa_onehot_to_index": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "ffa_confusion_matrix": (_i, [_p, _p, _ll, _i, _p, _p]),
    "ffa_slice_grid": (_ll, [_d, _d, _d, _d, _d, _d, _i, _i, _d, C.POINTER(Tile), _ll]),
    "ffa_write_window": (_i, [_d, _d, _d, _d, _d, _d, _d, _i, _i, C.POINTER(Window)]),
    "ffa_tiff_lzw_bound": (_ll, [_ll]),
    "ffa_tiff_lzw_decode": (_ll, [_p, _ll, _p, _ll]),
    "ffa_tiff_lzw_encode": (_ll, [_p, _ll, _p, _ll]),
    "ffa_tiff_hpredict": (_i, [_p, _ll, _ll, _i, _i, _i]),
    "ffa_probe_tr16": (_i, [_p, _p, _p]),
    "ffa_probe_mfma": (_i, [_p, _p, _p, _i, _p]),
}

_lib = None


def load() -> C.CDLL:
    """Load libflairhip.so once; raise FlairHipError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlairHipError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback for the HIP hot path")
    # torch first: its bundled HIP runtime must be the one already in the process when libflairhip.so (linked against
    # libamdhip64) is mapped -- loaded the other way round the library binds to a second, uninitialised runtime and
    # every launch fails with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ffa_last_error().decode("utf-8", "replace")
        raise FlairHipError(f"{what or 'libflairhip'} failed (code {rc}): {msg}")
