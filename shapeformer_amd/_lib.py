"""ctypes binding of libsfmi.so — the C-ABI drop-in boundary (include/sfmi.h).

The product path has NO fallback: if the library is missing or a symbol is
absent, import-time / call-time errors are raised loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsfmi.so")

_lib = None
SFMI_OK, SFMI_EINVAL = 0, -1      # include/sfmi.h

c_f32p = C.c_void_p
c_ptr = C.c_void_p
i32, i64, sz, f32 = C.c_int, C.c_longlong, C.c_size_t, C.c_float

# name -> (restype, argtypes)   -- mirrors include/sfmi.h
PROTOTYPES = {
    "sfmi_version": (i32, []),
    "sfmi_stream_spin": (i32, [i64, c_ptr]),
    "sfmi_stream_create_cumask": (i32, [c_ptr, i32, c_ptr]),
    "sfmi_stream_destroy": (i32, [c_ptr]),
    "sfmi_hwid_probe": (i32, [c_ptr, i32, i32, i64, c_ptr]),
    "sfmi_tune_set": (i32, [C.c_char_p, i32]),
    "sfmi_tune_get": (i32, [C.c_char_p]),
    "sfmi_tune_generation": (i32, []),
    # SDF query
    "sfmi_relu_bwd_f32": (i32, [c_ptr, c_ptr, c_ptr, i64, c_ptr]),
    "sfmi_lincomb_f32": (i32, [f32, c_ptr, f32, c_ptr, c_ptr, i64, c_ptr]),
    "sfmi_affine2_cl_f32": (i32, [c_ptr] * 6 + [i32, i64, i32, c_ptr]),
    "sfmi_chan_dot_stats_f32": (i32, [c_ptr, c_ptr, c_ptr, i32, i64, i32, i32, c_ptr]),
    "sfmi_upsample2_cl_f32": (i32, [c_ptr, c_ptr] + [i32] * 5 + [c_ptr]),
    "sfmi_sumpool2_cl_f32": (i32, [c_ptr, c_ptr] + [i32] * 7 + [c_ptr]),
    "sfmi_maxpool2_bwd_cl_f32": (i32, [c_ptr] * 4 + [i32] * 5 + [c_ptr]),
    "sfmi_cells_f32": (i32, [c_ptr, c_ptr, c_ptr, i32, i32, i32, c_ptr]),
    "sfmi_cell_max_f32": (i32, [c_ptr] * 4 + [i32, i32, i64, i32, i32, i32, c_ptr]),
    "sfmi_cell_scatter_add_f32": (i32, [c_ptr] * 4 + [i32, i32, i64, i32, i32, i32, c_ptr]),
    "sfmi_cell_max_bwd_f32": (i32, [c_ptr] * 5 + [i32, i32, i64, i32, i32, i32, c_ptr]),
    "sfmi_cell_mean_f32": (i32, [c_ptr] * 3 + [i32, i64, i32, c_ptr]),
    "sfmi_cell_mean_bwd_f32": (i32, [c_ptr] * 4 + [i32, i32, i64, i32, c_ptr]),
    "sfmi_trilinear_cl_f32": (i32, [c_ptr] * 3 + [i32, i64, i32, i32, c_ptr]),
    "sfmi_trilinear_bwd_cl_f32": (i32, [c_ptr] * 3 + [i32, i64, i32, i32, c_ptr]),
    "sfmi_bce_logits_f32": (i32, [c_ptr] * 4 + [i64, f32, c_ptr]),
    "sfmi_vq_stats_f32": (i32, [c_ptr] * 4 + [i64, i32, c_ptr]),
    "sfmi_vq_ema_update_f32": (i32, [c_ptr] * 5 + [i32, i32, f32, f32, c_ptr]),
    "sfmi_conv3d_wgrad_f32": (i32, [c_ptr] * 3 + [i32] * 12 + [c_ptr]),
    "sfmi_blas_available": (i32, []),
    "sfmi_sgemm_f32": (i32, [i32] * 5 + [f32, c_ptr, i32, c_ptr, i32, f32, c_ptr, i32, c_ptr]),
    "sfmi_gemm_blas_f32": (i32, [c_ptr] * 5 + [i32] * 4 + [c_ptr]),
    "sfmi_mc_workspace_bytes": (sz, [i32, i32]),
    "sfmi_mc_count_f32": (i32, [c_ptr, f32, i32, i32, c_ptr, c_ptr, c_ptr]),
    "sfmi_mc_emit_f32": (i32, [c_ptr, f32, i32, i32, c_ptr, c_ptr] + [f32] * 6 + [c_ptr, c_ptr, c_ptr]),
    "sfmi_sdf_pack_floats": (sz, []),
    "sfmi_sdf_pack_weights": (i32, [c_ptr] * 11),
    "sfmi_sdf_query_f32": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i32, i64, i32, i32, c_ptr]),
    "sfmi_sdf_query_grid_f32": (i32, [c_ptr, i32, c_ptr, c_ptr, c_ptr, i32, i32, i32, c_ptr]),
    "sfmi_sigmoid_f32": (i32, [c_ptr, c_ptr, i64, c_ptr]),
    "sfmi_sdf_query_grid_slab_f32": (i32, [c_ptr, i32, i32, i32, c_ptr, c_ptr, c_ptr, i32, i32, i32, c_ptr]),
    "sfmi_sdf_query_grid_aff_f32": (i32, [c_ptr, i32, i32, i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i32, i32, i32, c_ptr]),
    # encoder (per-point path)
    "sfmi_enc_pack_floats": (sz, []),
    "sfmi_enc_pack_weights": (i32, [c_ptr] * 10),
    "sfmi_enc_workspace_bytes": (sz, [i32, i32]),
    "sfmi_encode_points_f32": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i32, i32, i32, c_ptr]),
    "sfmi_encode_points_down_f32": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i32, i32, i32, i32, c_ptr]),
    "sfmi_encode_points_tap_f32": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, i32, i32, i32, c_ptr, c_ptr, c_ptr]),
    # conv / groupnorm / pooling
    "sfmi_conv_pack_weight": (i32, [c_ptr, i32, i32, i32, c_ptr]),
    "sfmi_conv3d_cl_f32": (i32, [c_ptr] * 6 + [i32] * 11 + [c_ptr]),
    "sfmi_conv_pack_weight_subpixel": (i32, [c_ptr, i32, i32, c_ptr]),
    "sfmi_conv3d_up2_cl_f32": (i32, [c_ptr] * 6 + [i32] * 7 + [c_ptr]),
    "sfmi_conv3d_cl_stats_f32": (i32, [c_ptr] * 6 + [i32] * 11 + [c_ptr, c_ptr, c_ptr]),
    "sfmi_conv3d_up2_cl_stats_f32": (i32, [c_ptr] * 6 + [i32] * 7 + [c_ptr, c_ptr, c_ptr]),
    "sfmi_groupnorm_coeffs_partial_f32": (i32, [c_ptr] * 5 + [i32, i32, i32, i32, i32, C.c_float, c_ptr]),
    "sfmi_gn_splits": (i32, [i32]),
    "sfmi_groupnorm_coeffs_f32": (i32, [c_ptr] * 6 + [i32, i32, i32, i32, C.c_float, c_ptr]),
    "sfmi_affine_cl_f32": (i32, [c_ptr] * 4 + [i32, i64, i32, c_ptr]),
    "sfmi_maxpool2_cl_f32": (i32, [c_ptr, c_ptr, i32, i32, i32, i32, i32, c_ptr]),
    "sfmi_upcat_cl_f32": (i32, [c_ptr, c_ptr, c_ptr, i32, i32, i32, i32, i32, i32, c_ptr]),
    # vector quantiser
    "sfmi_vq_pack_floats": (sz, [i32, i32]),
    "sfmi_vq_pack_codebook": (i32, [c_ptr, i32, i32, c_ptr]),
    "sfmi_vq_argmin_f32": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, i32, c_ptr]),
    "sfmi_vq_gather_f32": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, c_ptr]),
    # tokens
    "sfmi_mode_i32": (i32, [c_ptr, i64, i32, i32, c_ptr, c_ptr, c_ptr]),
    "sfmi_apply_mask_i32": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i64, i32, c_ptr]),
    "sfmi_dense2sparse_i32": (i32, [c_ptr, c_ptr, i32, c_ptr, c_ptr, i32, i32, i32, i32, i32, i32, c_ptr]),
    "sfmi_sparse2dense_i32": (i32, [c_ptr, c_ptr, c_ptr, c_ptr, i32, c_ptr, i32, i32, i32, i32, i32, c_ptr]),
    "sfmi_ar_n_extra_i32": (i32, [c_ptr, c_ptr, c_ptr, i32, i32, i32, i32, c_ptr]),
    # transformer
    "sfmi_gemm_f32": (i32, [c_ptr] * 5 + [i64, i32, i32, i32, i64, i64, c_ptr]),
    "sfmi_decode_gemm_padded_rows": (i32, [i32]),
    "sfmi_skinny16_pack_floats": (sz, [i32, i32]),
    "sfmi_skinny16_pack_weight": (i32, [c_ptr, i32, i32, c_ptr]),
    "sfmi_ln_fold_pack_f32": (i32, [c_ptr] * 7 + [i32, i32, c_ptr]),
    "sfmi_gpt_embed_f32": (i32, [c_ptr] * 15 + [i32] * 5 + [c_ptr, i32, c_ptr]),
    "sfmi_gpt_rowprep_f32": (i32, [c_ptr] * 12 + [i32] * 5 + [c_ptr, i32, c_ptr]),
    "sfmi_sgemm_mfma_splits": (i32, [i32, i32, i32]),
    "sfmi_sgemm_sk_tile": (i32, [i32, i32, i32]),
    "sfmi_sgemm_sk_slab_floats": (i64, []),
    "sfmi_sgemm_sk_cnt_ints": (i64, [i32, i32]),
    "sfmi_sgemm_sk_f32": (i32, [i32] * 5 + [c_ptr, i32, c_ptr, i32, c_ptr, c_ptr, i32, i32, c_ptr, i32, c_ptr, c_ptr, f32, C.c_uint, c_ptr, i64, c_ptr, i64,
                                c_ptr]),
    "sfmi_sgemm_mfma_f32": (i32, [i32] * 5 + [c_ptr, i32, c_ptr, i32, c_ptr, i32, i32, c_ptr, i32, c_ptr, c_ptr, i64, f32, C.c_uint, c_ptr]),
    "sfmi_ce_rows_f32": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, i32, c_ptr]),
    "sfmi_gpt_attn_decode_f32": (i32, [c_ptr] * 6 + [i32] * 5 + [c_ptr, c_ptr]),
    "sfmi_gpt_attn_decode_gated_f32": (i32, [c_ptr] * 5 + [i32] * 4 + [c_ptr, c_ptr, c_ptr, i32, c_ptr, c_ptr]),
    "sfmi_gpt_attn_prefill_f32": (i32, [c_ptr] * 5 + [i32] * 5 + [c_ptr, f32, C.c_uint, c_ptr]),
    "sfmi_gpt_attn_prefill_lse_f32": (i32, [c_ptr] * 5 + [i32] * 5 + [c_ptr, f32, C.c_uint, c_ptr, c_ptr]),
    "sfmi_gpt_sample_f32": (i32, [c_ptr] * 12 + [i32] * 10 + [C.c_float, C.c_float] + [i32] * 4 + [C.c_uint, c_ptr, i32, i32, i32, i32, c_ptr]),
    "sfmi_gpt_mask_logits_f32": (i32, [c_ptr] * 5 + [i32] * 9 + [c_ptr]),
    "sfmi_decode_gemm_f32": (i32, [c_ptr] * 6 + [i32] * 8 + [c_ptr, c_ptr, c_ptr]),
    "sfmi_decode_gemm_prof_f32": (i32, [c_ptr] * 6 + [i32] * 8 + [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "sfmi_decode_gemm_slab_floats": (sz, [i32, i32, i32]),
    "sfmi_gpt_embed_packed_f32": (i32, [c_ptr] * 9 + [i32] * 4 + [c_ptr]),
    "sfmi_set_len_i32": (i32, [c_ptr, c_ptr, i32, i32, c_ptr]),
    # training step (csrc/train.hip)
    "sfmi_transpose_f32": (i32, [c_ptr, c_ptr, i32, i32, i32, i32, c_ptr]),
    "sfmi_colsum_f32": (i32, [c_ptr, c_ptr, i32, i32, i32, i32, c_ptr]),
    "sfmi_colsum_slices": (i32, [i32, i32]),
    "sfmi_colsum_ws_f32": (i32, [c_ptr, c_ptr, i32, i32, i32, i32, c_ptr, c_ptr]),
    "sfmi_layernorm_bwd_scratch_floats": (sz, [i32, i32]),
    "sfmi_gelu_f32": (i32, [c_ptr, c_ptr, i64, c_ptr]),
    "sfmi_gelu_bwd_f32": (i32, [c_ptr, c_ptr, c_ptr, i64, c_ptr]),
    "sfmi_layernorm_bwd_f32": (i32, [c_ptr] * 8 + [i32, i32, c_ptr]),
    "sfmi_layernorm_bwd_rows_f32": (i32, [c_ptr] * 6 + [i32, i32, c_ptr]),
    "sfmi_layernorm_bwd_rows_drop_f32": (i32, [c_ptr] * 7 + [f32, C.c_uint, i32, i32, c_ptr]),
    "sfmi_col_reduce_slices": (i32, [i32]),
    "sfmi_col_reduce_part_floats": (i64, [i32, i32]),
    "sfmi_col_reduce_f32": (i32, [i32] + [c_ptr] * 8 + [i32, i32, c_ptr, i64, c_ptr, i64, c_ptr]),
    "sfmi_ce_fwd_bwd_f32": (i32, [c_ptr] * 4 + [i32] * 5 + [C.c_float, c_ptr]),
    "sfmi_attn_bwd_f32": (i32, [c_ptr] * 5 + [i32] * 4 + [f32, C.c_uint, c_ptr]),
    "sfmi_attn_train_fwd_small_f32": (i32, [c_ptr] * 3 + [i32] * 4 + [f32, C.c_uint, c_ptr]),
    "sfmi_attn_bwd_lse_f32": (i32, [c_ptr] * 6 + [i32] * 4 + [f32, C.c_uint, c_ptr]),
    "sfmi_dropout_f32": (i32, [c_ptr, c_ptr, i64, f32, C.c_uint, c_ptr]),
    "sfmi_embed_scatter_f32": (i32, [c_ptr, c_ptr, c_ptr, i64, i32, c_ptr]),
    "sfmi_fixed_to_float_f32": (i32, [c_ptr, c_ptr, i64, i32, c_ptr]),
    "sfmi_add_f32": (i32, [c_ptr, c_ptr, c_ptr, i64, c_ptr]),
    "sfmi_adamw_f32": (i32, [c_ptr] * 4 + [i64] + [C.c_float] * 5 + [i32, c_ptr]),
    "sfmi_adamw_multi_f32": (i32, [c_ptr] * 6 + [i32, c_ptr, c_ptr, c_ptr, f32, f32, f32, f32, i32, c_ptr]),
    "sfmi_adamw_multi_shard_f32": (i32, [c_ptr] * 6 + [i32, c_ptr, c_ptr, c_ptr, f32, f32, f32, f32, i32, c_ptr, c_ptr]),
    # device-resident step state (captured training step)
    "sfmi_sgemm_sk_sd_f32": (i32, [i32] * 5 + [c_ptr, i32, c_ptr, i32, c_ptr, c_ptr, i32, i32, c_ptr, i32, c_ptr, c_ptr, f32, C.c_uint, c_ptr, c_ptr, i64,
                                   c_ptr, i64, c_ptr]),
    "sfmi_gpt_attn_prefill_lse_sd_f32": (i32, [c_ptr] * 5 + [i32] * 5 + [c_ptr, f32, C.c_uint, c_ptr, c_ptr, c_ptr]),
    "sfmi_attn_train_fwd_small_sd_f32": (i32, [c_ptr] * 3 + [i32] * 4 + [f32, C.c_uint, c_ptr, c_ptr]),
    "sfmi_attn_bwd_lse_sd_f32": (i32, [c_ptr] * 6 + [i32] * 4 + [f32, C.c_uint, c_ptr, c_ptr]),
    "sfmi_layernorm_bwd_rows_drop_sd_f32": (i32, [c_ptr] * 7 + [f32, C.c_uint, c_ptr, i32, i32, c_ptr]),
    "sfmi_dropout_sd_f32": (i32, [c_ptr, c_ptr, i64, f32, C.c_uint, c_ptr, c_ptr]),
    "sfmi_adamw_bias_corrections": (i32, [f32, f32, i32, c_ptr]),
    "sfmi_adamw_multi_shard_bc_f32": (i32, [c_ptr] * 6 + [i32, c_ptr, c_ptr, c_ptr, f32, f32, f32, f32, i32, c_ptr, c_ptr, c_ptr]),
    "sfmi_unflatten_multi_f32": (i32, [c_ptr] * 5 + [i32, c_ptr, c_ptr]),
}


class SfmiError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfmiError(
                f"{LIB_PATH} not found: the HIP extension is REQUIRED (no CPU/eager fallback). "
                "Build it with `python -m shapeformer_amd.build`.")
        # torch's bundled HIP runtime (same SONAME libamdhip64.so.7) must be the one in the process BEFORE
        # libsfmi.so is loaded, otherwise two runtimes coexist and our launches see "no device" (hipError 100).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise SfmiError(f"libsfmi.so does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise SfmiError(f"{what} failed with code {rc}")


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array, or None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


_raw_stream = None


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  Called once per kernel launch (~900 times per training step):
    torch.cuda.current_stream() builds a Stream object per call; the raw-stream binding the compiler back ends use returns the
    same handle as a plain int."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get is not None and dev is not None:
            _raw_stream = lambda: get(dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return _raw_stream()
