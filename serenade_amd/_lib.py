"""ctypes binding of libserenade_hip.so (the C ABI declared in include/serenade_hip.h).

There is no fallback: if the library cannot be loaded the product path raises.  The CPU
oracle under ``oracle/`` is never imported from here.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SERENADE_AMD_LIB") or os.path.join(_HERE, "libserenade_hip.so")  # env: developer builds

SRN_MAX_TAPS = 16
ACT_NONE, ACT_LEAKY, ACT_SILU, ACT_MISH = 0, 1, 2, 3
RES_NONE, RES_ADD, RES_AXPY = 0, 1, 2
POST_NONE, POST_DIV, POST_TANH, POST_RELU, POST_LEAKY = 0, 1, 2, 3, 4
PREC_FP32, PREC_BF16X3, PREC_BF16X6 = 0, 1, 2


class SrnConvParams(ctypes.Structure):
    _fields_ = [
        ("n_batch", c_int32), ("n_head", c_int32), ("T_in", c_int32), ("T_out", c_int32),
        ("C_in", c_int32), ("C_in0", c_int32), ("C_w", c_int32), ("N", c_int32), ("N_out", c_int32),
        ("n_taps", c_int32), ("tap_off", c_int32 * SRN_MAX_TAPS), ("in_stride", c_int32),
        ("pad_reflect", c_int32), ("w_nmajor", c_int32), ("pro_act", c_int32), ("pro_slope", c_float),
        ("alpha", c_float), ("beta", c_float), ("geglu", c_int32), ("res_mode", c_int32), ("post", c_int32), ("post_div", c_float),
        ("out_t_stride", c_int32), ("out_t_off", c_int32), ("tile", c_int32),
        ("in0", c_void_p), ("in0_bs", c_int64), ("in0_hs", c_int64), ("ld_in0", c_int32),
        ("in1", c_void_p), ("in1_bs", c_int64), ("ld_in1", c_int32),
        ("w", c_void_p), ("w_bs", c_int64), ("w_hs", c_int64), ("ldw", c_int32),
        ("bias", c_void_p), ("len_in", c_void_p), ("len_out", c_void_p),
        ("res", c_void_p), ("res_bs", c_int64), ("res_hs", c_int64), ("ld_res", c_int32),
        ("res2", c_void_p), ("res2_bs", c_int64), ("ld_res2", c_int32),
        ("out", c_void_p), ("out_bs", c_int64), ("out_hs", c_int64), ("ld_out", c_int32),
        ("precision", c_int32), ("no_halo", c_int32), ("ws", c_void_p), ("ws_bytes", c_int64),
        ("w_hi", c_void_p), ("w_lo", c_void_p), ("gn_partials", c_void_p),
        ("out_tr", c_void_p), ("out_tr_bs", c_int64), ("ld_out_tr", c_int32), ("out_tr_col0", c_int32),
    ]


class SrnResUnitParams(ctypes.Structure):
    _fields_ = [
        ("n_batch", c_int32), ("T", c_int32), ("C", c_int32), ("k", c_int32), ("dilation", c_int32),
        ("slope", c_float), ("x", c_void_p), ("x_bs", c_int64),
        ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
        ("w1_hi", c_void_p), ("w2_hi", c_void_p), ("res2", c_void_p), ("res2_bs", c_int64),
        ("post_div", c_float), ("out", c_void_p), ("out_bs", c_int64), ("precision", c_int32),
    ]


SRN_COPY_LIST_MAX = 160


class SrnCopyList(ctypes.Structure):
    _fields_ = [("n", c_int32), ("pad_", c_int32), ("src", c_void_p * SRN_COPY_LIST_MAX),
                ("off", c_int64 * SRN_COPY_LIST_MAX), ("len", c_int64 * SRN_COPY_LIST_MAX)]


class SrnWorldParams(ctypes.Structure):
    _fields_ = [
        ("n_batch", c_int32), ("max_frames", c_int32), ("fs", c_int32), ("fft_size", c_int32),
        ("x", c_void_p), ("x_bs", c_int64), ("x_len", c_void_p),
        ("f0", c_void_p), ("t", c_void_p), ("f_bs", c_int64), ("n_frames", c_void_p), ("twiddle", c_void_p),
        ("q1", c_double), ("f0_floor", c_double), ("threshold", c_double), ("unvoiced_db", c_double),
        ("band_window", c_void_p), ("band_window_len", c_int32), ("n_bands", c_int32),
        ("out0", c_void_p), ("out0_bs", c_int64), ("ld_out0", c_int32),
        ("out1", c_void_p), ("out1_bs", c_int64), ("ld_out1", c_int32),
    ]


SRN_TR_LIST_MAX = 40


class SrnTransposeList(ctypes.Structure):
    _fields_ = [("n", c_int32), ("pad_", c_int32), ("src", c_void_p * SRN_TR_LIST_MAX), ("dst", c_void_p * SRN_TR_LIST_MAX),
                ("src_bs", c_int64 * SRN_TR_LIST_MAX), ("dst_bs", c_int64 * SRN_TR_LIST_MAX),
                ("B", c_int32 * SRN_TR_LIST_MAX), ("R", c_int32 * SRN_TR_LIST_MAX), ("Cc", c_int32 * SRN_TR_LIST_MAX),
                ("ld_src", c_int32 * SRN_TR_LIST_MAX), ("ld_dst", c_int32 * SRN_TR_LIST_MAX),
                ("first_block", c_int32 * (SRN_TR_LIST_MAX + 1))]


class SrnTnGemmParams(ctypes.Structure):
    _fields_ = [
        ("n_batch", c_int32), ("n_head", c_int32), ("n_items", c_int32), ("T_a", c_int32), ("T_b", c_int32),
        ("stride", c_int32), ("n_shifts", c_int32), ("shift", c_int32 * SRN_MAX_TAPS), ("M", c_int32), ("N", c_int32),
        ("a", c_void_p), ("a_bs", c_int64), ("a_hs", c_int64), ("a_is", c_int64), ("lda", c_int32),
        ("b", c_void_p), ("b_bs", c_int64), ("b_hs", c_int64), ("b_is", c_int64), ("ldb", c_int32),
        ("out", c_void_p), ("out_bs", c_int64), ("out_hs", c_int64), ("ldc", c_int32),
        ("alpha", c_float), ("ws", c_void_p), ("ws_bytes", c_int64),
        ("n_inner", c_int32), ("a_is2", c_int64), ("b_is2", c_int64), ("len_b", c_void_p), ("colsum", c_void_p),
    ]


class SrnExcitationParams(ctypes.Structure):
    _fields_ = [
        ("n_batch", c_int32), ("max_frames", c_int32), ("fs", c_int32), ("hop", c_int32),
        ("f0", c_void_p), ("df_f0", c_void_p), ("f_bs", c_int64), ("n_frames", c_void_p), ("phase_ws", c_void_p),
        ("noise", c_void_p), ("sine", c_void_p), ("sine_amp", c_float), ("noise_amp", c_float), ("n_df", c_int32),
        ("dfs", c_void_p * 4), ("df_upsample", c_int32 * 4), ("dense_factors", c_double * 4),
    ]


_P = c_void_p
_SIGS = {
    "srn_abi_version": (c_int, []),
    "srn_last_error": (c_char_p, []),
    "srn_conv_gemm": (c_int, [POINTER(SrnConvParams), _P]),
    "srn_conv_gemm_workspace_bytes": (c_int64, [POINTER(SrnConvParams)]),
    "srn_hifigan_resunit": (c_int, [POINTER(SrnResUnitParams), _P]),
    "srn_gn_mish_apply": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "srn_resblock_tail": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int, c_int, c_int, c_int,
                                  c_float, c_float, c_int, _P]),
    "srn_resblock_tail_ln": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int, c_int, c_int, c_int,
                                     c_float, c_float, c_int, _P, _P, _P, c_float, _P]),
    "srn_scatter_rows": (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "srn_layernorm": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "srn_softmax_rows": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "srn_sinusoidal_emb": (c_int, [_P, _P, c_int, c_int, c_int, c_float, _P]),
    "srn_copy_channels": (c_int, [_P, c_int64, c_int, c_int, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P]),
    "srn_transpose_ct": (c_int, [_P, _P, c_int, c_int, c_int, c_int64, c_int, c_int64, c_int, _P]),
    "srn_transpose_multi": (c_int, [POINTER(SrnTransposeList), _P]),
    "srn_weight_norm_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "srn_weight_norm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "srn_renorm": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "srn_out_conv_tanh": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "srn_pd_gather": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P]),
    "srn_pad_signal": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "srn_logmel": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, c_float, c_int, _P]),
    "srn_loudness": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P]),
    "srn_gru_recur_last": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "srn_style_token_attention_kv": (c_int, [_P] * 8 + [c_int] * 5 + [_P]),
    "srn_rowln_fwd": (c_int, [_P, _P, c_int64, _P, c_int64, _P, c_int, c_int, c_int, c_float, _P]),
    "srn_rowln_bwd": (c_int, [_P, _P, _P, c_int64, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "srn_rowln_chunks": (c_int, [c_int]),
    "srn_gn_mish_bwd_partial": (c_int, [_P] * 8 + [c_int] * 4 + [_P]),
    "srn_gn_mish_bwd_apply": (c_int, [_P] * 9 + [c_int] * 4 + [_P]),
    "srn_gn_chunks": (c_int, [c_int]),
    "srn_gn_stats": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "srn_chunk_colsum": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "srn_colsum": (c_int, [_P, _P, _P, c_int, c_int64, c_int, c_int, _P]),
    "srn_colsum_chunks": (c_int, [c_int64]),
    "srn_softmax_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, c_float, _P]),
    "srn_geglu_fwd": (c_int, [_P, _P, c_int64, c_int, _P]),
    "srn_geglu_bwd": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "srn_adamw": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P]),
    "srn_adamw_dyn": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, _P, _P]),
    "srn_sumsq": (c_int, [_P, c_int64, _P, _P]),
    "srn_dot": (c_int, [_P, _P, c_int64, _P, _P]),
    "srn_multi_copy": (c_int, [POINTER(SrnCopyList), _P, _P]),
    "srn_sumsq_blocks": (c_int, [c_int64]),
    "srn_bn_chunks": (c_int, [c_int64]),
    "srn_bn_relu_fwd": (c_int, [_P] * 8 + [c_int64, c_int, c_float, c_float, _P]),
    "srn_bn_relu_bwd": (c_int, [_P] * 8 + [c_int64, c_int, _P]),
    "srn_im2col_s2": (c_int, [_P, _P] + [c_int] * 5 + [_P]),
    "srn_col2im_s2": (c_int, [_P, _P] + [c_int] * 5 + [_P]),
    "srn_gru_train_fwd": (c_int, [_P] * 5 + [c_int] * 3 + [_P]),
    "srn_gru_train_bwd": (c_int, [_P] * 6 + [c_int] * 3 + [_P]),
    "srn_token_attn_fwd": (c_int, [_P] * 5 + [c_int] * 4 + [_P]),
    "srn_token_attn_bwd": (c_int, [_P] * 8 + [c_int] * 4 + [_P]),
    "srn_tn_gemm": (c_int, [POINTER(SrnTnGemmParams), _P]),
    "srn_tn_gemm_workspace_bytes": (c_int64, [POINTER(SrnTnGemmParams)]),
    "srn_world_cheaptrick": (c_int, [POINTER(SrnWorldParams), _P]),
    "srn_world_d4c": (c_int, [POINTER(SrnWorldParams), _P]),
    "srn_world_project": (c_int, [_P, c_int64, c_int, c_int, _P, c_int, c_int, _P, c_int, _P]),
    "srn_world_pack_features": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int64, c_int, _P]),
    "srn_wave_to_f64": (c_int, [_P, _P, c_int64, c_int, _P]),
    "srn_f0_match_length": (c_int, [_P, c_int64, _P, _P, c_int64, _P, c_int, c_int, _P]),
    "srn_cont_f0": (c_int, [_P, c_int64, _P, _P, _P, _P, c_int, _P]),
    "srn_sifigan_excitation": (c_int, [POINTER(SrnExcitationParams), _P]),
}

EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m serenade_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.srn_abi_version() != 3:
            raise RuntimeError("libserenade_hip.so ABI version mismatch")
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().srn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libserenade_hip {what} failed (rc={rc}): {msg}")
