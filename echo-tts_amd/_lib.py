"""ctypes binding of libechohip.so (the C ABI declared in include/echo_hip.h).

The product path has NO CPU fallback: importing this module never computes anything, but every
entry point raises if the HIP library is missing or no gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ECHO_LIB_PATH") or os.path.join(HERE, "libechohip.so")   # override: debugging builds only

ECHO_F32, ECHO_BF16 = 0, 1
ABI_VERSION = 7

c_i64 = C.c_int64
vp = C.c_void_p


class EchoConfig(C.Structure):
    _fields_ = [
        ("precision", C.c_int),
        ("latent_size", C.c_int), ("model_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
        ("intermediate_size", C.c_int),
        ("norm_eps", C.c_float),
        ("text_vocab_size", C.c_int), ("text_model_size", C.c_int), ("text_num_layers", C.c_int),
        ("text_num_heads", C.c_int), ("text_intermediate_size", C.c_int),
        ("speaker_patch_size", C.c_int), ("speaker_model_size", C.c_int), ("speaker_num_layers", C.c_int),
        ("speaker_num_heads", C.c_int), ("speaker_intermediate_size", C.c_int),
        ("timestep_embed_size", C.c_int), ("adaln_rank", C.c_int),
        ("has_latent_encoder", C.c_int),
        ("dac_latent_dim", C.c_int), ("dac_decoder_dim", C.c_int), ("dac_n_rates", C.c_int), ("dac_rates", C.c_int * 8),
        ("dac_post_layers", C.c_int), ("dac_post_heads", C.c_int), ("dac_post_head_dim", C.c_int),
        ("dac_post_ffn", C.c_int), ("dac_post_window", C.c_int),
        ("dac_n_up", C.c_int), ("dac_up_factors", C.c_int * 4),
        ("dac_norm_eps", C.c_float),
        ("dac_enc_dim", C.c_int), ("dac_enc_n_rates", C.c_int), ("dac_enc_rates", C.c_int * 8), ("dac_enc_tlayers", C.c_int * 8),
        ("dac_enc_window", C.c_int),
        ("dac_n_codebooks", C.c_int), ("dac_codebook_size", C.c_int), ("dac_codebook_dim", C.c_int), ("dac_semantic_size", C.c_int),
        ("dit_fp8", C.c_int),
    ]


class EchoStep(C.Structure):
    _fields_ = [("has_cfg", C.c_int), ("dt", C.c_float), ("rescale", C.c_int), ("r_inv1mt", C.c_float),
                ("r_ratio", C.c_float), ("r_1mt", C.c_float), ("kv_unscale_after", C.c_int)]


class EchoSamplerParams(C.Structure):
    _fields_ = [("B", C.c_int), ("S", C.c_int), ("num_steps", C.c_int), ("start_pos", C.c_int), ("use_latent", C.c_int),
                ("cfg_scale_text", C.c_float), ("cfg_scale_speaker", C.c_float), ("has_truncation", C.c_int), ("init_scale", C.c_float),
                ("kv_scale", C.c_float), ("kv_max_layers", C.c_int),
                ("steps", C.POINTER(EchoStep)), ("temb", vp)]


class EchoGemmDesc(C.Structure):
    _fields_ = [("A", vp), ("W", vp), ("C", vp), ("C2", vp),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("Npad", C.c_int),
                ("lda", c_i64), ("ldw", c_i64), ("ldc", c_i64),
                ("taps", C.c_int), ("tap_base", C.c_int), ("tap_shift", C.c_int),
                ("nbatch", C.c_int), ("nbi", C.c_int),
                ("a_bo", c_i64), ("a_bi", c_i64), ("w_bo", c_i64), ("w_bi", c_i64), ("c_bo", c_i64), ("c_bi", c_i64),
                ("acc_scale", C.c_float),
                ("bias", vp), ("bias_bo", c_i64), ("bias_bi", c_i64), ("vec_mod", C.c_int),
                ("div", C.c_float), ("act", C.c_int),
                ("colscale", vp),
                ("res", vp), ("ldres", c_i64), ("res_bo", c_i64), ("res_bi", c_i64),
                ("snake_alpha", vp),
                ("store_main", C.c_int), ("swiglu", C.c_int),
                ("cfg", C.c_int), ("ksplit", C.c_int), ("ws", vp), ("ws_bytes", c_i64), ("split3", C.c_int),
                ("fp8", C.c_int), ("a_scale", vp), ("w_scale", vp),
                ("qkv_mode", C.c_int), ("qkv_D", C.c_int), ("qkv_S", C.c_int), ("rope_heads", C.c_int), ("pos0", C.c_int), ("qk_eps", C.c_float),
                ("qk_w", vp), ("rope", vp), ("vt", vp), ("vt_ld", c_i64), ("vt_row_stride", c_i64), ("w_presplit", C.c_int),
                ("a_scale_const", C.c_float), ("c8", vp), ("c8_ld", c_i64), ("c8_inv", C.c_float), ("qkv_gate_act", C.c_int)]


class EchoAttnSeg(C.Structure):
    _fields_ = [("K", vp), ("k_ld", c_i64), ("k_row_stride", c_i64), ("k_head_stride", c_i64),
                ("Vt", vp), ("vt_ld", c_i64), ("vt_row_stride", c_i64), ("vt_head_stride", c_i64),
                ("nkeys", vp), ("bias", vp), ("bias_row_stride", c_i64), ("kv_mod", C.c_int)]


class EchoAttnDesc(C.Structure):
    _fields_ = [("Q", vp), ("q_ld", c_i64), ("q_row_stride", c_i64),
                ("O", vp), ("o_ld", c_i64), ("o_row_stride", c_i64),
                ("G", vp), ("g_ld", c_i64), ("g_row_stride", c_i64),
                ("S", C.c_int), ("H", C.c_int), ("rows", C.c_int), ("nseg", C.c_int),
                ("seg", EchoAttnSeg * 4), ("causal", C.c_int), ("scale", C.c_float), ("prof", vp), ("redo", vp),
                ("O8", vp), ("o8_ld", c_i64), ("o8_row_stride", c_i64), ("o8_inv", C.c_float), ("g_activated", C.c_int)]


class EchoProfile(C.Structure):
    _fields_ = [("ms_mod", C.c_float), ("ms_steps", C.c_float), ("ms_total", C.c_float), ("ms_gemm_sum", C.c_float),
                ("n_gemm", C.c_int), ("ms_pp_sum", C.c_float), ("n_pp", C.c_int), ("flops_pp", C.c_double),
                ("ms_attn_sum", C.c_float), ("n_attn", C.c_int)]


# name -> (restype, argtypes); every symbol include/echo_hip.h declares
SIGNATURES = {
    "echo_abi_version": (C.c_int, []),
    "echo_last_error": (C.c_char_p, [vp]),
    "echo_ctx_create": (C.c_int, [C.POINTER(EchoConfig), C.c_int, C.POINTER(vp)]),
    "echo_ctx_destroy": (None, [vp]),
    "echo_load_tensor": (C.c_int, [vp, C.c_char_p, vp, C.c_int, C.c_int, C.POINTER(c_i64), C.c_int]),
    "echo_finalize_dit": (C.c_int, [vp, vp]),
    "echo_finalize_dac": (C.c_int, [vp, vp]),
    "echo_set_rope_table": (C.c_int, [vp, vp, C.c_int]),
    "echo_set_ae_rope_table": (C.c_int, [vp, vp, C.c_int]),
    "echo_encode_text": (C.c_int, [vp, vp, vp, C.POINTER(C.c_int32), C.c_int, C.c_int, vp]),
    "echo_encode_speaker": (C.c_int, [vp, vp, vp, C.POINTER(C.c_int32), C.c_int, C.c_int, vp]),
    "echo_encode_latent_prefix": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_long, vp]),
    "echo_scale_speaker_kv": (C.c_int, [vp, C.c_float, C.c_int, vp]),
    "echo_dit_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), vp, vp]),
    "echo_dit_forward_t": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp, vp]),
    "echo_sample_euler": (C.c_int, [vp, C.POINTER(EchoSamplerParams), vp, vp, vp]),
    "echo_dac_decode": (C.c_int, [vp, vp, C.c_int, C.c_float, vp, vp]),
    "echo_dac_decode_zq": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "echo_dac_decode_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, c_i64, vp]),
    "echo_set_pca": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "echo_dac_hop": (C.c_int, [vp]),
    "echo_finalize_dac_encoder": (C.c_int, [vp, vp]),
    "echo_dac_encode": (C.c_int, [vp, vp, C.c_long, vp, vp, vp, vp]),
    "echo_set_pca_encode": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, vp]),
    "echo_op_gemm": (C.c_int, [C.c_int, C.POINTER(EchoGemmDesc), vp]),
    "echo_op_quant_rows_fp8": (C.c_int, [vp, c_i64, vp, c_i64, vp, C.c_int, C.c_int, vp]),
    "echo_op_norm_adaln_fp8": (C.c_int, [vp, c_i64, vp, c_i64, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
    "echo_op_presplit_weights": (C.c_int, [vp, c_i64, c_i64, vp]),
    "echo_op_pack_rows": (C.c_int, [vp, C.c_int, c_i64, vp, C.c_int, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "echo_op_attention_bf16": (C.c_int, [C.POINTER(EchoAttnDesc), vp]),
    "echo_op_norm": (C.c_int, [C.c_int, C.c_int, vp, c_i64, vp, c_i64, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
    "echo_op_headnorm_rope": (C.c_int, [C.c_int, vp, c_i64, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, c_i64, C.c_float,
                                        C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "echo_op_transpose_heads": (C.c_int, [C.c_int, vp, c_i64, vp, c_i64, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "echo_debug_get_kv": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "echo_debug_corrupt_tile": (C.c_int, [vp, C.c_int]),
    "echo_op_resample": (C.c_int, [vp, c_i64, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, c_i64, vp]),
    "echo_voice_capture": (C.c_int, [vp, C.POINTER(vp), vp]),
    "echo_voice_bind": (C.c_int, [vp, vp, vp]),
    "echo_voice_bytes": (c_i64, [vp]),
    "echo_voice_destroy": (None, [vp]),
    "echo_dac_decode_tail": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]),
    "echo_workspace_bytes": (c_i64, [vp]),
    "echo_fp8_calibrate": (C.c_int, [vp, C.c_int]),
    "echo_fp8_calibration": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int]),
    "echo_fp8_set_static_scales": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int]),
    "echo_reserve_workspace": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "echo_op_find_flattening_point": (C.c_int, [vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]),
    "echo_op_trailing_quiet": (C.c_int, [C.POINTER(vp), C.POINTER(c_i64), C.c_int, C.c_int, C.c_float, vp, vp]),
    "echo_op_assemble_chunks": (C.c_int, [C.POINTER(vp), C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(C.c_int32),
                                          C.c_int, vp, c_i64, vp]),
    "echo_set_profiling": (C.c_int, [vp, C.c_int]),
    "echo_get_profile": (C.c_int, [vp, C.POINTER(EchoProfile)]),
}

_lib = None


class EchoHipError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load libechohip.so and bind every declared symbol.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EchoHipError(
            f"{LIB_PATH} is missing: build it with `python echo-tts_amd/build.py` (hipcc, gfx950). "
            "There is no CPU fallback for the Echo-TTS hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.echo_abi_version() != ABI_VERSION:
        raise EchoHipError("libechohip ABI version mismatch; rebuild the library")
    _lib = lib
    return lib


def check(status: int, ctx=None) -> None:
    if status != 0:
        lib = load_library()
        msg = lib.echo_last_error(ctx)
        raise EchoHipError(msg.decode("utf-8", "replace") if msg else "libechohip call failed")
