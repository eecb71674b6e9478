"""ctypes binding of libaldm_hip.so (the C-ABI declared in include/aldm_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALDM_LIB") or os.path.join(_HERE, "libaldm_hip.so")   # ALDM_LIB: diagnostic builds only

ACT_NONE, ACT_SILU, ACT_LRELU, ACT_TANH, ACT_GELU = 0, 1, 2, 3, 4
OUT_BF16, OUT_F32 = 0, 1
TILE_AUTO, TILE_128x128, TILE_64x64, TILE_128x64, TILE_64x128 = 0, 1, 2, 3, 4


class IgemmArgs(C.Structure):
    """Mirror of aldm_igemm_t (include/aldm_hip.h) -- field order and types must match exactly."""
    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p),
        ("B", C.c_int), ("IH", C.c_int), ("IW", C.c_int), ("Cin", C.c_int), ("Cin2", C.c_int),
        ("UH", C.c_int), ("UW", C.c_int),
        ("in_dilate", C.c_int),
        ("w", C.c_void_p),
        ("Kpad", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("stride_h", C.c_int), ("stride_w", C.c_int),
        ("pad_h", C.c_int), ("pad_w", C.c_int), ("dil_h", C.c_int), ("dil_w", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int),
        ("in_act", C.c_int), ("in_slope", C.c_float),
        ("ln_s", C.c_void_p), ("ln_sa", C.c_void_p), ("ln_ca", C.c_void_p), ("ln_eps", C.c_float),
        ("lora_a", C.c_void_p), ("lora_b", C.c_void_p),
        ("Rp", C.c_int),
        ("lora_t_out", C.c_void_p),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p),
        ("rowbias_ld", C.c_int),
        ("geglu", C.c_int),
        ("out_act", C.c_int), ("out_slope", C.c_float),
        ("res", C.c_void_p), ("res2", C.c_void_p),
        ("alpha", C.c_float),
        ("post_act", C.c_int), ("post_slope", C.c_float), ("out2", C.c_void_p),
        ("out", C.c_void_p), ("out_dtype", C.c_int), ("out_ld", C.c_int),
        ("out_batch_stride", C.c_longlong),
        ("out_pix_stride", C.c_int), ("out_pix_offset", C.c_int),
        ("vt", C.c_void_p), ("vt_col0", C.c_int), ("vt_ld", C.c_int), ("vt_batch_stride", C.c_longlong),
        ("splits", C.c_int), ("workspace", C.c_void_p),
        ("tile", C.c_int),
        ("ring", C.c_int),
        ("defer_reduce", C.c_int),
        ("rowstat_out", C.c_void_p), ("ln_parts", C.c_void_p), ("ln_nparts", C.c_int),
        ("qstat_out", C.c_void_p),
        ("x3", C.c_void_p), ("x4", C.c_void_p), ("Cin3", C.c_int), ("Cin4", C.c_int),
        ("vt_dual", C.c_int),
        ("gnin_gamma", C.c_void_p), ("gnin_beta", C.c_void_p), ("gnin_q1", C.c_void_p), ("gnin_q2", C.c_void_p),
        ("gnin_bm1", C.c_int), ("gnin_tpi1", C.c_int), ("gnin_bm2", C.c_int), ("gnin_tpi2", C.c_int),
        ("gnin_groups", C.c_int), ("gnin_act", C.c_int), ("gnin_eps", C.c_float),
        ("xcd_map", C.c_int),
    ]


class PgemmArgs(C.Structure):
    """Mirror of aldm_pgemm_t (include/aldm_hip.h) -- field order and types must match exactly."""
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("bias", C.c_void_p),
        ("ln_s", C.c_void_p), ("ln_sa", C.c_void_p), ("ln_ca", C.c_void_p), ("ln_eps", C.c_float),
        ("ln_parts", C.c_void_p), ("ln_nparts", C.c_int),
        ("lora_a", C.c_void_p), ("lora_b", C.c_void_p),
        ("Rp", C.c_int), ("ranks_used", C.c_int),
        ("geglu", C.c_int),
        ("res", C.c_void_p),
        ("out", C.c_void_p), ("out_ld", C.c_int),
        ("vt", C.c_void_p), ("vt_col0", C.c_int), ("vt_ld", C.c_int), ("vt_batch_stride", C.c_longlong), ("OHW", C.c_int),
        ("rowstat_out", C.c_void_p),
        ("mi", C.c_int), ("nt", C.c_int), ("tiles_per_range", C.c_int),
        ("max_ranges", C.c_int),
        ("waves", C.c_int),
        ("lora_t_out", C.c_void_p),
        ("vt_dual", C.c_int),
    ]


# name -> (restype, argtypes): every symbol include/aldm_hip.h declares
PROTOTYPES = {
    "aldm_version": (C.c_char_p, []),
    "aldm_last_error": (C.c_char_p, []),
    "aldm_igemm": (C.c_int, [C.POINTER(IgemmArgs), C.c_void_p]),
    "aldm_igemm_workspace_bytes": (C.c_size_t, [C.POINTER(IgemmArgs)]),
    "aldm_igemm_effective_splits": (C.c_int, [C.POINTER(IgemmArgs)]),
    "aldm_pgemm_supported": (C.c_int, [C.c_int]),
    "aldm_pgemm_plan": (C.c_int, [C.POINTER(PgemmArgs)]),
    "aldm_pgemm": (C.c_int, [C.POINTER(PgemmArgs), C.c_void_p]),
    "aldm_groupnorm_partials": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_groupnorm_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "aldm_gn_silu_conv3x3_small": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_layernorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                 C.c_void_p]),
    "aldm_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_attention_prescaled": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_attn_block64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "aldm_attn_block64_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "aldm_attn_block256": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "aldm_hifigan_respair_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "aldm_hifigan_respair": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "aldm_conv1d_to1": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_attention_wide": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_attention_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_attention_varlen": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p]),
    "aldm_embed_layernorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_attention_lse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "aldm_attention_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_groupnorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "aldm_groupnorm_bwd_partials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "aldm_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "aldm_geglu_fwd": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_geglu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_add_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "aldm_upsample_nearest_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p]),
    "aldm_tn_small": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_tn_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aldm_lora_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_transpose_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_mse_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aldm_softmax_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "aldm_timestep_embedding": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_silu": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "aldm_nchw_f32_to_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_nhwc_to_nchw_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_longlong, C.c_float, C.c_void_p, C.c_void_p]),
    "aldm_cfg_ddim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_float,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aldm_ddim_step_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aldm_add_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
    "aldm_add_noise_t": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
    "aldm_gaussian_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
    "aldm_sleep_us": (C.c_int, [C.c_int, C.c_void_p]),
    "aldm_gather_row": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "aldm_advance_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_log_mel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "aldm_adamw_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p]),
}

_lib = None


class AldmError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AldmError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C audioldm_with_lora_amd/csrc`.  There is no CPU fallback on the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().aldm_last_error().decode()
        raise AldmError(f"{what} failed (rc={rc}): {msg}")
