"""ctypes binding of librsvld_hip.so (C ABI declared in include/rsvld_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call returns an error
code this module raises, so a silent CPU/eager path can never stand in for the kernels.
"""
import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RSVLD_LIB: developer override used by the ablation / A-B tools (another build of the same library)
LIB_PATH = os.environ.get("RSVLD_LIB") or os.path.join(_PKG_DIR, "librsvld_hip.so")

F16, BF16, F32 = 0, 1, 2   # F32: the *_f32 entry points only (fp32-operand VAE family)
SPLIT = 3                   # rsvld_conv_desc.dtype: bf16 planes + weight triples (the split-operand product path)
F16W2 = 4                   # rsvld_conv_desc.dtype: fp16 activations x fp16 weight pairs [W_lo | W_hi]
F16W1 = 5                   # rsvld_conv_desc.dtype: fp16 activations x fp16 weights, fp32 out + fp32 residual (one MFMA per product)
F16Q8 = 6                   # rsvld_conv_desc.dtype: fp16 hi x hi + the split product's two cross terms in e4m3 (rsvld_conv3x3_halo_nhwc)
ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2
# rsvld_conv_desc.tune (developer A/B overrides)
TUNE_TILE = {"256x64": 1, "128x64": 2, "128x128": 3, "64x128": 4}
TUNE_STAGES_SHIFT, TUNE_NO_KSPLIT, TUNE_REG_STAGING = 3, 1 << 6, 1 << 7
TUNE_HALO_NW4, TUNE_HALO_NW8, TUNE_NO_GEMM256 = 1 << 8, 1 << 9, 1 << 10
TUNE_GEMM_ONE_TILE = 1 << 12   # gemm256: one tile per workgroup instead of the persistent form (A/B)
TUNE_F32_SPLIT = 1 << 11   # rsvld_conv2d_nhwc_f32: split-operand precision mode (three 16-bit MFMAs per fp32 product)

ERRORS = {-1: "RSVLD_EINVAL (bad shape / pointer / combination)",
          -2: "RSVLD_EUNSUPPORTED", -3: "RSVLD_ELAUNCH (HIP launch failed)"}


class RsvldError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("rowvec", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Cin2", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32),
        ("pad_l", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("upsample", C.c_int32), ("dtype", C.c_int32), ("out_f32", C.c_int32), ("act", C.c_int32),
        ("alpha", C.c_float), ("beta", C.c_float), ("rowvec_stride", C.c_int32),
        ("plan_div", C.c_int32), ("tune", C.c_int32),
    ]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every symbol include/rsvld_hip.h declares
SIGNATURES = {
    "rsvld_version": (C.c_char_p, []),
    "rsvld_conv2d_nhwc": (_i, [C.POINTER(ConvDesc), _vp]),
    "rsvld_conv3x3_halo_supported": (_i, [C.POINTER(ConvDesc)]),
    "rsvld_conv3x3_halo_nhwc": (_i, [C.POINTER(ConvDesc), _vp, _i, _vp, _vp]),
    "rsvld_groupnorm_scale_shift_from_partials": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "rsvld_groupnorm_scale_shift": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "rsvld_groupnorm_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "rsvld_groupnorm_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp]),
    "rsvld_groupnorm_stats": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rsvld_groupnorm_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "rsvld_layernorm": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _i, _vp]),
    "rsvld_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                             _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _i, _vp, _vp]),
    "rsvld_attention_tuned": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                             _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _i, _vp, _vp, _i]),
    "rsvld_attention_ws_bytes": (_i64, [_i, _i, _i, _i, _i, _i]),
    "rsvld_linear_small_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsvld_sinusoidal_embedding": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "rsvld_nchw_f32_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "rsvld_nhwc_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsvld_axpby": (_i, [_vp, _vp, _vp, _i64, _f, _f, _i, _vp]),
    "rsvld_geglu": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "rsvld_ddpm_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _vp]),
    "rsvld_denoiser_out": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "rsvld_lerp_f32": (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
    "rsvld_axpy_f32": (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
    "rsvld_euler_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _vp]),
    "rsvld_tile_blend_accumulate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsvld_tile_blend_finish": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "rsvld_absdiff_ws_bytes": (_i64, [_i, _i64]),
    "rsvld_absdiff_sums": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp]),
    "rsvld_gaussian_sample": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "rsvld_wavelet_blur": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsvld_add_f32": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "rsvld_adain": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "rsvld_concat_c": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "rsvld_conv2d_nhwc_f32": (_i, [C.POINTER(ConvDesc), _vp]),
    "rsvld_groupnorm_f32_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "rsvld_groupnorm_stats_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rsvld_groupnorm_apply_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "rsvld_gemv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rsvld_gemv_fused": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _i, _i, _vp]),
    "rsvld_llama_decode_attention_ws_bytes": (_i64, [_i, _i, _i]),
    "rsvld_llama_decode_attention": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "rsvld_layernorm_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    "rsvld_concat_c_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "rsvld_axpby_f32": (_i, [_vp, _vp, _vp, _i64, _f, _f, _vp]),
    "rsvld_absdiff_sums_f32": (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    "rsvld_attention_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                 _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp]),
    "rsvld_attention_f32_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                 _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp]),
    "rsvld_nchw_f32_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "rsvld_split_planes": (_i, [_vp, _vp, _i64, _i, _vp]),
    "rsvld_merge_planes": (_i, [_vp, _vp, _i64, _i, _vp]),
    "rsvld_planes_to_f16": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _vp]),
    "rsvld_f16_to_planes": (_i, [_vp, _i64, _vp, _i64, _i, _vp]),
    "rsvld_split_pack_weights": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "rsvld_pack_weight_pairs": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "rsvld_pack_weight_hq8": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "rsvld_split_hq8": (_i, [_vp, _vp, _i64, _i, _vp]),
    "rsvld_planes_transpose_triple": (_i, [_vp, _vp, _i64, _i64, _i, _i64, _vp]),
    "rsvld_planes_to_triple": (_i, [_vp, _vp, _i64, _i64, _i, _i64, _vp]),
    "rsvld_softmax_rows_split": (_i, [_vp, _vp, _i64, _i, _i, _i64, _f, _vp]),
    "rsvld_groupnorm_scale_shift_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "rsvld_groupnorm_stats_f32_fast": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rsvld_groupnorm_scale_shift_from_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "rsvld_groupnorm_apply_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsvld_layernorm_split": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _i, _vp]),
    "rsvld_attention_split_d512_shared": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _vp]),
    "rsvld_attention_split_d64": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                       _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RsvldError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  If this library were loaded before torch, the
    # loader would bind it to the system copy under /opt/rocm and the process would hold two HIP runtimes -- kernels launched
    # through one on streams and pointers owned by the other fail with hipErrorInvalid* (seen as RSVLD_ELAUNCH).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RsvldError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {os.path.join(_PKG_DIR, 'csrc')}`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RsvldError(f"{what} failed: {ERRORS.get(rc, rc)}")
