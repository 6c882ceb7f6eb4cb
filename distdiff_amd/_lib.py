"""ctypes binding of libdistdiff_hip.so (the C-ABI drop-in boundary, include/*.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the
ops raises. PyTorch is used only for device memory / streams (tensor.data_ptr()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DD_LIB") or os.path.join(_HERE, "libdistdiff_hip.so")   # DD_LIB: A/B builds for benchmarking

ABI_VERSION = 6      # DD_ABI_VERSION of include/distdiff_hip.h this package's ctypes mirrors were written against

_lib = None


class DistDiffLibraryError(RuntimeError):
    pass


def lib():
    """Returns the loaded shared library; raises loudly when it is absent (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DistDiffLibraryError(
                "libdistdiff_hip.so not built: run `python -m distdiff_amd.build` (needs hipcc, gfx950). "
                "There is no CPU fallback for the product path.")
        try:
            # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's): load torch FIRST so that this library binds to
            # the HIP runtime instance torch uses (one runtime per process; the other order leaves a runtime that sees no device)
            import torch  # noqa: F401
            _lib = C.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise DistDiffLibraryError("cannot load %s: %s" % (LIB_PATH, e))
        # one version for both header surfaces (struct layouts and argument lists): a stale build must not be called with new mirrors
        got = _lib.dd_abi_version() if hasattr(_lib, "dd_abi_version") else 5
        if got != ABI_VERSION:
            _lib = None
            raise DistDiffLibraryError("%s was built as ABI version %d, this package speaks %d: rebuild with `python -m distdiff_amd.build`"
                                       % (LIB_PATH, got, ABI_VERSION))
        _declare(_lib)
    return _lib


vp = C.c_void_p


class ConvGemmParams(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("taptab", vp), ("y", vp), ("bias", vp), ("bias_sel", vp), ("res", vp),
                ("mask", vp), ("raw", vp), ("partial", vp), ("stats", vp), ("ln_stats", vp), ("ln_c1", vp), ("gn_coef", vp), ("rowpart", vp),
                ("x_ld", C.c_int), ("y_ld", C.c_int), ("res_ld", C.c_int), ("mask_ld", C.c_int), ("raw_ld", C.c_int),
                ("bias_stride", C.c_int), ("stats_ld", C.c_int), ("rowpart_ld", C.c_int), ("gn_silu", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("stride", C.c_int),
                ("shift", C.c_int), ("parity", C.c_int), ("cin", C.c_int), ("ntaps", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ksplit", C.c_int), ("flags", C.c_int),
                ("alpha", C.c_float), ("force_small", C.c_int), ("wgroup_rows", C.c_int), ("wgroup_elems", C.c_longlong)]


class GroupNormParams(C.Structure):
    _fields_ = [("x", vp), ("x_ld", C.c_int), ("y", vp), ("y_ld", C.c_int), ("gamma", vp), ("beta", vp),
                ("stats", vp), ("scratch", vp), ("coef", vp), ("chan_part", vp), ("part_ld", C.c_int),
                ("B", C.c_int), ("HW", C.c_int), ("C", C.c_int), ("G", C.c_int), ("eps", C.c_float), ("silu", C.c_int),
                ("dy", vp), ("dy_ld", C.c_int), ("dx", vp), ("dx_ld", C.c_int), ("accumulate", C.c_int)]


class LayerNormParams(C.Structure):
    _fields_ = [("x", vp), ("x_ld", C.c_int), ("y", vp), ("y_ld", C.c_int), ("gamma", vp), ("beta", vp), ("stats", vp),
                ("M", C.c_int), ("C", C.c_int), ("eps", C.c_float), ("rowpart", vp), ("rowpart_ld", C.c_int), ("spans", C.c_int),
                ("dy", vp), ("dy_ld", C.c_int), ("dx", vp), ("dx_ld", C.c_int), ("accumulate", C.c_int)]


class AttnParams(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("o", vp), ("lse", vp),
                ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int), ("D", C.c_int), ("scale", C.c_float),
                ("d_o", vp), ("lddo", C.c_int), ("dq", vp), ("dk", vp), ("dv", vp),
                ("lddq", C.c_int), ("lddk", C.c_int), ("lddv", C.c_int), ("delta", vp), ("q_prescaled", C.c_int), ("causal", C.c_int), ("pv_fp8", C.c_int),
                ("no_shortk", C.c_int)]


class ConvF32Params(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("taptab", vp), ("y", vp), ("bias", vp), ("res", vp), ("mask", vp),
                ("x_ld", C.c_int), ("y_ld", C.c_int), ("res_ld", C.c_int), ("mask_ld", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("stride", C.c_int),
                ("shift", C.c_int), ("parity", C.c_int), ("cin", C.c_int), ("ntaps", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("groups", C.c_int), ("cpg_in", C.c_int), ("cpg_out", C.c_int), ("flags", C.c_int)]


CF_BIAS, CF_RES, CF_RELU, CF_GEGLU, CF_OUT_F32, CF_MASK, CF_RES_F32, CF_GEGLU_RAW, CF_STATS = 1, 2, 4, 8, 16, 32, 64, 128, 256
CF_LNFOLD, CF_ROWSTATS, CF_GNFOLD = 1024, 2048, 4096

# every symbol declared in include/distdiff_hip_ops.h and include/distdiff_hip.h (checked by tests/test_abi.py)
OPS_SYMBOLS = [
    "dd_op_conv_gemm", "dd_op_conv_gemm_check", "dd_op_groupnorm_fwd", "dd_op_groupnorm_bwd", "dd_op_groupnorm_scratch_bytes",
    "dd_op_layernorm_fwd", "dd_op_layernorm_bwd", "dd_op_attention_fwd", "dd_op_attention_bwd",
    "dd_op_attention_gemm_workspace", "dd_op_attention_gemm_fwd", "dd_op_attention_gemm_bwd",
    "dd_pack_conv_weight", "dd_op_conv_f32", "dd_pack_conv_weight_f32", "dd_op_nchw_f32_to_nhwc_bf16", "dd_op_nhwc_to_nchw_f32", "dd_op_cfg_ddim",
    "dd_op_cfg_ddim_bwd", "dd_op_sumpool2x2", "dd_op_geglu_bwd", "dd_op_maxpool3x3s2", "dd_op_maxpool3x3s2_bwd",
    "dd_op_bicubic", "dd_op_bicubic_bwd", "dd_op_gap", "dd_op_energy", "dd_op_transform_update", "dd_op_affine",
    "dd_debug_tensor", "dd_debug_num_tensors", "dd_debug_set_image", "dd_debug_set_images",
]
ENGINE_SYMBOLS = [
    "dd_abi_version", "dd_create", "dd_destroy", "dd_last_error", "dd_load_tensor", "dd_finalize_weights", "dd_set_prototypes",
    "dd_set_schedule", "dd_add_noise", "dd_denoise_step", "dd_transform_guidance", "dd_direct_guidance", "dd_decode",
    "dd_expand", "dd_image_to_u8", "dd_guide_encode", "dd_guide_encode_pooled", "dd_unet_forward", "dd_unet_vjp", "dd_decode_vjp", "dd_guide_vjp",
    "dd_set_prompt", "dd_set_added_cond", "dd_vae_encode", "dd_text_encode", "dd_text_encode_tower", "dd_set_sample_weights", "dd_get_image_scores", "dd_declare_tensor", "dd_packed_bytes", "dd_export_packed", "dd_import_packed", "dd_profile_enable", "dd_profile_read", "dd_workspace_bytes", "dd_flops_last",
]


def _declare(l):
    i, f, sz = C.c_int, C.c_float, C.c_size_t
    l.dd_op_conv_gemm.argtypes = [C.POINTER(ConvGemmParams), sz, vp]
    l.dd_op_conv_gemm_check.argtypes = [C.POINTER(ConvGemmParams), vp]
    l.dd_op_groupnorm_fwd.argtypes = [C.POINTER(GroupNormParams), vp]
    l.dd_op_groupnorm_bwd.argtypes = [C.POINTER(GroupNormParams), vp]
    l.dd_op_groupnorm_scratch_bytes.argtypes = [i, i]
    l.dd_op_groupnorm_scratch_bytes.restype = sz
    l.dd_op_layernorm_fwd.argtypes = [C.POINTER(LayerNormParams), vp]
    l.dd_op_layernorm_bwd.argtypes = [C.POINTER(LayerNormParams), vp]
    l.dd_op_attention_fwd.argtypes = [C.POINTER(AttnParams), vp]
    l.dd_op_attention_bwd.argtypes = [C.POINTER(AttnParams), vp]
    l.dd_pack_conv_weight.argtypes = [vp, i, i, i, i, i, i, i, vp, vp, vp]
    l.dd_op_conv_f32.argtypes = [C.POINTER(ConvF32Params), vp]
    l.dd_pack_conv_weight_f32.argtypes = [vp, i, i, i, i, i, i, i, vp, vp, vp]
    l.dd_op_nchw_f32_to_nhwc_bf16.argtypes = [vp, vp, i, i, i, i, i, i, i, f, vp]
    l.dd_op_nhwc_to_nchw_f32.argtypes = [vp, i, vp, i, i, i, i, i, f, f, i, f, f, vp]
    l.dd_op_cfg_ddim.argtypes = [vp, i, vp, vp, vp, i, i, i, vp, vp]
    l.dd_op_cfg_ddim_bwd.argtypes = [vp, vp, vp, i, vp, i, i, i, vp, vp]
    l.dd_op_sumpool2x2.argtypes = [vp, i, vp, i, i, i, i, i, i, vp]
    l.dd_op_geglu_bwd.argtypes = [vp, i, vp, i, vp, i, i, i, vp]
    l.dd_op_maxpool3x3s2.argtypes = [vp, vp, i, i, i, i, vp]
    l.dd_op_maxpool3x3s2_bwd.argtypes = [vp, vp, vp, i, i, i, i, vp]
    l.dd_op_bicubic.argtypes = [vp, i, vp, i, i, i, i, i, i, i, i, vp]
    l.dd_op_bicubic_bwd.argtypes = [vp, i, vp, i, i, i, i, i, i, i, vp]
    l.dd_op_gap.argtypes = [vp, i, vp, i, i, i, vp]
    l.dd_op_energy.argtypes = [vp, vp, vp, vp, i, i, i, f, f, i, i, i, f, vp, vp, vp]
    l.dd_op_transform_update.argtypes = [vp, vp, vp, vp, vp, i, i, f, f, vp]
    l.dd_op_affine.argtypes = [vp, vp, vp, vp, i, i, vp]
    if hasattr(l, "dd_create"):
        from . import engine as _engine  # noqa: F401  (declares the engine prototypes)
        _engine._declare(l)


def check(err, what=""):
    if err != 0:
        raise RuntimeError("HIP error %d in %s" % (err, what))
