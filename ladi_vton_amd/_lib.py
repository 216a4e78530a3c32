"""ctypes binding of libladi_native.so (C ABI declared in include/ladi_native.h).

torch is imported first on purpose: libladi_native links against libamdhip64.so.7 and must resolve to the copy torch already
loaded so that torch streams / allocations and our kernels share one HIP runtime.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_longlong, c_void_p

import torch  # noqa: F401  (must precede CDLL)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libladi_native.so")

F32, F16 = 0, 1


class NativeError(RuntimeError):
    pass


class UNetConfig(Structure):
    _fields_ = [("in_channels", c_int), ("out_channels", c_int), ("block_out_channels", c_int * 4), ("num_heads", c_int * 4),
                ("layers_per_block", c_int), ("cross_attention_dim", c_int), ("norm_num_groups", c_int), ("norm_eps", c_float)]


class VAEConfig(Structure):
    _fields_ = [("in_channels", c_int), ("out_channels", c_int), ("latent_channels", c_int), ("block_out_channels", c_int * 4),
                ("layers_per_block", c_int), ("norm_num_groups", c_int), ("scaling_factor", c_float)]


class EMASCConfig(Structure):
    _fields_ = [("n", c_int), ("in_channels", c_int * 8), ("out_channels", c_int * 8)]


class AdapterConfig(Structure):
    _fields_ = [("hidden", c_int), ("heads", c_int), ("mlp_dim", c_int), ("head_hidden", c_int), ("out_dim", c_int),
                ("layer_norm_eps", c_float)]


class TextConfig(Structure):
    _fields_ = [("vocab_size", c_int), ("hidden", c_int), ("heads", c_int), ("mlp_dim", c_int), ("layers", c_int), ("max_positions", c_int),
                ("vstar_token_id", c_int), ("layer_norm_eps", c_float)]


class VisionConfig(Structure):
    _fields_ = [("hidden", c_int), ("heads", c_int), ("mlp_dim", c_int), ("layers", c_int), ("image_size", c_int), ("patch_size", c_int),
                ("layer_norm_eps", c_float)]


class RefineConfig(Structure):
    _fields_ = [("in_channels", c_int), ("out_channels", c_int), ("base_channels", c_int), ("bn_eps", c_float)]


class TpsConfig(Structure):
    _fields_ = [("height", c_int), ("width", c_int), ("input_nc", c_int), ("n_layers", c_int), ("grid_size", c_int), ("ngf", c_int),
                ("bn_eps", c_float)]


class TryOnInputs(Structure):
    _fields_ = [("batch", c_int), ("height", c_int), ("width", c_int), ("in_dtype", c_int),
                ("image_dev", c_void_p), ("mask_image_dev", c_void_p), ("pose_map_dev", c_void_p), ("warped_cloth_dev", c_void_p),
                ("pose_channels", c_int),
                ("prompt_embeds_dev", c_void_p), ("negative_prompt_embeds_dev", c_void_p), ("L", c_int),
                ("noise_cloth_dev", c_void_p), ("noise_latents_dev", c_void_p), ("noise_masked_dev", c_void_p),
                ("num_inference_steps", c_int), ("guidance_scale", c_float), ("scheduler", c_int), ("cloth_zero_from_eval", c_int),
                ("no_pose", c_int), ("use_graph", c_int), ("alphas_cumprod_host", c_void_p)]


class IGemmDesc(Structure):
    _fields_ = [("src0", c_void_p), ("src1", c_void_p), ("C0", c_int), ("C1", c_int), ("ld0", c_int), ("ld1", c_int),
                ("Hs", c_int), ("Ws", c_int), ("Ho", c_int), ("Wo", c_int), ("P", c_int), ("ksize", c_int), ("stride", c_int),
                ("pad", c_int), ("ups", c_int),
                ("W", c_void_p), ("Q", c_int), ("K", c_int), ("ldw", c_int),
                ("bs_src0", c_longlong), ("bs_w", c_longlong), ("bs_out", c_longlong), ("bs_res", c_longlong),
                ("bias", c_void_p), ("bias_per_pixel", c_int), ("rowadd", c_void_p), ("rowadd_idx", c_void_p),
                ("rowadd_stride", c_int), ("act", c_int), ("out_scale", c_float),
                ("res0", c_void_p), ("res1", c_void_p), ("ldr0", c_int), ("ldr1", c_int), ("mask", c_void_p),
                ("out", c_void_p), ("ldo", c_int), ("out_f32", c_int), ("stats", c_void_p), ("stats_groups", c_int), ("splitk", c_int), ("tile_map", c_int),
                ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float), ("bias_mul", c_float), ("ln_scratch", c_void_p),
                ("sk_ws", c_void_p), ("sk_cnt", c_void_p), ("gn_ss", c_void_p), ("gn_hw", c_int)]


# every symbol include/ladi_native.h declares: name -> (restype, argtypes)
_P = c_void_p
SIGNATURES = {
    "ladi_last_error": (c_char_p, []),
    "ladi_version": (c_int, []),
    "ladi_device_count": (c_int, []),
    "ladi_weights_create": (_P, []),
    "ladi_weights_add": (c_int, [_P, c_char_p, _P, c_int, c_int, POINTER(c_int64)]),
    "ladi_weights_count": (c_int, [_P]),
    "ladi_weights_destroy": (None, [_P]),
    "ladi_unet_create": (_P, [POINTER(UNetConfig), _P]),
    "ladi_unet_destroy": (None, [_P]),
    "ladi_unet_set_context": (c_int, [_P, _P, c_int, c_int, _P]),
    "ladi_unet_forward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, c_int, _P]),
    "ladi_unet_time_forward": (c_int, [_P, c_int, c_int, c_int, c_int, POINTER(c_float), _P]),
    "ladi_unet_time_forward_lanes": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), _P]),
    "ladi_vae_create": (_P, [POINTER(VAEConfig), _P]),
    "ladi_vae_destroy": (None, [_P]),
    "ladi_vae_encode": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, POINTER(_P), _P]),
    "ladi_vae_decode": (c_int, [_P, _P, c_int, c_int, c_int, POINTER(_P), _P, c_int, _P]),
    "ladi_emasc_create": (_P, [POINTER(EMASCConfig), _P]),
    "ladi_emasc_destroy": (None, [_P]),
    "ladi_emasc_forward": (c_int, [_P, POINTER(_P), POINTER(c_int), POINTER(c_int), c_int, _P, c_int, c_int, POINTER(_P), _P]),
    "ladi_mask_features": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "ladi_adapter_create": (_P, [POINTER(AdapterConfig), _P]),
    "ladi_adapter_destroy": (None, [_P]),
    "ladi_adapter_forward": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "ladi_vision_encoder_create": (_P, [POINTER(VisionConfig), _P]),
    "ladi_vision_encoder_destroy": (None, [_P]),
    "ladi_vision_encoder_forward": (c_int, [_P, _P, c_int, c_int, _P, _P, _P]),
    "ladi_tps_create": (_P, [POINTER(TpsConfig), _P]),
    "ladi_tps_destroy": (None, [_P]),
    "ladi_tps_forward": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P]),
    "ladi_refine_create": (_P, [POINTER(RefineConfig), _P]),
    "ladi_refine_destroy": (None, [_P]),
    "ladi_refine_forward": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "ladi_text_encoder_create": (_P, [POINTER(TextConfig), _P]),
    "ladi_text_encoder_destroy": (None, [_P]),
    "ladi_text_encoder_forward": (c_int, [_P, _P, c_int, c_int, _P, c_int, _P, _P, _P]),
    "ladi_text_encoder_forward_dev": (c_int, [_P, _P, c_int, c_int, _P, c_int, _P, _P, _P]),
    "ladi_sched_timesteps": (c_int, [c_int, c_int, POINTER(c_int), c_int]),
    "ladi_sched_lms": (c_int, [c_int, _P, _P, _P, _P]),
    "ladi_sched_alphas_cumprod": (c_int, [POINTER(c_float)]),
    "ladi_tryon_create": (_P, [_P, _P, _P]),
    "ladi_tryon_destroy": (None, [_P]),
    "ladi_tryon_run": (c_int, [_P, POINTER(TryOnInputs), _P, _P, _P]),
    "ladi_tryon_run_u8": (c_int, [_P, POINTER(TryOnInputs), _P, _P, _P]),
    "ladi_tryon_stage_ms": (c_int, [_P, POINTER(c_float)]),
    "ladi_tryon_poll_overflow": (c_int, [_P]),
    "ladi_tryon_set_trace": (c_int, [_P, _P, _P, c_int]),
    "ladi_tryon_set_lanes": (c_int, [_P, c_int]),
    "ladi_tryon_lanes": (c_int, [_P]),
    "ladi_vae_set_range_shift": (c_int, [_P, c_int]),
    "ladi_vae_last_range_shift": (c_int, [_P]),
    "ladi_igemm_set_autotune": (None, [c_int]),
    "ladi_igemm_set_splitk_two_pass": (None, [c_int]),
    "ladi_profile_igemm_enable": (None, [c_int]),
    "ladi_profile_igemm_collect": (c_int, [POINTER(ctypes.c_double), c_int]),
    "ladi_profile_igemm_symbols": (c_int, [ctypes.c_char_p, c_int]),
    "ladi_igemm_cfg_count": (c_int, []),
    "ladi_igemm_cfg_symbol_name": (ctypes.c_char_p, [c_int]),
    "ladi_op_igemm": (c_int, [POINTER(IGemmDesc), c_int, c_int, _P]),
    "ladi_op_group_norm": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_float, c_int, _P, _P, _P, _P]),
    "ladi_op_layer_norm": (c_int, [_P, _P, _P, c_float, c_int, c_int, _P, _P]),
    "ladi_op_xattn_block": (c_int, [_P, _P, _P, c_float, _P, _P, c_int, _P, _P, c_int, c_int, _P, _P]),
    "ladi_op_ff_block": (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, _P, _P]),
    "ladi_op_attention": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_longlong,
                                  c_int, c_int, c_int, c_int, c_float, _P]),
    "ladi_op_attention_causal": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_longlong,
                                         c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "ladi_op_attention_generic": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_longlong,
                                          c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "ladi_op_attention_wide": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_longlong,
                                       c_int, c_int, c_int, c_int, c_float, _P]),
    "ladi_op_resize_bilinear_aa": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "ladi_clock_probe": (c_int, [ctypes.c_ulonglong, _P, _P]),
    "ladi_op_clip_preprocess": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), _P, _P]),
    "ladi_op_grid_sample_border": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_int, _P]),
    "ladi_op_maxpool2": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "ladi_op_upsample2x_bilinear": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "ladi_op_softmax_rows": (c_int, [_P, c_int, c_int, c_float, _P, _P]),
    "ladi_op_small_linear": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "ladi_op_nchw_to_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "ladi_op_nhwc_to_nchw": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "ladi_op_sched_run": (c_int, [c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P]),
    "ladi_op_prepare_mask": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "ladi_op_mask_down": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "ladi_op_pose_down8": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "ladi_op_posterior_sample": (c_int, [_P, c_int, _P, c_int, c_int, c_float, _P, _P]),
    "ladi_op_assemble_input": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P]),
}

_lib = None


def load(build_if_missing=True):
    """Load (building first if the .so is absent) and type every entry point. Raises if the library cannot be produced."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing:
        # build() is a no-op when the source digest stamp matches the .so; a stale library (older kernels / ABI structs than the ctypes
        # mirrors below) must never be loaded silently
        from . import build as _build
        _build.build()
    elif not os.path.exists(LIB_PATH):
        raise NativeError("libladi_native.so is not built (python -m ladi_vton_amd.build)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().ladi_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise NativeError("%s failed (rc=%d): %s" % (what, rc, last_error()))


def require_gpu():
    """The product path has no CPU fallback (include/ladi_native.h)."""
    if not torch.cuda.is_available() or load().ladi_device_count() <= 0:
        raise NativeError("ladi_vton_amd requires a ROCm GPU (gfx950); there is no CPU fallback")


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise NativeError("unsupported dtype %s (float32 / float16 only)" % t.dtype)
