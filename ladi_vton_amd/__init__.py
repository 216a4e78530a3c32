"""ladi_vton_amd — MI355X-native (gfx950) implementation of the LaDI-VTON denoising hot path.

Drop-in module shims + pipeline over libladi_native.so (hand-written HIP, see csrc/ and include/ladi_native.h).
There is no CPU fallback: constructing any Native* module without a ROCm GPU raises NativeError.
"""
from . import configs  # noqa: F401
from . import dataset  # noqa: F401  (host-side VITON-HD / DressCode readers)
from ._lib import NativeError  # noqa: F401
from .modules import (NativeEMASC, NativeInversionAdapter, NativeUNet, NativeVAE, mask_features)  # noqa: F401
from .pipeline import StableDiffusionTryOnePipeline  # noqa: F401
from .schedulers import DDIMScheduler, LMSDiscreteScheduler, PNDMScheduler  # noqa: F401
from .text import NativeCLIPTextEncoder, encode_text_word_embedding  # noqa: F401
from .vision import NativeCLIPVisionEncoder  # noqa: F401
from .warp import NativeRefinementUNet, NativeTPS, clip_preprocess, grid_sample_border, resize_antialias, warp_cloth  # noqa: F401


def build_random_init_pipeline(size="full", scheduler="ddim", with_emasc=True):
    """Random-init (deterministic synthetic checkpoint) models of the released architecture — no weights are reachable
    offline. Returns (pipeline, adapter_or_None, cfgs)."""
    C = configs
    ucfg, vcfg = (C.UNET_FULL, C.VAE_FULL) if size == "full" else (C.UNET_TINY, C.VAE_TINY)
    ecfg = C.emasc_for_vae(vcfg)
    unet = NativeUNet(ucfg, C.synth_state_dict(C.unet_shapes(ucfg), "unet."))
    vae = NativeVAE(vcfg, C.synth_state_dict(C.vae_shapes(vcfg), "vae."))
    emasc = NativeEMASC(ecfg, C.synth_state_dict(C.emasc_shapes(ecfg), "emasc.")) if with_emasc else None
    sch = DDIMScheduler() if scheduler in ("ddim", 0) else LMSDiscreteScheduler() if scheduler in ("lms", 2) else PNDMScheduler()
    pipe = StableDiffusionTryOnePipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, scheduler=sch, emasc=emasc,
                                         emasc_int_layers=[1, 2, 3, 4, 5] if with_emasc else None)
    return pipe, dict(unet=ucfg, vae=vcfg, emasc=ecfg)
