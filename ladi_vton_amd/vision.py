"""CLIP ViT-H/14 vision encoder over libladi_native: drop-in for the `vision_encoder` (transformers CLIPVisionModelWithProjection) the
reference calls at src/inference.py:269-273:

    vision_encoder = NativeCLIPVisionEncoder(configs.VISION_FULL, vision_encoder.state_dict())        # once
    clip_cloth_features = vision_encoder(processed_images.pixel_values.to(device, dtype=weight_dtype)).last_hidden_state

`pixel_values` is the CLIPProcessor output ([B, 3, 224, 224], CLIP-normalised); `last_hidden_state` is the encoder output without
post_layernorm (what the inversion adapter consumes), `pooler_output` = post_layernorm(last_hidden_state[:, 0]).  `image_embeds` (the
visual projection) is not on the reference's path and is not produced.  No CPU fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import NativeError, VisionConfig, check, dtype_code, ptr, stream_ptr
from .modules import _Shim, _Weights


class VisionEncoderOutput:
    def __init__(self, last_hidden_state, pooler_output):
        self.last_hidden_state, self.pooler_output = last_hidden_state, pooler_output
        self.image_embeds, self.hidden_states, self.attentions = None, None, None

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]


class NativeCLIPVisionEncoder(_Shim):
    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = VisionConfig()
        c.hidden, c.heads, c.mlp_dim, c.layers = cfg["hidden"], cfg["heads"], cfg["mlp_dim"], cfg["layers"]
        c.image_size, c.patch_size, c.layer_norm_eps = cfg["image_size"], cfg["patch_size"], cfg["layer_norm_eps"]
        with _Weights(state_dict) as w:
            self.h = self.lib.ladi_vision_encoder_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_vision_encoder_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)
        self.dtype = torch.float16
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_vision_encoder_destroy(self.h)
            self.h = None

    def __call__(self, pixel_values):
        S = self.cfg["image_size"]
        if pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != (3, S, S):
            raise ValueError("pixel_values must be [B, 3, %d, %d]" % (S, S))
        x = pixel_values.to(self.device)
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        x = x.contiguous()
        B, H = x.shape[0], self.cfg["hidden"]
        T = 1 + (S // self.cfg["patch_size"]) ** 2
        hidden = torch.empty((B, T, H), dtype=torch.float16, device=self.device)
        pooled = torch.empty((B, H), dtype=torch.float16, device=self.device)
        check(self.lib.ladi_vision_encoder_forward(self.h, ptr(x), dtype_code(x), B, ptr(hidden), ptr(pooled), stream_ptr()),
              "ladi_vision_encoder_forward")
        return VisionEncoderOutput(hidden, pooled)
