"""Drop-in module shims over libladi_native: the nn.Module duck types StableDiffusionTryOnePipeline is built from
(reference: src/vto_pipelines/tryon_pipe.py:56-68,129-137; src/inference.py:212-220).

Each class exposes exactly the attributes / call conventions the reference pipeline touches and forwards the arithmetic to the
C ABI.  Tensors crossing between our own modules (the encoder features / EMASC skips) are NCHW-shaped views of NHWC fp16
buffers (channels_last strides), so no layout conversion happens between them.
"""
import ctypes
from ctypes import c_float, c_int, c_int64, c_void_p
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import (AdapterConfig, EMASCConfig, NativeError, UNetConfig, VAEConfig, check, dtype_code, ptr, stream_ptr)


# ---------------------------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------------------------
class _Weights:
    """host staging of a diffusers-format state_dict (ladi_weights_*)"""

    def __init__(self, state_dict):
        self.lib = _lib.load()
        self.h = self.lib.ladi_weights_create()
        if not self.h:
            raise NativeError("ladi_weights_create failed")
        items = state_dict.items() if hasattr(state_dict, "items") else state_dict   # dict or iterator of (key, tensor) pairs
        for k, v in items:
            t = v.detach().to("cpu")
            if t.dtype not in (torch.float32, torch.float16):
                t = t.float()
            t = t.contiguous()
            shape = (c_int64 * max(t.dim(), 1))(*t.shape)
            check(self.lib.ladi_weights_add(self.h, k.encode(), c_void_p(t.data_ptr()), dtype_code(t), t.dim(), shape), "ladi_weights_add(%s)" % k)

    def close(self):
        if self.h:
            self.lib.ladi_weights_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _is_nhwc_dense(t):
    return t.dim() == 4 and t.dtype == torch.float16 and t.permute(0, 2, 3, 1).is_contiguous()


def _nhwc_buffer(t):
    """return an NHWC-dense fp16 [B,H,W,C] tensor holding `t` ([B,C,H,W] any strides / fp32|fp16) using the native converter"""
    if _is_nhwc_dense(t):
        return t.permute(0, 2, 3, 1)
    lib = _lib.load()
    src = t.contiguous()
    B, C, H, W = src.shape
    out = torch.empty((B, H, W, C), dtype=torch.float16, device=src.device)
    check(lib.ladi_op_nchw_to_nhwc(ptr(src), dtype_code(src), B, C, H, W, ptr(out), C, stream_ptr()), "nchw_to_nhwc")
    return out


class _Shim:
    """the nn.Module housekeeping the unmodified src/inference.py:192-209 performs on every module (`.to(device, dtype=...)`, `.eval()`):
    no-ops here, the weights already live on the device in the native layout"""

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def half(self):
        return self

    def float(self):
        return self

    def requires_grad_(self, requires_grad=False):
        return self

    def modules(self):
        return iter(())

    def parameters(self):
        return iter(())


class _Base(_Shim):
    dtype = torch.float16

    @property
    def device(self):
        return torch.device("cuda", torch.cuda.current_device())


# ---------------------------------------------------------------------------------------------------------------
# UNet (tryon_pipe.py:732: unet(x, t, encoder_hidden_states=...).sample)
# ---------------------------------------------------------------------------------------------------------------
class NativeUNet(_Base):
    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = UNetConfig()
        c.in_channels, c.out_channels = cfg["in_channels"], cfg["out_channels"]
        c.block_out_channels = (c_int * 4)(*cfg["block_out_channels"])
        c.num_heads = (c_int * 4)(*cfg["num_heads"])
        c.layers_per_block, c.cross_attention_dim = cfg["layers_per_block"], cfg["cross_attention_dim"]
        c.norm_num_groups, c.norm_eps = cfg["norm_num_groups"], cfg["norm_eps"]
        with _Weights(state_dict) as w:
            self.h = self.lib.ladi_unet_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_unet_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)
        self.config = SimpleNamespace(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"], sample_size=64,
                                      _diffusers_version="0.14.0")
        self._ctx_key = None
        self._ctx_keepalive = None

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_unet_destroy(self.h)
            self.h = None

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None  # attention is always fused (flash-style) in the native kernels

    def set_context(self, ehs):
        # the cross-attention K/V projections are step-invariant: skip the upload when the SAME tensor object (kept alive here, so its
        # id / storage cannot be recycled by the caching allocator) is passed again unmodified
        key = (id(ehs), ehs._version)
        if self._ctx_key == key and self._ctx_keepalive is not None and self._ctx_keepalive[0] is ehs:
            return
        e = ehs.to(dtype=torch.float16).contiguous()
        check(self.lib.ladi_unet_set_context(self.h, ptr(e), e.shape[0], e.shape[1], stream_ptr()), "ladi_unet_set_context")
        self._ctx_key = key
        self._ctx_keepalive = (ehs, e)

    def __call__(self, sample, timestep, encoder_hidden_states=None, **kw):
        if encoder_hidden_states is None:
            raise ValueError("encoder_hidden_states is required")
        self.set_context(encoder_hidden_states)
        x = sample.contiguous()
        n, C, h, w = x.shape
        if C != self.cfg["in_channels"]:
            raise ValueError("expected %d input channels, got %d" % (self.cfg["in_channels"], C))
        out = torch.empty((n, self.cfg["out_channels"], h, w), dtype=x.dtype, device=x.device)
        check(self.lib.ladi_unet_forward(self.h, ptr(x), dtype_code(x), n, h, w, float(timestep), ptr(out), dtype_code(out), stream_ptr()),
              "ladi_unet_forward")
        return SimpleNamespace(sample=out)

    def time_forward(self, n, h, w, iters):
        ms = c_float(0)
        check(self.lib.ladi_unet_time_forward(self.h, n, h, w, iters, ctypes.byref(ms), stream_ptr()), "ladi_unet_time_forward")
        return ms.value

    def time_forward_lanes(self, n, h, w, iters, lanes=0, use_graph=True):
        """average ms of one forward run the way the denoising loop runs it: `lanes` sample groups on as many streams, one hipGraph"""
        ms = c_float(0)
        check(self.lib.ladi_unet_time_forward_lanes(self.h, n, h, w, iters, lanes, 1 if use_graph else 0, ctypes.byref(ms), stream_ptr()),
              "ladi_unet_time_forward_lanes")
        return ms.value


# ---------------------------------------------------------------------------------------------------------------
# VAE (AutoencoderKL.encode / .decode with EMASC wiring)
# ---------------------------------------------------------------------------------------------------------------
class DiagonalGaussianDistribution:
    """src/models/vae.py:329-348 over moments produced natively"""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            dev = generator.device if generator is not None else self.mean.device
            noise = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise.to(self.mean.dtype)

    def mode(self):
        return self.mean


class NativeVAE(_Base):
    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = VAEConfig()
        c.in_channels, c.out_channels, c.latent_channels = cfg["in_channels"], cfg["out_channels"], cfg["latent_channels"]
        c.block_out_channels = (c_int * 4)(*cfg["block_out_channels"])
        c.layers_per_block, c.norm_num_groups, c.scaling_factor = cfg["layers_per_block"], cfg["norm_num_groups"], cfg["scaling_factor"]
        with _Weights(state_dict) as w:
            self.h = self.lib.ladi_vae_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_vae_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)
        self.config = SimpleNamespace(scaling_factor=cfg["scaling_factor"], latent_channels=cfg["latent_channels"],
                                      block_out_channels=tuple(cfg["block_out_channels"]))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_vae_destroy(self.h)
            self.h = None

    @property
    def range_shift(self):
        """fp16-range guard of the decoder: "auto" (default) decodes with the residual stream at its true scale and, when a GroupNorm
        reports non-finite statistics (an fp16 overflow upstream), again with the stream stored x 2^-4, then 2^-8; an int k fixes the stream
        scale at 2^-k.  The reference decodes in whatever dtype the caller chose (src/models/AutoencoderKL.py:159-188); SD VAE checkpoints
        are known to leave the fp16 range there.  `last_range_shift` tells what the last decode used."""
        return getattr(self, "_range_shift", "auto")

    @range_shift.setter
    def range_shift(self, k):
        kk = -1 if k in ("auto", None) else int(k)
        check(self.lib.ladi_vae_set_range_shift(self.h, kk), "ladi_vae_set_range_shift")
        self._range_shift = "auto" if kk < 0 else kk

    @property
    def last_range_shift(self):
        return self.lib.ladi_vae_last_range_shift(self.h)

    def encode(self, x, return_dict=True):
        x = x.contiguous()
        B, _, H, W = x.shape
        boc = self.cfg["block_out_channels"]
        dev = x.device
        moments = torch.empty((B, 2 * self.cfg["latent_channels"], H // 8, W // 8), dtype=torch.float32, device=dev)
        shapes = [(B, H, W, boc[0]), None, (B, H // 2, W // 2, boc[0]), (B, H // 4, W // 4, boc[1]), (B, H // 8, W // 8, boc[2])]
        bufs = [torch.empty(s, dtype=torch.float16, device=dev) if s else None for s in shapes]
        arr = (c_void_p * 5)(*[b.data_ptr() if b is not None else None for b in bufs])
        check(self.lib.ladi_vae_encode(self.h, ptr(x), dtype_code(x), B, H, W, ptr(moments), arr, stream_ptr()), "ladi_vae_encode")
        views = [b.permute(0, 3, 1, 2) if b is not None else None for b in bufs]
        views[1] = views[0]  # idx1 and idx2 are the same tensor in the reference (vae.py:104-109)
        feats = [x] + views
        post = DiagonalGaussianDistribution(moments.to(x.dtype) if x.dtype == torch.float16 else moments)
        if not return_dict:
            return (post,)
        return SimpleNamespace(latent_dist=post), feats

    def decode(self, z, intermediate_features=None, int_layers=None, return_dict=True):
        B, _, h, w = z.shape
        zf = z.float().contiguous()
        arr = None
        keep = []
        if intermediate_features:
            # Decoder.forward (vae.py:188-205) zips the REVERSED list with the four up blocks (a list shorter than four silently SKIPS the
            # remaining up blocks, which the released channel widths then reject at conv_norm_out) and adds layer 1 after conv_norm_out:
            # the only selections whose shapes line up are [1, 2, 3, 4, 5] and [2, 3, 4, 5]
            layers = list(int_layers or [])
            if layers not in ([1, 2, 3, 4, 5], [2, 3, 4, 5]) or len(intermediate_features) != len(layers):
                raise ValueError("int_layers must be [1, 2, 3, 4, 5] or [2, 3, 4, 5] with one feature per layer (got %r with %d features): "
                                 "the reference's Decoder.forward cannot run any other selection on the released architecture"
                                 % (int_layers, len(intermediate_features)))
            keep = [_nhwc_buffer(f) for f in intermediate_features]
            slots = [None] * 5
            for layer, k in zip(layers, keep):
                slots[layer - 1] = k.data_ptr()
            arr = (c_void_p * 5)(*slots)
            intermediate_features.reverse()  # the reference reverses the caller's list in place (vae.py:190)
        out = torch.empty((B, self.cfg["out_channels"], 8 * h, 8 * w), dtype=z.dtype if z.dtype in (torch.float16, torch.float32) else torch.float32,
                          device=z.device)
        check(self.lib.ladi_vae_decode(self.h, ptr(zf), B, h, w, arr, ptr(out), dtype_code(out), stream_ptr()), "ladi_vae_decode")
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


# ---------------------------------------------------------------------------------------------------------------
# EMASC + mask_features
# ---------------------------------------------------------------------------------------------------------------
class NativeEMASC(_Base):
    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = EMASCConfig()
        n = len(cfg["in_channels"])
        c.n = n
        c.in_channels = (c_int * 8)(*(list(cfg["in_channels"]) + [0] * (8 - n)))
        c.out_channels = (c_int * 8)(*(list(cfg["out_channels"]) + [0] * (8 - n)))
        with _Weights(state_dict) as w:
            self.h = self.lib.ladi_emasc_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_emasc_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_emasc_destroy(self.h)
            self.h = None

    def __bool__(self):
        return True

    def __call__(self, x, mask=None):
        """x: list of feature tensors (replaced in place like emasc.py:37-40). mask (optional extension): binarised
        [B,1,H,W] mask fused as mask_features."""
        n = len(self.cfg["in_channels"])
        if len(x) != n:
            raise ValueError("EMASC expects %d features" % n)
        src = [_nhwc_buffer(f) for f in x]
        B = src[0].shape[0]
        outs = [torch.empty((B, s.shape[1], s.shape[2], co), dtype=torch.float16, device=s.device) for s, co in zip(src, self.cfg["out_channels"])]
        fa = (c_void_p * n)(*[s.data_ptr() for s in src])
        oa = (c_void_p * n)(*[o.data_ptr() for o in outs])
        hs = (c_int * n)(*[s.shape[1] for s in src])
        ws = (c_int * n)(*[s.shape[2] for s in src])
        mk, Hm, Wm = None, 0, 0
        if mask is not None:
            mk = (mask >= 0.5).to(torch.float16).contiguous()
            Hm, Wm = mk.shape[-2:]
        check(self.lib.ladi_emasc_forward(self.h, fa, hs, ws, B, ptr(mk), Hm, Wm, oa, stream_ptr()), "ladi_emasc_forward")
        for i in range(n):
            x[i] = outs[i].permute(0, 3, 1, 2)
        return x


def mask_features(features, mask):
    """Native counterpart of src/utils/data_utils.py:4-16 on NHWC-backed features (in place)."""
    lib = _lib.load()
    mk = mask.to(torch.float16).contiguous()
    Hm, Wm = mk.shape[-2:]
    for i, f in enumerate(features):
        buf = _nhwc_buffer(f)
        B, h, w, C = buf.shape
        check(lib.ladi_mask_features(ptr(buf), B, h, w, C, ptr(mk), Hm, Wm, stream_ptr()), "ladi_mask_features")
        features[i] = buf.permute(0, 3, 1, 2)
    return features


# ---------------------------------------------------------------------------------------------------------------
# inversion adapter (inference.py:276)
# ---------------------------------------------------------------------------------------------------------------
class NativeInversionAdapter(_Base):
    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = AdapterConfig()
        c.hidden, c.heads, c.mlp_dim, c.head_hidden, c.out_dim = cfg["hidden"], cfg["heads"], cfg["mlp_dim"], cfg["head_hidden"], cfg["out_dim"]
        c.layer_norm_eps = cfg["layer_norm_eps"]
        with _Weights(state_dict) as w:
            self.h = self.lib.ladi_adapter_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_adapter_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_adapter_destroy(self.h)
            self.h = None

    def __call__(self, x):
        xin = x.to(torch.float16).contiguous()
        B, T, H = xin.shape
        out = torch.empty((B, self.cfg["out_dim"]), dtype=torch.float16, device=xin.device)
        check(self.lib.ladi_adapter_forward(self.h, ptr(xin), B, T, ptr(out), stream_ptr()), "ladi_adapter_forward")
        return out.to(x.dtype) if x.dtype in (torch.float32,) else out
