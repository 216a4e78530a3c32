"""Warping-module pieces over libladi_native.  Built so far: the refinement UNet — drop-in for the `refinement`
nn.Module of hubconf.py:57 (src/models/UNet.py UNetVanilla(24, 3, bilinear=True)) called at src/inference.py:264:

    refinement = NativeRefinementUNet(configs.REFINE_FULL, refinement.state_dict())     # once, after hubconf's load_state_dict
    warped_cloth = refinement(torch.cat([im_mask, pose_map, warped_cloth], 1).to(torch.float32)).clamp(-1, 1)

Returns a tensor of the input's dtype (fp32 in the reference's call) and shape [B, 3, H, W]; H and W must be multiples of 16 (512x384 in
the reference).  The TPS matching network (ConvNet_TPS) is not native yet; its oracle is pinned (oracle/warp.py).  No CPU fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import NativeError, RefineConfig, check, dtype_code, ptr, stream_ptr
from .modules import _Weights


class NativeRefinementUNet:
    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = RefineConfig()
        c.in_channels, c.out_channels, c.base_channels, c.bn_eps = cfg["in_channels"], cfg["out_channels"], cfg["base"], cfg.get("bn_eps", 1e-5)
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        with _Weights(sd) as w:
            self.h = self.lib.ladi_refine_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_refine_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_refine_destroy(self.h)
            self.h = None

    def eval(self):
        return self

    def __call__(self, x):
        if x.dim() != 4 or x.shape[1] != self.cfg["in_channels"]:
            raise ValueError("expected [B, %d, H, W]" % self.cfg["in_channels"])
        B, _, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError("H and W must be multiples of 16")
        xin = x.to(self.device)
        if xin.dtype not in (torch.float16, torch.float32):
            xin = xin.float()
        xin = xin.contiguous()
        out = torch.empty((B, self.cfg["out_channels"], H, W), dtype=xin.dtype, device=self.device)
        check(self.lib.ladi_refine_forward(self.h, ptr(xin), dtype_code(xin), B, H, W, ptr(out), dtype_code(out), stream_ptr()), "ladi_refine_forward")
        return out
