"""Warping module over libladi_native: drop-ins for the two nn.Modules hubconf.py:56-58 returns.

    tps, refinement = NativeTPS(configs.TPS_FULL, tps.state_dict()), NativeRefinementUNet(configs.REFINE_FULL, refinement.state_dict())
    low_grid, theta, rx, ry, cx, cy, rg, cg = tps(low_cloth.to(torch.float32), agnostic.to(torch.float32))          # inference.py:253
    warped_cloth = refinement(torch.cat([im_mask, pose_map, warped_cloth], 1).to(torch.float32)).clamp(-1, 1)       # inference.py:263-265

`NativeTPS` returns the sampling grid [B, 256, 192, 2] (fp32, what F.grid_sample consumes) and the source control points `theta`
[B, 25, 2]; the six training-only regulariser values of the reference's 8-tuple are returned as None (inference.py never reads them).
`NativeRefinementUNet` returns a tensor of the input's dtype, [B, 3, H, W], H and W multiples of 16.  The resizes and the grid_sample call
between the two stay on the caller's side (torchvision / torch.nn.functional, inference.py:242-260).  No CPU fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import NativeError, RefineConfig, TpsConfig, check, dtype_code, ptr, stream_ptr
from .modules import _Shim, _Weights


class NativeRefinementUNet(_Shim):
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


class NativeTPS(_Shim):
    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = TpsConfig()
        c.height, c.width, c.input_nc, c.n_layers, c.grid_size, c.ngf = cfg["height"], cfg["width"], cfg["input_nc"], cfg["n_layers"], cfg["grid"], cfg["ngf"]
        c.bn_eps = cfg.get("bn_eps", 1e-5)
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked") and not k.startswith("gridGen.")}
        with _Weights(sd) as w:
            self.h = self.lib.ladi_tps_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_tps_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_tps_destroy(self.h)
            self.h = None

    def eval(self):
        return self

    def __call__(self, input_a, input_b):
        H, W = self.cfg["height"], self.cfg["width"]
        if tuple(input_a.shape[1:]) != (3, H, W) or tuple(input_b.shape[1:]) != (self.cfg["input_nc"], H, W) or input_a.shape[0] != input_b.shape[0]:
            raise ValueError("expected inputA [B, 3, %d, %d] and inputB [B, %d, %d, %d]" % (H, W, self.cfg["input_nc"], H, W))
        dt = torch.float16 if input_a.dtype == torch.float16 and input_b.dtype == torch.float16 else torch.float32
        a = input_a.to(device=self.device, dtype=dt).contiguous()
        b = input_b.to(device=self.device, dtype=dt).contiguous()
        B, N = a.shape[0], self.cfg["grid"] ** 2
        grid = torch.empty((B, H, W, 2), dtype=torch.float32, device=self.device)
        coor = torch.empty((B, N, 2), dtype=torch.float32, device=self.device)
        check(self.lib.ladi_tps_forward(self.h, ptr(a), ptr(b), dtype_code(a), B, ptr(grid), ptr(coor), stream_ptr()), "ladi_tps_forward")
        return grid, coor, None, None, None, None, None, None


def resize_antialias(x, size):
    """torchvision.transforms.functional.resize(x, size, InterpolationMode.BILINEAR, antialias=True) on the device (src/inference.py:242-258,
    267-268): x [B, C, H, W] fp32 / fp16 -> [B, C, size[0], size[1]], same dtype."""
    if x.dim() != 4:
        raise ValueError("expected a [B, C, H, W] tensor")
    lib = _lib.load()
    xin = x.to(torch.device("cuda", torch.cuda.current_device()))
    if xin.dtype not in (torch.float16, torch.float32):
        xin = xin.float()
    xin = xin.contiguous()
    B, C, H, W = xin.shape
    Ho, Wo = int(size[0]), int(size[1])
    out = torch.empty((B, C, Ho, Wo), dtype=xin.dtype, device=xin.device)
    check(lib.ladi_op_resize_bilinear_aa(ptr(xin), dtype_code(xin), B * C, H, W, ptr(out), dtype_code(out), Ho, Wo, stream_ptr()),
          "ladi_op_resize_bilinear_aa")
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(cloth, size=224, mean=CLIP_MEAN, std=CLIP_STD):
    """src/inference.py:268-272 in one kernel: resize((cloth + 1) / 2, (size, size), antialias=True).clamp(0, 1), the processor's 8-bit
    round trip floor(v * 255) / 255 (transformers 4.27.3 CLIPImageProcessor on a float32 image: the reference's default, non-mixed-
    precision run -- include/ladi_native.h says what is and is not emulated), CLIP mean / std normalisation, fp16 pixel_values
    [B, 3, size, size] for the vision encoder.  cloth [B, 3, H, W] in [-1, 1], fp32 / fp16."""
    if cloth.dim() != 4 or cloth.shape[1] != 3:
        raise ValueError("expected a [B, 3, H, W] tensor")
    lib = _lib.load()
    x = cloth.to(torch.device("cuda", torch.cuda.current_device()))
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    x = x.contiguous()
    B, _, H, W = x.shape
    out = torch.empty((B, 3, size, size), dtype=torch.float16, device=x.device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib.ladi_op_clip_preprocess(ptr(x), dtype_code(x), B, H, W, int(size), m3, s3, ptr(out), stream_ptr()), "ladi_op_clip_preprocess")
    return out


def grid_sample_border(x, grid):
    """F.grid_sample(x, grid, padding_mode="border") (bilinear, align_corners=False; src/inference.py:260): x [B, C, H, W], grid
    [B, Ho, Wo, 2] in [-1, 1] (x, y) -> [B, C, Ho, Wo] of x's dtype."""
    if x.dim() != 4 or grid.dim() != 4 or grid.shape[-1] != 2 or grid.shape[0] != x.shape[0]:
        raise ValueError("expected x [B, C, H, W] and grid [B, Ho, Wo, 2]")
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    xin = x.to(dev)
    if xin.dtype not in (torch.float16, torch.float32):
        xin = xin.float()
    xin = xin.contiguous()
    g = grid.to(device=dev, dtype=torch.float32).contiguous()
    B, C, H, W = xin.shape
    Ho, Wo = g.shape[1], g.shape[2]
    out = torch.empty((B, C, Ho, Wo), dtype=xin.dtype, device=dev)
    check(lib.ladi_op_grid_sample_border(ptr(xin), dtype_code(xin), B, C, H, W, ptr(g), Ho, Wo, ptr(out), dtype_code(out), stream_ptr()),
          "ladi_op_grid_sample_border")
    return out


def warp_cloth(tps, refinement, cloth, im_mask, pose_map, low_size=(256, 192)):
    """The warping stage of src/inference.py:239-266 on the native kernels end to end: antialiased down-sampling of cloth / agnostic
    inputs, TPS matching network, grid up-sampling, border grid_sample and the refinement UNet.  Returns the refined warped cloth
    [B, 3, H, W] (fp32, clamped to [-1, 1]) and the TPS control points."""
    H, W = cloth.shape[-2:]
    low_cloth = resize_antialias(cloth, low_size)
    agnostic = torch.cat([resize_antialias(im_mask, low_size), resize_antialias(pose_map, low_size)], 1)
    low_grid, theta = tps(low_cloth.to(torch.float32), agnostic.to(torch.float32))[:2]
    grid = resize_antialias(low_grid.permute(0, 3, 1, 2), (H, W)).permute(0, 2, 3, 1)
    warped = grid_sample_border(cloth.to(torch.float32), grid)
    dev = warped.device
    refined = refinement(torch.cat([im_mask.to(dev, torch.float32), pose_map.to(dev, torch.float32), warped], 1))
    return refined.clamp(-1, 1), theta
