"""CPU fp32 oracle of the warping module: TPS geometric matching + grid_sample + refinement UNet.
TEST INFRASTRUCTURE ONLY (no product code exists for this row yet; the oracle and its reference pins come first, per the scope order).

Restates, from state_dict tensors:
  * `ConvNet_TPS.forward` (reference src/models/ConvNet_TPS.py:315-337): FeatureExtraction x2 (:28-56, conv4x4 s2 -> ReLU -> BatchNorm,
    twice conv3x3), FeatureL2Norm (:59-66), FeatureCorrelation (:69-81), BoundedGridLocNet's regression (:92-127, 197-207; its
    regulariser terms :208-224 are training-only and hard-code `.cuda()`, so they are not part of the inference path), TPSGridGen
    (:130-185) with target control points on the 5x5 lattice of range 0.9 (:291-306);
  * the call-site warp `F.grid_sample(cloth, grid, padding_mode='border')` (src/inference.py:260);
  * `UNetVanilla.forward` (src/models/UNet.py:23-34, src/models/unet_parts.py: DoubleConv / Down / Up(bilinear, align_corners=True) /
    OutConv) as instantiated by hubconf.py:56-58.
BatchNorm is evaluated in inference mode (running statistics).  The bilinear antialiased resizes around the module (inference.py:242-258)
are torchvision calls on the caller's side and are not restated.

Pinned (tests/test_cpu.py::test_warp_oracle_matches_reference_modules_golden) against tests/golden/warp_modules.safetensors, produced by
oracle/make_golden.py from the REAL reference modules with the deterministic synthetic checkpoint.
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def feature_extraction(sd, p, x, n_layers):
    i = 0
    for _ in range(n_layers + 1):                                     # conv4x4 s2 -> ReLU -> BN
        x = _bn(sd, "%s.model.%d" % (p, i + 2), F.relu(_conv(sd, "%s.model.%d" % (p, i), x, 2, 1)))
        i += 3
    x = _bn(sd, "%s.model.%d" % (p, i + 2), F.relu(_conv(sd, "%s.model.%d" % (p, i), x)))
    return F.relu(_conv(sd, "%s.model.%d" % (p, i + 3), x))


def l2norm(f):
    return f / torch.sqrt((f * f).sum(1, keepdim=True) + 1e-6)


def correlation(fa, fb):
    b, c, h, w = fa.shape
    a = fa.transpose(2, 3).reshape(b, c, h * w)                       # A positions enumerated column-major (w outer, h inner)
    bb = fb.reshape(b, c, h * w).transpose(1, 2)
    return torch.bmm(bb, a).view(b, h, w, h * w).transpose(2, 3).transpose(1, 2)


def regression(sd, x):
    p = "loc_net.regression.conv"
    for i, (s, pad) in enumerate(((2, 1), (2, 1), (1, 1), (1, 1))):
        x = F.relu(_bn(sd, "%s.%d" % (p, 3 * i + 1), _conv(sd, "%s.%d" % (p, 3 * i), x, s, pad)))   # conv -> BN -> ReLU
    x = F.linear(x.reshape(x.shape[0], -1), sd["loc_net.regression.linear.weight"], sd["loc_net.regression.linear.bias"])
    return torch.tanh(x)


def _partial_repr(inp, ctrl):
    d = inp.view(-1, 1, 2) - ctrl.view(1, -1, 2)
    r2 = (d * d).sum(2)
    rep = 0.5 * r2 * torch.log(r2)
    return torch.where(torch.isnan(rep), torch.zeros_like(rep), rep)


def tps_grid_matrices(height, width, grid=5, rng=0.9):
    """TPSGridGen.__init__ (:132-170) for the 5x5 control lattice of ConvNet_TPS.__init__ (:291-306)"""
    pts = torch.tensor(list(itertools.product(np.arange(-rng, rng + 0.00001, 2.0 * rng / (grid - 1)),
                                              np.arange(-rng, rng + 0.00001, 2.0 * rng / (grid - 1)))), dtype=torch.float32)
    ctrl = torch.cat([pts[:, 1:2], pts[:, 0:1]], dim=1)              # (y, x) -> (x, y)
    n = ctrl.shape[0]
    K = torch.zeros(n + 3, n + 3)
    K[:n, :n] = _partial_repr(ctrl, ctrl)
    K[:n, -3] = 1
    K[-3, :n] = 1
    K[:n, -2:] = ctrl
    K[-2:, :n] = ctrl.t()
    inv = torch.inverse(K)
    yy, xx = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    coord = torch.stack([xx.reshape(-1) * 2 / (width - 1) - 1, yy.reshape(-1) * 2 / (height - 1) - 1], dim=1)
    rep = torch.cat([_partial_repr(coord, ctrl), torch.ones(height * width, 1), coord], dim=1)
    return ctrl, inv, rep


def tps_forward(sd, cfg, input_a, input_b):
    """-> (grid [B, H, W, 2], source control points [B, 25, 2])"""
    fa = l2norm(feature_extraction(sd, "extractionA", input_a, cfg["n_layers"]))
    fb = l2norm(feature_extraction(sd, "extractionB", input_b, cfg["n_layers"]))
    coor = regression(sd, correlation(fa, fb)).view(input_a.shape[0], -1, 2)
    _, inv, rep = tps_grid_matrices(cfg["height"], cfg["width"], cfg["grid"])
    y = torch.cat([coor, torch.zeros(coor.shape[0], 3, 2)], dim=1)
    grid = torch.matmul(rep, torch.matmul(inv, y))
    return grid.view(-1, cfg["height"], cfg["width"], 2), coor


def warp(cloth, grid):
    return F.grid_sample(cloth, grid, padding_mode="border")


def _double_conv(sd, p, x):
    x = F.relu(_bn(sd, p + ".double_conv.1", F.conv2d(x, sd[p + ".double_conv.0.weight"], None, padding=1)))
    return F.relu(_bn(sd, p + ".double_conv.4", F.conv2d(x, sd[p + ".double_conv.3.weight"], None, padding=1)))


def _up(sd, p, x1, x2):
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return _double_conv(sd, p + ".conv", torch.cat([x2, x1], dim=1))


def refinement_forward(sd, x):
    x1 = _double_conv(sd, "inc", x)
    x2 = _double_conv(sd, "down1.maxpool_conv.1", F.max_pool2d(x1, 2))
    x3 = _double_conv(sd, "down2.maxpool_conv.1", F.max_pool2d(x2, 2))
    x4 = _double_conv(sd, "down3.maxpool_conv.1", F.max_pool2d(x3, 2))
    x5 = _double_conv(sd, "down4.maxpool_conv.1", F.max_pool2d(x4, 2))
    x = _up(sd, "up1", x5, x4)
    x = _up(sd, "up2", x, x3)
    x = _up(sd, "up3", x, x2)
    x = _up(sd, "up4", x, x1)
    return F.conv2d(x, sd["outc.conv.weight"], sd["outc.conv.bias"])


def resize_antialias(x, size):
    """torchvision.transforms.functional.resize(x, size, BILINEAR, antialias=True) for tensors (src/inference.py:242-258)"""
    return F.interpolate(x, size=size, mode="bilinear", antialias=True, align_corners=False)


def warp_cloth(tsd, tcfg, rsd, cloth, im_mask, pose_map, low_size=(256, 192)):
    """the composed warping stage, src/inference.py:239-266 -> (refined warped cloth [B,3,H,W] clamped to [-1,1], theta, low_grid, warped)"""
    H, W = cloth.shape[-2:]
    low_cloth = resize_antialias(cloth, low_size)
    agnostic = torch.cat([resize_antialias(im_mask, low_size), resize_antialias(pose_map, low_size)], 1)
    low_grid, theta = tps_forward(tsd, tcfg, low_cloth.float(), agnostic.float())
    grid = resize_antialias(low_grid.permute(0, 3, 1, 2), (H, W)).permute(0, 2, 3, 1)
    warped = warp(cloth.float(), grid.float())
    refined = refinement_forward(rsd, torch.cat([im_mask, pose_map, warped], 1).float()).clamp(-1, 1)
    return refined, theta, low_grid, warped
