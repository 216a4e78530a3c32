"""One-shot probe of the GroupNorm-of-the-output epilogue (tools/experiments/next/gn_out_fusion.patch applied and built): every admissible
(tile configuration, UNet-level shape) pair against F.group_norm on the fp16-rounded convolution, one line per case, nothing stops at the
first failure.  python tools/experiments/next/gno_probe.py"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from tests import util as U  # noqa: E402


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().float()


def main():
    torch.set_num_threads(U.cpu_quota_threads())
    t_all = time.time()
    shapes = ((2, 64, 320, 32, 24, 0, 330), (3, 192, 640, 32, 24, 1, 310), (2, 128, 320, 64, 48, 1, 300), (4, 640, 1280, 16, 12, 1, 320),
              (1, 1280, 640, 16, 12, 1, 340))
    bp = {84: 128, 85: 192, 88: 128, 89: 192, 92: 192, 96: 192, 97: 192}
    for (N, cin, cout, h, w_, silu, seed) in shapes:
        x = rnd((N, cin, h, w_), seed)
        wt, b = rnd((cout, cin, 3, 3), seed + 1, 1 / math.sqrt(9 * cin)), rnd((cout,), seed + 2, 0.1)
        temb = rnd((cout,), seed + 3, 0.5)
        gamma, beta = 1.0 + rnd((cout,), seed + 4, 0.2), rnd((cout,), seed + 5, 0.2)
        pre = (F.conv2d(x, wt, b, padding=1) + temb[None, :, None, None]).half().float()
        ref = F.group_norm(pre, 32, gamma.half().float(), beta.half().float(), 1e-5)
        if silu:
            ref = F.silu(ref)
        X, Wp = U.nhwc16(x), U.pack_conv_weight(wt)
        # the unfused product path on the same inputs: plain conv, then what the reference arithmetic gives on ITS output
        y0 = U.igemm(X, Wp, cout, bias=b, rowadd=temb, cfg=88 if w_ <= 24 else 84)
        print("shape N=%d %d->%d %dx%d: plain conv vs torch %.2e" % (N, cin, cout, h, w_, U.rel_l2(U.to_nchw(y0), pre)), flush=True)
        for cfg in (88, 84, 89, 85, 92, 96, 97, 0):
            HW = h * w_
            if cfg and ((HW % bp[cfg]) or (cfg in (88, 89, 96, 97) and w_ > 24) or (cfg in (84, 85) and HW == 3072)):
                continue
            try:
                t0 = time.time()
                y = U.igemm(X, Wp, cout, bias=b, rowadd=temb, cfg=cfg, gno=(gamma, beta, 1e-5, 32, silu))
                err = U.rel_l2(U.to_nchw(y), ref)
                same = all(torch.equal(U.igemm(X, Wp, cout, bias=b, rowadd=temb, cfg=cfg, gno=(gamma, beta, 1e-5, 32, silu)), y) for _ in range(3))
                print("  cfg %2d  rel_l2 %.3e  max|d| %.3e  repeat-equal %s  nan %d  %.2fs" % (
                    cfg, err, float((U.to_nchw(y) - ref).abs().max()), same, int(torch.isnan(y).sum()), time.time() - t0), flush=True)
            except Exception as e:   # noqa: BLE001
                print("  cfg %2d  FAILED: %s" % (cfg, str(e)[:200]), flush=True)
    print("total %.1fs" % (time.time() - t_all), flush=True)


if __name__ == "__main__":
    main()
