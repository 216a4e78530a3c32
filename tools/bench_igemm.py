"""Micro-benchmark of the implicit-GEMM kernel family on the hot-path layer shapes (run on the GPU box).
python tools/bench_igemm.py [--cfgs 0,1,2,3,4] [--n 16]  -> TFLOP/s per shape and tile config (algorithmic 2*P*Q*K)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util as U  # noqa: E402
from ladi_vton_amd import _lib  # noqa: E402
from ladi_vton_amd._lib import IGemmDesc, stream_ptr  # noqa: E402
import ctypes  # noqa: E402

# (name, H, W, Cin, Cout, ksize, per-sample scaling: True = multiply batch by n)
SHAPES = [
    ("unet conv3 320->320 @64x48", 64, 48, 320, 320, 3),
    ("ovh lin 64->320 T3072", 3072, 1, 64, 320, 1),
    ("ovh lin 128->320 T3072", 3072, 1, 128, 320, 1),
    ("ovh lin 256->320 T3072", 3072, 1, 256, 320, 1),
    ("ovh lin 512->320 T3072", 3072, 1, 512, 320, 1),
    ("ovh conv3 64->320 @64x48", 64, 48, 64, 320, 3),
    ("unet conv3 640->640 @32x24", 32, 24, 640, 640, 3),
    ("unet conv3 1280->1280 @16x12", 16, 12, 1280, 1280, 3),
    ("unet conv3 1280->1280 @8x6", 8, 6, 1280, 1280, 3),
    ("unet conv3 2560->1280 @16x12", 16, 12, 2560, 1280, 3),
    ("unet conv3 960->320 @64x48", 64, 48, 960, 320, 3),
    ("unet lin 320->320 T3072", 3072, 1, 320, 320, 1),
    ("unet lin 320->2560 T3072 (geglu)", 3072, 1, 320, 2560, 1),
    ("unet lin 1280->320 T3072", 3072, 1, 1280, 320, 1),
    ("unet lin 640->5120 T768 (geglu)", 768, 1, 640, 5120, 1),
    ("unet lin 1280->10240 T192", 192, 1, 1280, 10240, 1),
    ("unet lin 5120->1280 T192", 192, 1, 5120, 1280, 1),
    ("unet lin 1280->1280 T192", 192, 1, 1280, 1280, 1),
    ("unet lin 640->640 T768", 768, 1, 640, 640, 1),
    ("unet lin 320->960 T3072 (qkv)", 3072, 1, 320, 960, 1),
    ("unet lin 320->320 T3072 +res", 3072, 1, 320, 320, 1),
    ("unet lin 640->640 T768 +res", 768, 1, 640, 640, 1),
    ("unet lin 640->1920 T768 (qkv)", 768, 1, 640, 1920, 1),
]
VAE_SHAPES = [
    ("vae conv3 128->128 @512x384", 512, 384, 128, 128, 3),
    ("vae conv3 256->256 @256x192", 256, 192, 256, 256, 3),
    ("vae conv3 512->512 @128x96", 128, 96, 512, 512, 3),
    ("vae conv3 512->512 @64x48", 64, 48, 512, 512, 3),
]


def run(name, n, H, W, cin, cout, k, cfg, iters):
    lib = _lib.load()
    dev = U.dev()
    x = torch.randn((n, H, W, cin), dtype=torch.float16, device=dev)
    w = (torch.randn((cout, k * k * cin), dtype=torch.float16, device=dev) * 0.02)
    out = torch.empty((n, H, W, cout), dtype=torch.float16, device=dev)
    d = IGemmDesc()
    d.src0, d.C0, d.ld0 = x.data_ptr(), cin, cin
    d.Hs, d.Ws, d.Ho, d.Wo, d.P = H, W, H, W, n * H * W
    d.ksize, d.stride, d.pad, d.ups = k, 1, k // 2, 0
    d.W, d.Q, d.K = w.data_ptr(), cout, k * k * cin
    d.out, d.ldo, d.out_scale = out.data_ptr(), cout, 1.0
    if "geglu" in name:
        d.act = 3
        d.ldo = cout // 2
    if "+res" in name:
        d.res0, d.ldr0 = x.data_ptr(), cin
    st = stream_ptr()
    for _ in range(3):
        rc = lib.ladi_op_igemm(ctypes.byref(d), 1, cfg, st)
        if rc != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.ladi_op_igemm(ctypes.byref(d), 1, cfg, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * n * H * W * cout * k * k * cin / ms / 1e9, ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="0,1,2,3,4,5,6")
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--vae-n", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(",")]
    print("%-40s " % "shape" + " ".join("cfg%-2d TF/s (ms)   " % c for c in cfgs))
    for shapes, n in ((SHAPES, a.n), (VAE_SHAPES, a.vae_n)):
        for (name, H, W, cin, cout, k) in shapes:
            if a.only and a.only not in name:
                continue
            cells = []
            for c in cfgs:
                r = run(name, n, H, W, cin, cout, k, c, a.iters)
                cells.append("%7.1f (%6.3f)   " % r if r else "   n/a            ")
            print("%-40s " % name + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
