"""VAE encode / EMASC / VAE decode stages alone (no hipGraph), the command behind profiles/r03_vae_*: per-stage time, algorithmic
TFLOP/s and fused-minimal HBM GB/s (SURVEY.md §8d per-image figures), and -- under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE -- the
per-kernel HBM bytes of the conv / GroupNorm / attention kernels of these stages.
python tools/bench_vae.py [--batch 8] [--height 512 --width 384] [--iters 3]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ladi_vton_amd as L  # noqa: E402
from ladi_vton_amd import configs as C  # noqa: E402

# per 512x384 image: GFLOP, fused-minimal fp16 GB
WORK = dict(encode=(831.05, 1.464), emasc=(434.87, 0.802), decode=(1879.45, 2.527))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    vcfg, ecfg = C.VAE_FULL, C.EMASC_FULL
    vae = L.NativeVAE(vcfg, C.synth_items(C.vae_shapes(vcfg), "vae."))
    em = L.NativeEMASC(ecfg, C.synth_items(C.emasc_shapes(ecfg), "emasc."))
    B, H, W = a.batch, a.height, a.width
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((B, 3, H, W), generator=g) * 2 - 1).half().to(dev)
    mask = torch.zeros((B, 1, H, W)); mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1
    mask = mask.half().to(dev)
    z = torch.randn((B, 4, H // 8, W // 8), generator=g).to(dev)
    scale = (H * W) / (512.0 * 384.0)

    def stage(fn, n):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, out

    t_enc, (enc, feats) = stage(lambda: vae.encode(x), a.iters)
    t_em, sk = stage(lambda: em([f for f in feats[1:6]], mask=mask), a.iters)
    t_dec, _ = stage(lambda: vae.decode(z, intermediate_features=list(sk), int_layers=[1, 2, 3, 4, 5]), a.iters)
    for name, t in (("encode", t_enc), ("emasc", t_em), ("decode", t_dec)):
        gf, gb = WORK[name]
        print("%-7s B=%d %dx%d: %8.3f ms  %7.1f TFLOP/s algorithmic  %7.1f GB/s fused-minimal" %
              (name, B, H, W, t, gf * scale * B / t, gb * scale * B / t * 1e3))


if __name__ == "__main__":
    main()
