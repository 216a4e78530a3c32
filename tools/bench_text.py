"""Timing of the native CLIP text encoder + pseudo-word splice and of the ViT-H/14 vision encoder at the bench
batch, with the CPU oracle beside each.
python tools/bench_text.py [--batch 8] [--which text|vision|both]   -> one JSON line per encoder"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ladi_vton_amd as L  # noqa: E402
from ladi_vton_amd import configs as C  # noqa: E402


def bench_vision(a):
    cfg, B = C.VISION_FULL, a.batch
    sd = C.synth_state_dict(C.vision_shapes(cfg), "vision.")
    enc = L.NativeCLIPVisionEncoder(cfg, sd)
    px = (torch.randn((B, 3, 224, 224), generator=torch.Generator().manual_seed(6)) * 1.2).half()
    pxd = px.cuda()
    for _ in range(3):
        out = enc(pxd)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.iters):
        out = enc(pxd)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / a.iters * 1e3
    H, M, Ly, T, d = cfg["hidden"], cfg["mlp_dim"], cfg["layers"], 257, 80
    flop = 2.0 * B * T * Ly * (4 * H * H + 2 * H * M) + 4.0 * B * Ly * cfg["heads"] * T * T * d + 2.0 * B * 256 * 588 * H
    line = {"what": "CLIP ViT-H/14 vision encoder, full size (32 layers, 1280-d, 16 heads of 80), B=%d x 257 tokens" % B,
            "ms_per_batch": round(ms, 3), "tflops": round(flop / ms / 1e9, 1), "gflop_per_batch": round(flop / 1e9, 1)}
    if not a.no_cpu:
        from oracle import vision as OV  # test infrastructure: CPU baseline only
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        with torch.no_grad():
            t0 = time.time(); ref, _ = OV.clip_vision_forward(sd, cfg, px[:2].float()); cpu_s = time.time() - t0
        got = out.last_hidden_state[:2].float().cpu()
        line["cpu_oracle_ms_per_batch"] = round(cpu_s * 1e3 * B / 2, 1)
        line["cpu_sample"] = "2 images timed, scaled to the batch"
        line["rel_l2_vs_oracle"] = float((got - ref).norm() / ref.norm())
    print(json.dumps(line))
    del enc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--which", default="both", choices=["text", "vision", "both"])
    a = ap.parse_args()
    if a.which in ("vision", "both"):
        bench_vision(a)
    if a.which == "vision":
        return
    cfg, B, T, NV = C.TEXT_FULL, a.batch, 77, 16
    sd = C.synth_state_dict(C.text_shapes(cfg), "text.")
    enc = L.NativeCLIPTextEncoder(cfg, sd)
    g = torch.Generator().manual_seed(5)
    ids = torch.zeros((B, T), dtype=torch.int32)
    ids[:, 0] = 49406
    ids[:, 1:10] = torch.randint(300, 40000, (B, 9), generator=g).int()
    ids[:, 10:10 + NV] = 259
    ids[:, 10 + NV] = 49407
    we = (torch.randn((B, NV, cfg["hidden"]), generator=g) * 0.05).half().cuda()
    for _ in range(3):
        out = enc(ids, we, NV)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.iters):
        out = enc(ids, we, NV)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / a.iters * 1e3
    H, M, Ly = cfg["hidden"], cfg["mlp_dim"], cfg["layers"]
    flop = 2.0 * B * T * Ly * (4 * H * H + 2 * H * M) + 4.0 * B * Ly * cfg["heads"] * T * T * 64
    line = {"what": "CLIP text encoder + '$' splice, full size (23 layers, 1024-d), B=%d x 77 tokens, %d pseudo-words" % (B, NV),
            "ms_per_batch": round(ms, 3), "tflops": round(flop / ms / 1e9, 1), "gflop_per_batch": round(flop / 1e9, 1)}
    if not a.no_cpu:
        from oracle import text as OT  # test infrastructure: CPU baseline only
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        with torch.no_grad():
            t0 = time.time(); ref, _ = OT.clip_text_forward(sd, cfg, ids, we.float().cpu(), NV); cpu_s = time.time() - t0
        got = out.last_hidden_state.float().cpu()
        line["cpu_oracle_ms_per_batch"] = round(cpu_s * 1e3, 1)
        line["rel_l2_vs_oracle"] = float((got - ref).norm() / ref.norm())
    print(json.dumps(line))


if __name__ == "__main__":
    main()
