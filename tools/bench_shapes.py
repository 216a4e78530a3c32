"""Every implicit-GEMM shape of one CFG UNet forward (n = 2 x batch samples at 64x48 latents) x every admissible tile configuration
(run on the GPU box).  Prints, per shape, the fastest configuration among the round-2 set and among all, and the per-forward totals
weighted by the number of launches of each shape -- the table behind the tile-selection changes of round 3.

python tools/bench_shapes.py [--n 16] [--iters 10] [--json gpurun_out/shapes.json] [--cfgs 7,39,...]"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util as U  # noqa: E402
from ladi_vton_amd import _lib  # noqa: E402
from ladi_vton_amd._lib import IGemmDesc, stream_ptr  # noqa: E402

# (pixels per sample H, W, Cin, Cout, ksize, launches per forward, epilogue) of the UNet at 64x48 latents; from LADI_PROF_DUMP=1 of the
# round-2 bench (profiles/r02_*): P = n*H*W, K = ksize^2 * Cin
SHAPES = [
    (64, 48, 320, 320, 3, 7, ""), (64, 48, 640, 320, 3, 2, ""), (64, 48, 960, 320, 3, 1, ""), (64, 48, 640, 640, 3, 1, ""),
    (64, 48, 64, 320, 3, 1, ""), (64, 48, 320, 4, 3, 1, ""),
    (64, 48, 320, 320, 1, 25, "res"), (64, 48, 320, 960, 1, 5, ""), (64, 48, 320, 2560, 1, 5, "geglu"), (64, 48, 1280, 320, 1, 5, "res"),
    (64, 48, 640, 320, 1, 2, ""), (64, 48, 960, 320, 1, 1, ""),
    (32, 24, 320, 320, 3, 1, ""), (32, 24, 320, 640, 3, 1, ""), (32, 24, 640, 640, 3, 6, ""), (32, 24, 960, 640, 3, 1, ""),
    (32, 24, 1280, 640, 3, 1, ""), (32, 24, 1920, 640, 3, 1, ""), (32, 24, 1280, 1280, 3, 1, ""),
    (32, 24, 640, 640, 1, 25, "res"), (32, 24, 640, 1920, 1, 5, ""), (32, 24, 640, 5120, 1, 5, "geglu"), (32, 24, 2560, 640, 1, 5, "res"),
    (32, 24, 320, 640, 1, 1, ""), (32, 24, 960, 640, 1, 1, ""), (32, 24, 1280, 640, 1, 1, ""), (32, 24, 1920, 640, 1, 1, ""),
    (16, 12, 640, 640, 3, 1, ""), (16, 12, 640, 1280, 3, 1, ""), (16, 12, 1280, 1280, 3, 7, ""), (16, 12, 1920, 1280, 3, 1, ""),
    (16, 12, 2560, 1280, 3, 2, ""),
    (16, 12, 1280, 1280, 1, 25, "res"), (16, 12, 1280, 3840, 1, 5, ""), (16, 12, 1280, 10240, 1, 5, "geglu"), (16, 12, 5120, 1280, 1, 5, "res"),
    (16, 12, 640, 1280, 1, 1, ""), (16, 12, 1920, 1280, 1, 1, ""), (16, 12, 2560, 1280, 1, 2, ""),
    (8, 6, 1280, 1280, 3, 12, ""), (8, 6, 2560, 1280, 3, 3, ""),
    (8, 6, 1280, 1280, 1, 5, "res"), (8, 6, 1280, 3840, 1, 1, ""), (8, 6, 1280, 10240, 1, 1, "geglu"), (8, 6, 5120, 1280, 1, 1, "res"),
    (8, 6, 2560, 1280, 1, 3, ""),
]
R2_MAX = 38   # configurations 1..38 existed in round 2


def make_problem(n, H, W, cin, cout, k, epi):
    dev = U.dev()
    cp = (cin + 63) // 64 * 64
    x = torch.randn((n, H, W, cp), dtype=torch.float16, device=dev)
    w = torch.randn((cout, k * k * cp), dtype=torch.float16, device=dev) * 0.02
    b = torch.randn((cout,), dtype=torch.float16, device=dev)
    qout = cout // 2 if epi == "geglu" else cout
    ldo = (qout + 7) // 8 * 8
    out = torch.empty((n, H, W, ldo), dtype=torch.float16, device=dev)
    d = IGemmDesc()
    d.src0, d.C0, d.ld0 = x.data_ptr(), cp, cp
    d.Hs, d.Ws, d.Ho, d.Wo, d.P = H, W, H, W, n * H * W
    d.ksize, d.stride, d.pad, d.ups = k, 1, k // 2, 0
    d.W, d.Q, d.K = w.data_ptr(), cout, k * k * cp
    d.bias = b.data_ptr()
    d.out, d.ldo, d.out_scale = out.data_ptr(), ldo, 1.0
    keep = [x, w, b, out]
    if epi == "geglu":
        d.act = 3
    if epi == "res":
        res = torch.randn((n, H, W, cout), dtype=torch.float16, device=dev)
        d.res0, d.ldr0 = res.data_ptr(), cout
        keep.append(res)
    return d, keep


def run(lib, d, cfg, iters):
    st = stream_ptr()
    for _ in range(2):
        if lib.ladi_op_igemm(ctypes.byref(d), 1, cfg, st) != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.ladi_op_igemm(ctypes.byref(d), 1, cfg, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cfgs", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--filter", default="", help="substring of the shape name (e.g. conv3)")
    a = ap.parse_args()
    lib = _lib.load()
    lib.ladi_igemm_set_autotune(0)
    ncfg = lib.ladi_igemm_cfg_count()
    cfgs = [int(c) for c in a.cfgs.split(",")] if a.cfgs else list(range(1, ncfg + 1))
    tot_old = tot_new = 0.0
    rows = []
    print("%-34s %5s | %-18s | %-18s | gain  | runners-up" % ("shape", "x", "best of cfg 1..%d" % R2_MAX, "best of all"))
    for (H, W, cin, cout, k, calls, epi) in SHAPES:
        if a.filter and a.filter not in "%dx%d %s%d %d->%d %s" % (H, W, "conv" if k == 3 else "lin", k, cin, cout, epi):
            continue
        res = {}
        d, keep = make_problem(a.n, H, W, cin, cout, k, epi)
        for c in cfgs:
            t = run(lib, d, c, a.iters)
            if t is not None:
                res[c] = t
        if not res:
            continue
        old = {c: t for c, t in res.items() if c <= R2_MAX}
        bo = min(old, key=old.get) if old else None
        bn = min(res, key=res.get)
        gf = 2.0 * a.n * H * W * cout * k * k * cin / 1e9
        top = sorted(res.items(), key=lambda kv: kv[1])[:5]
        name = "%dx%d %s%d %d->%d %s" % (H, W, "conv" if k == 3 else "lin", k, cin, cout, epi)
        print("%-34s %5d | cfg%-3s %7.1f us %4.0f | cfg%-3d %7.1f us %4.0f | %5.2f | %s" % (
            name, calls, bo, old[bo] if old else 0, gf / old[bo] * 1e3 if old else 0, bn, res[bn], gf / res[bn] * 1e3,
            (old[bo] / res[bn]) if old else 0, " ".join("%d:%.0f" % kv for kv in top)), flush=True)
        if old:
            tot_old += old[bo] * calls
        tot_new += res[bn] * calls
        rows.append(dict(shape=name, H=H, W=W, cin=cin, cout=cout, k=k, calls=calls, epi=epi, gflop=gf, us={str(c): round(t, 2) for c, t in res.items()}))
    print("per forward: best-of-old %.3f ms, best-of-all %.3f ms" % (tot_old / 1e3, tot_new / 1e3))
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        json.dump(dict(n=a.n, iters=a.iters, rows=rows, total_old_ms=tot_old / 1e3, total_new_ms=tot_new / 1e3), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
