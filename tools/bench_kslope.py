"""Where does the time of the small token-wise GEMMs go?  For the projection shapes of the three transformer levels at the bench batch
(P x Q = 49152 x 320, 12288 x 640, 3072 x 1280) this times a 1x1 layer over a sweep of K (same tiles, same epilogue): the slope of
time(K) is the cost of a K step, the intercept is what a launch pays regardless of its reduction depth (ramp-up, prologue latency,
epilogue, drain).  python tools/bench_kslope.py [--iters 30] [--json out.json]"""
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

LEVELS = [(64, 48, 320), (32, 24, 640), (16, 12, 1280)]
KS = [320, 640, 1280, 2560, 5120]
CFGS = [9, 48, 7, 47, 14, 62, 66, 22, 33, 25, 26, 56, 57, 4, 17]


def problem(n, H, W, cin, cout, res):
    dev = U.dev()
    x = torch.randn((n, H, W, cin), dtype=torch.float16, device=dev)
    w = torch.randn((cout, cin), dtype=torch.float16, device=dev) * 0.02
    b = torch.randn((cout,), dtype=torch.float16, device=dev)
    out = torch.empty((n, H, W, cout), dtype=torch.float16, device=dev)
    d = IGemmDesc()
    d.src0, d.C0, d.ld0 = x.data_ptr(), cin, cin
    d.Hs, d.Ws, d.Ho, d.Wo, d.P = H, W, H, W, n * H * W
    d.ksize, d.stride, d.pad = 1, 1, 0
    d.W, d.Q, d.K = w.data_ptr(), cout, cin
    d.bias, d.out, d.ldo, d.out_scale = b.data_ptr(), out.data_ptr(), cout, 1.0
    keep = [x, w, b, out]
    if res:
        r = torch.randn((n, H, W, cout), dtype=torch.float16, device=dev)
        d.res0, d.ldr0 = r.data_ptr(), cout
        keep.append(r)
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
    return e0.elapsed_time(e1) / iters * 1000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    lib = _lib.load()
    lib.ladi_igemm_set_autotune(0)
    rows = []
    for (H, W, C) in LEVELS:
        for res in (False, True):
            print("P = %d, Q = %d, %s" % (a.n * H * W, C, "bias + residual" if res else "bias"))
            print("  cfg  " + "".join("K=%-9d" % k for k in KS) + " us per K step of 64 (slope) | intercept us")
            for cfg in CFGS:
                ts = []
                for K in KS:
                    d, keep = problem(a.n, H, W, K, C, res)
                    ts.append(run(lib, d, cfg, a.iters))
                    del keep
                if any(t is None for t in ts):
                    continue
                # least squares over the sweep
                xs = [k / 64.0 for k in KS]
                mx, my = sum(xs) / len(xs), sum(ts) / len(ts)
                slope = sum((x - mx) * (t - my) for x, t in zip(xs, ts)) / sum((x - mx) ** 2 for x in xs)
                icpt = my - slope * mx
                print("  %-4d " % cfg + "".join("%-11.1f" % t for t in ts) + " %.3f | %.1f" % (slope, icpt), flush=True)
                rows.append(dict(H=H, W=W, Q=C, res=res, cfg=cfg, us=dict(zip(map(str, KS), ts)), slope_us_per_kstep=slope, intercept_us=icpt))
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
