"""round 6, GPU call 23: what the forward's launches lose to COLD operands.  tools/bench_shapes.py (and every isolated A/B of rounds 2-6) times
one launch back to back on the same buffers -- weights and pixels sit in L2 / Infinity Cache.  Inside a forward every launch meets its weights
for the first time (the UNet holds 1.7 GB of them, the Infinity Cache 256 MB).  Per shape: hot (same buffers), cold weights (rotating over
> 600 MB of weight copies), cold weights + a concurrent side-stream touch of the NEXT launch's weights (premise test for a prefetcher)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from ladi_vton_amd import _lib                      # noqa: E402
from ladi_vton_amd._lib import IGemmDesc, stream_ptr  # noqa: E402

SHAPES = [  # (name, n, H, W, cin, cout, ksize)
    ("8x6 conv3 2560->1280", 16, 8, 6, 2560, 1280, 3),
    ("8x6 conv3 1280->1280", 16, 8, 6, 1280, 1280, 3),
    ("16x12 conv3 1280->1280", 16, 16, 12, 1280, 1280, 3),
    ("16x12 lin1 1280->1280", 16, 16, 12, 1280, 1280, 1),
    ("16x12 lin1 1280->3840", 16, 16, 12, 1280, 3840, 1),
    ("16x12 lin1 5120->1280", 16, 16, 12, 5120, 1280, 1),
    ("32x24 lin1 640->640", 16, 32, 24, 640, 640, 1),
    ("32x24 lin1 640->1920", 16, 32, 24, 640, 1920, 1),
    ("32x24 conv3 640->640", 16, 32, 24, 640, 640, 3),
    ("64x48 lin1 320->320", 16, 64, 48, 320, 320, 1),
    ("64x48 conv3 320->320", 16, 64, 48, 320, 320, 3),
]


def main():
    lib = _lib.load()
    d = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    print("%-26s %9s %9s %9s %9s   (us per launch; copies of W)" % ("shape", "hot", "cold_W", "cold_WX", "cold+pf"))
    for name, n, H, W, ci, co, ks in SHAPES:
        K = ks * ks * ci
        wbytes = co * K * 2
        ncopy = max(4, min(64, int(700e6 // wbytes) + 1))
        ws = [(torch.randn((co, K), device=d) * 0.02).half() for _ in range(ncopy)]
        nx = max(2, min(64, int(400e6 // (n * H * W * ci * 2)) + 1))
        xs = [(torch.randn((n, H, W, ci), device=d) * 0.5).half() for _ in range(nx)]
        out = torch.empty((n, H, W, co), dtype=torch.float16, device=d)
        bias = torch.zeros((co,), dtype=torch.float16, device=d)

        def desc(x, w):
            g = IGemmDesc()
            g.src0, g.C0, g.ld0 = x.data_ptr(), ci, ci
            g.Hs, g.Ws, g.Ho, g.Wo, g.P = H, W, H, W, n * H * W
            g.ksize, g.stride, g.pad, g.ups = ks, 1, ks // 2, 0
            g.W, g.Q, g.K, g.ldw = w.data_ptr(), co, K, 0
            g.bias, g.act, g.out_scale = bias.data_ptr(), 0, 1.0
            g.out, g.ldo, g.out_f32 = out.data_ptr(), co, 0
            return g
        descs_hot = [desc(xs[0], ws[0])]
        descs_cw = [desc(xs[0], w) for w in ws]
        descs_cwx = [desc(xs[i % nx], ws[i % ncopy]) for i in range(max(nx, ncopy))]

        def run(descs, iters, prefetch=False):
            st = stream_ptr()
            for i in range(8):
                assert lib.ladi_op_igemm(ctypes.byref(descs[i % len(descs)]), 1, 0, st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                if prefetch:                               # touch the NEXT launch's weights from a side stream while this launch runs
                    nxt = ws[(i + 1) % ncopy]
                    with torch.cuda.stream(side):
                        nxt.view(-1).view(torch.int32)[::32].sum()
                assert lib.ladi_op_igemm(ctypes.byref(descs[i % len(descs)]), 1, 0, st) == 0
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3
        iters = 3 * max(nx, ncopy)
        hot = run(descs_hot, iters)
        cw = run(descs_cw, iters)
        cwx = run(descs_cwx, iters)
        pf = run(descs_cw, iters, prefetch=True)
        print("%-26s %9.1f %9.1f %9.1f %9.1f   (%d x %.1f MB)" % (name, hot, cw, cwx, pf, ncopy, wbytes / 1e6), flush=True)
        del ws, xs


if __name__ == "__main__":
    main()
