"""round 6, GPU call 22: hipBLASLt (torch F.linear, fp16) on the plain-GEMM shapes of the CFG UNet forward (1x1 convolutions and linears of
tools/r06/call20.sh's per-shape table), as the known-achievable rate per shape next to the HIP path's per-launch time.  A measurement script:
imports torch only (no oracle, no product library)."""
import sys
import torch
import torch.nn.functional as F

SHAPES = [  # (P, Q, K, hip_path_us from gpurun_out/r06c20/prof_dump.txt, what)
    (3072, 1280, 1280, 27.4, "16x12 projections (25 per forward)"),
    (12288, 5120, 640, 111.9, "32x24 GEGLU ff.net.0 (value+gate, GELU and product fused in ours)"),
    (49152, 2560, 320, 110.5, "64x48 GEGLU ff.net.0"),
    (3072, 10240, 1280, 100.5, "16x12 GEGLU ff.net.0"),
    (49152, 320, 320, 29.9, "64x48 projections (25 per forward)"),
    (12288, 640, 640, 26.9, "32x24 projections (25 per forward)"),
    (49152, 320, 1280, 67.9, "64x48 ff.net.2 (+ residual in ours)"),
    (3072, 1280, 5120, 63.0, "16x12 ff.net.2"),
    (12288, 1920, 640, 59.4, "32x24 fused q/k/v"),
    (49152, 960, 320, 56.6, "64x48 fused q/k/v (LayerNorm fused in ours)"),
    (3072, 3840, 1280, 56.2, "16x12 fused q/k/v"),
    (12288, 640, 2560, 55.2, "32x24 ff.net.2"),
    (768, 1280, 1280, 19.1, "8x6 projections"),
    (49152, 320, 640, 41.1, "64x48 shortcut 1x1 on the concat"),
    (3072, 1280, 2560, 39.2, "16x12 shortcut 1x1"),
]


def t_us(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(1)
    print("%-8s %-6s %-6s %10s %10s %10s %10s  %s" % ("P", "Q", "K", "vendor_us", "vendor_TF", "hip_us", "hip_TF", "what"))
    for P, Q, K, ours, what in SHAPES:
        a = (torch.randn((P, K), generator=g) * 0.1).half().cuda()
        w = (torch.randn((Q, K), generator=g) * 0.1).half().cuda()
        b = torch.zeros((Q,), dtype=torch.float16, device="cuda")
        with torch.no_grad():
            us = min(t_us(lambda: F.linear(a, w)), t_us(lambda: F.linear(a, w, b)))
        fl = 2.0 * P * Q * K
        print("%-8d %-6d %-6d %10.1f %10.0f %10.1f %10.0f  %s" % (P, Q, K, us, fl / us / 1e6, ours, fl / ours / 1e6, what))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
