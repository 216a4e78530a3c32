"""Micro-benchmark of flash_attn64 / GroupNorm / LayerNorm on the UNet shapes (run on the GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util as U  # noqa: E402
from ladi_vton_amd import _lib  # noqa: E402
from ladi_vton_amd._lib import ptr, stream_ptr  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    lib = _lib.load()
    dev = U.dev()
    n = 16
    print("flash_attn64 (n=%d)" % n)
    for name, heads, Nq, Nk in (("self L0", 5, 3072, 3072), ("self L1", 10, 768, 768), ("self L2", 20, 192, 192), ("cross L0", 5, 3072, 77),
                                ("cross L1", 10, 768, 77), ("cross L2", 20, 192, 77)):
        if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] != name.replace(" ", "_"):
            continue
        C = heads * 64
        q = torch.randn((n, Nq, C), dtype=torch.float16, device=dev)
        k = torch.randn((n, Nk, C), dtype=torch.float16, device=dev)
        v = torch.randn((n, Nk, C), dtype=torch.float16, device=dev)
        o = torch.empty_like(q)
        ms = timeit(lambda: lib.ladi_op_attention(ptr(q), ptr(k), ptr(v), ptr(o), C, C, C, C, Nq * C, Nk * C, Nk * C, Nq * C, n, heads, Nq, Nk, 0.125, stream_ptr()))
        print("  %-10s %8.3f ms  %7.1f TF/s" % (name, ms, 4.0 * n * heads * Nq * Nk * 64 / ms / 1e9))
    if "--attn-only" in sys.argv:
        return
    print("group_norm (stats+apply, silu)")
    for name, HW, C in (("L0 320", 3072, 320), ("L0 960", 3072, 960), ("L1 640", 768, 640), ("L2 1280", 192, 1280), ("L2 2560", 192, 2560),
                        ("vae 128@512x384 n=2", 196608, 128), ("vae 512@128x96 n=2", 12288, 512)):
        nn = 2 if name.startswith("vae") else n
        x = torch.randn((nn, HW, C), dtype=torch.float16, device=dev)
        g = torch.ones((C,), dtype=torch.float16, device=dev)
        b = torch.zeros((C,), dtype=torch.float16, device=dev)
        out = torch.empty_like(x)
        stats = torch.empty((nn * 64,), dtype=torch.float32, device=dev)
        ms = timeit(lambda: lib.ladi_op_group_norm(ptr(x), C, None, 0, nn, HW, 32, ptr(g), ptr(b), 1e-5, 1, None, ptr(out), ptr(stats), stream_ptr()))
        byts = x.numel() * 2
        print("  %-22s %8.3f ms  %7.2f TB/s (3 passes of %.1f MB)" % (name, ms, 3 * byts / ms / 1e9, byts / 1e6))
    print("layer_norm")
    for name, rows, C in (("L0", n * 3072, 320), ("L1", n * 768, 640), ("L2", n * 192, 1280)):
        x = torch.randn((rows, C), dtype=torch.float16, device=dev)
        g = torch.ones((C,), dtype=torch.float16, device=dev)
        b = torch.zeros((C,), dtype=torch.float16, device=dev)
        out = torch.empty_like(x)
        ms = timeit(lambda: lib.ladi_op_layer_norm(ptr(x), ptr(g), ptr(b), 1e-5, rows, C, ptr(out), stream_ptr()))
        print("  %-10s %8.3f ms  %7.2f TB/s" % (name, ms, 2 * x.numel() * 2 / ms / 1e9))


if __name__ == "__main__":
    main()
