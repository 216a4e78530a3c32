"""Same-process sweep of the GroupNorm launch geometry (tools/experiments/next/gn_grid_sweep.patch applied and built): CFG UNet forward at
the bench batch for several (smallest block, target grid) pairs of norm.hip pick_ppb -- gn_apply and gn_partial share it.
python tools/experiments/next/gn_grid_ab.py [--iters 10] [--rounds 3]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
MODES = (("48", "768"), ("24", "1024"), ("12", "1024"), ("12", "2048"), ("6", "2048"), ("4", "4096"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--n", type=int, default=16)
    a = ap.parse_args()
    t0 = time.time()
    import ladi_vton_amd as L
    from ladi_vton_amd import configs as C
    dev = torch.device("cuda", 0)
    unet = L.NativeUNet(C.UNET_FULL, C.synth_items(C.unet_shapes(C.UNET_FULL), "unet."))
    g = torch.Generator().manual_seed(0)
    unet.set_context(torch.randn((a.n, 77, 1024), generator=g).half().to(dev).contiguous())
    print("built in %.1fs" % (time.time() - t0), flush=True)
    res = {"%s/%s" % m: [] for m in MODES}
    for r in range(a.rounds + 1):
        for m in MODES:
            os.environ["LADI_GN_PPB_FLOOR"], os.environ["LADI_GN_BLOCKS"] = m
            ms = unet.time_forward(a.n, 64, 48, a.iters if r else 2)
            if r:
                res["%s/%s" % m].append(round(ms, 3))
            print(m, round(ms, 3), flush=True)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
