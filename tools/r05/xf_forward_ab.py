"""Same-process A/B of the CFG UNet forward at the bench batch: fused attn2 / feed-forward blocks off / attn2 only / both, plain vs pipelined
feed-forward loop.  LADI_XF_FUSE is latched when a UNet is LOADED (round 6; it was read per planning pass in round 5), so every arm builds its
own UNet handle in this process.  python tools/r05/xf_forward_ab.py [--iters 10] [--rounds 3]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--modes", default="0,1,2,2p0")
    a = ap.parse_args()
    t0 = time.time()
    import ladi_vton_amd as L
    from ladi_vton_amd import configs as C
    dev = torch.device("cuda", 0)
    ucfg = C.UNET_FULL
    g = torch.Generator().manual_seed(0)
    ehs = torch.randn((a.n, 77, 1024), generator=g).half().to(dev)
    modes = a.modes.split(",")
    unets = {}
    for lvl in sorted({m[0] for m in modes}):
        os.environ["LADI_XF_FUSE"] = lvl
        unets[lvl] = L.NativeUNet(ucfg, C.synth_items(C.unet_shapes(ucfg), "unet."))
        unets[lvl].set_context(ehs.contiguous())
    print("built in %.1fs" % (time.time() - t0), flush=True)
    res = {m: [] for m in modes}
    for r in range(a.rounds + 1):
        for tag in modes:
            unet = unets[tag[0]]
            os.environ["LADI_FF_PIPE"] = "0" if tag.endswith("p0") else "1"
            ms = unet.time_forward(a.n, 64, 48, a.iters if r else 2)
            if r:
                res[tag].append(round(ms, 3))
            print(tag, round(ms, 3), flush=True)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
