"""Same-process A/B of the CFG UNet forward at the bench batch under different environment settings (every LADI_* switch that is read per
planning pass / per launch).  python tools/r05/forward_ab.py --modes "base:;old:LADI_GN_ONEPASS=0;p128:LADI_GN_PPB=128" [--iters 10] [--rounds 3]"""
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
    ap.add_argument("--hw", default="64x48")
    ap.add_argument("--modes", required=True)
    a = ap.parse_args()
    t0 = time.time()
    import ladi_vton_amd as L
    from ladi_vton_amd import configs as C
    dev = torch.device("cuda", 0)
    ucfg = C.UNET_FULL
    unet = L.NativeUNet(ucfg, C.synth_items(C.unet_shapes(ucfg), "unet."))
    g = torch.Generator().manual_seed(0)
    ehs = torch.randn((a.n, 77, 1024), generator=g).half().to(dev)
    unet.set_context(ehs.contiguous())
    print("built in %.1fs" % (time.time() - t0), flush=True)
    h, w = (int(v) for v in a.hw.split("x"))
    modes = []
    for m in a.modes.split(";"):
        name, _, kv = m.partition(":")
        modes.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
    keys = sorted({k for _, e in modes for k in e})
    res = {m: [] for m, _ in modes}
    for r in range(a.rounds + 1):
        for name, env in modes:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            ms = unet.time_forward(a.n, h, w, a.iters if r else 2)
            if r:
                res[name].append(round(ms, 3))
            print(name, round(ms, 3), flush=True)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
