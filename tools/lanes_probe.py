"""Sample-group lanes A/B on one process (one model build): CFG UNet forward at the bench batch for G = 1, 2, 4, 8 lanes (hipGraph with
G parallel branches, and eager multi-stream), then the whole BASELINE configs[1] step at the lane counts given.
  python tools/lanes_probe.py [--batch 8] [--iters 10] [--pipe-lanes 1,2,4] [--steps 2]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pipe-lanes", default="1,2,4")
    ap.add_argument("--lanes-list", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--inference-steps", type=int, default=50)
    a = ap.parse_args()
    import ladi_vton_amd as L
    from ladi_vton_amd import configs as C
    import bench
    dev = torch.device("cuda", 0)
    ucfg, vcfg = C.UNET_FULL, C.VAE_FULL
    ecfg = C.emasc_for_vae(vcfg)
    unet = L.NativeUNet(ucfg, C.synth_items(C.unet_shapes(ucfg), "unet."))
    vae = L.NativeVAE(vcfg, C.synth_items(C.vae_shapes(vcfg), "vae."))
    emasc = L.NativeEMASC(ecfg, C.synth_items(C.emasc_shapes(ecfg), "emasc."))
    B, H, W = a.batch, a.height, a.width
    n, h, w = 2 * B, H // 8, W // 8
    local = bench.make_rows(0, B, H, W, 77, 1024, dev)
    unet.set_context(torch.cat([local["negative_prompt_embeds"], local["prompt_embeds"]]).contiguous())
    res = {"n": n, "h": h, "w": w}
    res["single_stream_ms"] = round(unet.time_forward(n, h, w, a.iters), 3)
    for g in [int(v) for v in a.lanes_list.split(",") if v]:
        if n % g:
            continue
        for graph in (True, False):
            try:
                ms = unet.time_forward_lanes(n, h, w, a.iters, g, graph)
                ms2 = unet.time_forward_lanes(n, h, w, a.iters, g, graph)
                res["lanes%d_%s_ms" % (g, "graph" if graph else "eager")] = [round(ms, 3), round(ms2, 3)]
            except Exception as e:
                res["lanes%d_%s_ms" % (g, "graph" if graph else "eager")] = repr(e)
        print(json.dumps(res), flush=True)
    pipe = L.StableDiffusionTryOnePipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, scheduler=L.PNDMScheduler(), emasc=emasc,
                                           emasc_int_layers=[1, 2, 3, 4, 5])
    for g in [int(x) for x in a.pipe_lanes.split(",") if x]:
        pipe.lanes = g
        def step():
            return pipe._run_fused(local["image"], local["mask_image"], local["pose_map"], local["warped_cloth"], local["prompt_embeds"],
                                   local["negative_prompt_embeds"], local["noise_cloth"], local["noise_latents"], local["noise_masked"], H, W,
                                   a.inference_steps, 7.5, 1.0, False, True, return_device=True, out_uint8=True)
        step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.steps
        res["pipeline_lanes%d" % g] = {"ms_per_batch": round(dt * 1e3, 1), "images_per_s": round(B / dt, 3), "lanes_used": pipe.lib_lanes(),
                                      "checksum": int(out.to(torch.int64).sum())}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
