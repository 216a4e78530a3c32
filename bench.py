#!/usr/bin/env python
"""bench.py — try-on images/s of the native LaDI-VTON hot path (BASELINE.json metric).

One "step" = one full pass of the hot path over one batch: B try-on pairs in (device resident) -> VAE encodes + EMASC ->
50 scheduler steps of the CFG UNet -> VAE decode with EMASC skips -> uint8 images (all-gathered over ranks when N > 1).
Workload = BASELINE.json configs[1]: batch 8 per GPU, 50 steps, 512x384, fp16 storage / fp32 accumulate, synthetic inputs and a
deterministic random-init checkpoint of the released architecture (no weights or datasets are reachable offline).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work (SURVEY.md §8d / BASELINE.md §2), 2*MAC FLOPs
UNET_FLOP_PER_SAMPLE_64x48 = 581.70e9
TRYON_FLOP_PER_IMAGE = {"ddim": 62.16e12, "pndm": 63.32e12}
PEAK_F16_TFLOPS = 2500.0   # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=8, help="try-on pairs per GPU (weak scaling)")
    p.add_argument("--inference-steps", type=int, default=50)
    p.add_argument("--scheduler", default="pndm", choices=["pndm", "ddim"])
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=384)
    p.add_argument("--size", default="full", choices=["full", "tiny"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--roofline-only", action="store_true",
                   help="only the dominant-kernel measurement: CFG UNet forwards at the bench batch (the command the rocprofv3 "
                        "summaries under profiles/ are taken from)")
    p.add_argument("--roofline-iters", type=int, default=4)
    return p.parse_args()


def synthetic_device_inputs(B, H, W, L, D, device, seed):
    """synthetic inputs with the datasets' shapes and value ranges, generated per rank on the device"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    F = torch.nn.functional

    def smooth():
        low = torch.rand((B, 3, H // 8, W // 8), generator=g) * 2 - 1
        return F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False).clamp(-1, 1)

    image, cloth = smooth(), smooth()
    mask = torch.zeros(B, 1, H, W)
    mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
    ys = torch.arange(H, dtype=torch.float32)[None, None, :, None]
    xs = torch.arange(W, dtype=torch.float32)[None, None, None, :]
    cy = torch.rand((B, 18, 1, 1), generator=g) * H
    cx = torch.rand((B, 18, 1, 1), generator=g) * W
    pose = torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 81.0)
    pose[:, 7::9] = 0.0
    h, w = H // 8, W // 8
    d = dict(image=image, mask_image=mask, pose_map=pose, warped_cloth=cloth,
             prompt_embeds=torch.randn((B, L, D), generator=g), negative_prompt_embeds=torch.randn((1, L, D), generator=g).expand(B, L, D).contiguous(),
             noise_cloth=torch.randn((B, 4, h, w), generator=g), noise_latents=torch.randn((B, 4, h, w), generator=g),
             noise_masked=torch.randn((B, 4, h, w), generator=g))
    out = {}
    for k, v in d.items():
        if k.startswith("noise"):
            out[k] = v.to(device)
        else:
            out[k] = v.to(device=device, dtype=torch.float16)
    return out


def pmc_traffic(kernel_label):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r01_pmc_*.txt: separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM). None if absent."""
    import re
    m = re.match(r"igemm_kernel<([0-9,]+)>", kernel_label)
    if m:
        pat = "igemm_kernel<" + ", ".join(m.group(1).split(",")) + ">"
    elif kernel_label.startswith("linear_xs_kernel"):
        pat = "linear_xs_kernel"     # first (= most time-consuming) instantiation listed in the PMC summary
    else:
        return None
    vals = {}
    for tag in ("fetch", "write"):
        path = os.path.join(ROOT, "profiles", "r01_pmc_%s_size.txt" % tag)
        if not os.path.exists(path):
            return None
        for line in open(path):
            if pat in line:
                f = line.split()
                vals[tag] = float(f[-2]) * 1024.0   # avg KiB per dispatch -> bytes
                break
    if len(vals) != 2:
        return None
    return {"bytes_per_launch": round(2.0 * vals["fetch"] + vals["write"]), "fetch_bytes_x2": round(2.0 * vals["fetch"]),
            "write_bytes": round(vals["write"]), "source": "profiles/r01_pmc_{fetch,write}_size.txt (avg over all launches of this kernel)"}


def cpu_baseline(sds, cfgs, H, W, evals, L, D):
    """The fp32 CPU oracle (a port: the reference itself is not importable) timed on this box's host cores on a
    bounded sample of the same workload: ONE CFG UNet evaluation (n=2) + VAE encode + EMASC + VAE decode for one 512x384 image;
    images/s extrapolated as 1 / (evals * t_unet + 2 t_enc + t_emasc + t_dec)."""
    from oracle import models as M  # test infrastructure: only used as the reported CPU baseline
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # respect the container's cgroup CPU quota (the GPU box shows 256 logical CPUs but grants 16)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    h, w = H // 8, W // 8
    with torch.no_grad():
        x = torch.randn((2, 31, h, w), generator=g)
        ehs = torch.randn((2, L, D), generator=g)
        t0 = time.time(); M.unet_forward(sds["unet"], cfgs["unet"], x, 481, ehs); t_unet = time.time() - t0
        img = torch.rand((1, 3, H, W), generator=g) * 2 - 1
        t0 = time.time(); mom, feats = M.vae_encode(sds["vae"], cfgs["vae"], img); t_enc = time.time() - t0
        t0 = time.time(); sk = M.emasc_forward(sds["emasc"], feats[1:6]); t_em = time.time() - t0
        z = torch.randn((1, 4, h, w), generator=g)
        t0 = time.time(); M.vae_decode(sds["vae"], cfgs["vae"], z, sk, [1, 2, 3, 4, 5]); t_dec = time.time() - t0
    per_image = evals * t_unet + 2 * t_enc + t_em + t_dec
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "fp32 torch-CPU oracle, full-size model: 1 CFG UNet eval (n=2, 64x48) %.2fs + VAE encode %.2fs + EMASC %.2fs + VAE decode %.2fs "
                      "for one 512x384 image; extrapolated to %d evals + 2 encodes + EMASC + decode" % (t_unet, t_enc, t_em, t_dec, evals)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the native path has no CPU fallback"
    torch.cuda.set_device(local_rank)          # bind the rank to its GPU before RCCL is initialised
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import ladi_vton_amd as L
    from ladi_vton_amd import _lib, configs as C
    from ladi_vton_amd.parallel import all_gather_images, to_uint8
    ucfg, vcfg = (C.UNET_FULL, C.VAE_FULL) if a.size == "full" else (C.UNET_TINY, C.VAE_TINY)
    ecfg = C.emasc_for_vae(vcfg)
    cfgs = dict(unet=ucfg, vae=vcfg, emasc=ecfg)
    t_build = time.time()
    want_cpu = (rank == 0 and world == 1 and not a.no_cpu_baseline and not a.roofline_only)
    if want_cpu:   # the CPU baseline needs the fp32 checkpoint on the host; otherwise stream it tensor by tensor
        sds = dict(unet=C.synth_state_dict(C.unet_shapes(ucfg), "unet."), vae=C.synth_state_dict(C.vae_shapes(vcfg), "vae."),
                   emasc=C.synth_state_dict(C.emasc_shapes(ecfg), "emasc."))
        unet, vae, emasc = L.NativeUNet(ucfg, sds["unet"]), L.NativeVAE(vcfg, sds["vae"]), L.NativeEMASC(ecfg, sds["emasc"])
    else:
        sds = None
        unet = L.NativeUNet(ucfg, C.synth_items(C.unet_shapes(ucfg), "unet."))
        vae = L.NativeVAE(vcfg, C.synth_items(C.vae_shapes(vcfg), "vae."))
        emasc = L.NativeEMASC(ecfg, C.synth_items(C.emasc_shapes(ecfg), "emasc."))
    sch = L.DDIMScheduler() if a.scheduler == "ddim" else L.PNDMScheduler()
    pipe = L.StableDiffusionTryOnePipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, scheduler=sch, emasc=emasc,
                                           emasc_int_layers=[1, 2, 3, 4, 5])
    t_build = time.time() - t_build
    B, H, W = a.batch, a.height, a.width
    Ltok, D = 77, ucfg["cross_attention_dim"]
    inp = synthetic_device_inputs(B, H, W, Ltok, D, dev, seed=1234 + rank)
    global_B = B * world

    def one_step():
        imgs = pipe._run_fused(inp["image"], inp["mask_image"], inp["pose_map"], inp["warped_cloth"], inp["prompt_embeds"],
                               inp["negative_prompt_embeds"], inp["noise_cloth"], inp["noise_latents"], inp["noise_masked"], H, W,
                               a.inference_steps, 7.5, 1.0, False, not a.no_graph, return_device=True)
        u8 = to_uint8(imgs)
        return all_gather_images(u8, global_B)   # the path's only collective (RCCL all-gather of decoded images)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.roofline_only:
        # context for the stand-alone UNet forwards; one untimed forward first (per-shape tile measurement happens there)
        e = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]]).contiguous()
        unet.set_context(e)
        unet.time_forward(2 * B, H // 8, W // 8, 1)
        out = torch.zeros((global_B, H, W, 3), dtype=torch.uint8, device=dev)
        a.warmup, a.steps = 0, 0
    for _ in range(a.warmup):
        out = one_step()
    fence()
    t0 = time.time()
    for _ in range(a.steps):
        out = one_step()
    fence()
    dt = max(time.time() - t0, 1e-9)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    assert out.shape[0] == global_B and out.dtype == torch.uint8
    images_per_s = global_B * a.steps / dt if a.steps else 0.0
    evals = a.inference_steps + (1 if a.scheduler == "pndm" else 0)
    lib = _lib.load()
    stage = (ctypes.c_float * 3)()
    stage_ms = list(stage) if (pipe._tryon and lib.ladi_tryon_stage_ms(pipe._tryon, stage) == 0) else None
    if stage_ms is not None:
        stage_ms = [float(stage[i]) for i in range(3)]

    roofline = None
    if rank == 0 and not a.no_roofline:
        # dominant kernel = the MFMA implicit-GEMM family (conv3x3 / conv1x1 / linear = 88% of UNet FLOPs, UNet = 94% of the path).
        # Per-launch HIP events on the launch stream around every igemm launch of CFG UNet forwards at the bench batch.
        n = 2 * B
        h, w = H // 8, W // 8
        whole_ms = unet.time_forward(n, h, w, a.roofline_iters)
        lib.ladi_profile_igemm_enable(1)
        unet.time_forward(n, h, w, a.roofline_iters)   # 1 warm-up + roofline_iters timed forwards, all recorded
        lib.ladi_profile_igemm_enable(0)
        prof = (ctypes.c_double * 192)()
        lib.ladi_profile_igemm_collect(prof, 192)
        names = {1: "igemm_kernel<2,2,2,4,32,3> (Q128xP256)", 2: "igemm_kernel<2,2,5,2,32,2> (Q320xP128)", 3: "igemm_kernel<2,2,2,2,32,3> (Q128xP128)",
                 4: "igemm_kernel<2,2,2,1,32,3> (Q128xP64)", 5: "igemm_kernel<2,2,1,1,32,3> (Q64xP64)", 6: "igemm_kernel<2,2,4,2,32,3> (Q256xP128)"}
        per = {}
        names.update({7: "igemm_kernel<2,2,2,2,64,2> (Q128xP128 BK64)", 8: "igemm_kernel<2,2,2,4,64,2> (Q128xP256 BK64)", 9: "igemm_kernel<2,2,2,1,64,3> (Q128xP64 BK64)",
                      10: "igemm_kernel<2,2,5,2,64,2> (Q320xP128 BK64)"})
        names.update({11: "igemm_kernel<2,2,2,1,64,3> + split-K 2", 12: "igemm_kernel<2,2,2,1,64,3> + split-K 4", 13: "igemm_kernel<2,2,2,1,64,3> + split-K 8",
                      14: "igemm_kernel<2,2,2,2,64,2> + split-K 2", 15: "igemm_kernel<2,2,2,2,64,2> + split-K 4"})
        names.update({16: "igemm_kernel<2,2,1,1,32,4> (Q64xP64 NST4)", 17: "igemm_kernel<2,2,2,1,32,4> (Q128xP64 NST4)", 18: "igemm_kernel<2,2,2,2,32,4> (Q128xP128 NST4)",
                      19: "igemm_kernel<2,4,2,2,32,3> (Q128xP256, 8 waves)", 20: "igemm_kernel<4,2,2,2,32,3> (Q256xP128, 8 waves)",
                      21: "igemm_kernel<2,4,4,2,32,3> (Q256xP256, 8 waves)", 22: "igemm_kernel<2,4,5,2,64,2> (Q320xP256, 8 waves)"})
        names.update({23: "linear_xs_kernel (64 px/wave, 1 channel slice)", 24: "linear_xs_kernel (64 px/wave, 2 channel slices)",
                      25: "linear_xs_kernel (32 px/wave, 1 channel slice)", 26: "linear_xs_kernel (32 px/wave, 2 channel slices)",
                      27: "linear_xs_kernel (32 px/wave, 5 channel slices)"})
        names.update({28: "igemm_kernel<4,2,2,2,32,3> + split-K 4", 29: "igemm_kernel<4,2,2,2,32,3> + split-K 8",
                      30: "igemm_kernel<2,4,2,2,32,3> + split-K 4", 31: "igemm_kernel<2,4,2,2,32,3> + split-K 8"})
        names.update({32: "igemm8_kernel<5,2> (Q320xP256, phase-staggered)", 33: "igemm8_kernel<4,2> (Q256xP256, phase-staggered)",
                      34: "igemm8_kernel<5,2> + split-K 2", 35: "igemm8_kernel<5,2> + split-K 4", 36: "igemm8_kernel<4,2> + split-K 2",
                      37: "igemm8_kernel<4,2> + split-K 4", 38: "igemm8_kernel<4,2> + split-K 8"})
        for c_ in range(1, 39):
            ms, fl, cnt = prof[c_ * 3], prof[c_ * 3 + 1], prof[c_ * 3 + 2]
            if cnt > 0:
                per[c_] = dict(kernel=names[c_], launches=int(cnt), avg_ms=ms / cnt, flop_per_launch=fl / cnt, tflops=fl / ms / 1e9)
        dom = max(per, key=lambda k: per[k]["avg_ms"] * per[k]["launches"]) if per else None
        if dom:
            ach = per[dom]["tflops"]
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_TFLOPS, 4),
                        "traffic": pmc_traffic(per[dom]["kernel"]), "kernel": per[dom]["kernel"], "avg_launch_ms": round(per[dom]["avg_ms"], 5),
                        "flop_per_launch": per[dom]["flop_per_launch"], "launches_profiled": per[dom]["launches"],
                        "igemm_all_tflops": round(prof[1] / prof[0] / 1e9, 2) if prof[0] > 0 else None,
                        "per_config": {per[k]["kernel"]: {"tflops": round(per[k]["tflops"], 1), "launches": per[k]["launches"],
                                                          "avg_ms": round(per[k]["avg_ms"], 5)} for k in per},
                        "unet_forward_ms": round(whole_ms, 3),
                        "unet_forward_tflops": round(UNET_FLOP_PER_SAMPLE_64x48 * n * (h * w / 3072.0) / whole_ms / 1e9, 2) if a.size == "full" else None,
                        "whole_path_frac": round(images_per_s / world * TRYON_FLOP_PER_IMAGE[a.scheduler] / 1e12 / PEAK_F16_TFLOPS, 4) if a.size == "full" and (H, W) == (512, 384) and a.inference_steps == 50 else None}
    cpu = None
    if want_cpu:
        try:
            cpu = cpu_baseline(sds, cfgs, H, W, evals, Ltok, D)
        except Exception as e:  # the baseline is reported, never required
            cpu = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        line = {
            "metric": "try-on images/sec @%dx%d, %d %s steps" % (H, W, a.inference_steps, a.scheduler.upper()), "value": round(images_per_s, 4), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1000.0, 2) if a.steps else None, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: VITON-HD-paired-like, batch %d per GPU, %d %s steps (%d UNet evals, CFG 7.5), %dx%d, "
                                   "EMASC skips on, fp16 storage / fp32 accumulate, %s-size random-init checkpoint" % (B, a.inference_steps, a.scheduler.upper(), evals, H, W, a.size),
                       "global_batch": global_B, "parallelism": "dp%d (batch sharding + RCCL all-gather of uint8 images)" % world,
                       "hipgraph": not a.no_graph},
            "stage_ms_rank0": stage_ms, "model_build_s": round(t_build, 1),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
