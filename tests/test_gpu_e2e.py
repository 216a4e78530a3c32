"""End-to-end parity ON THE BASELINE CONFIGURATION: the full-size model (released
architecture, deterministic synthetic checkpoint) at 512x384 through the fused hipGraph loop for 50 PNDM (51 evaluations) and
50 DDIM steps at B = 1, against the CPU fp32 oracle on identical fp16-rounded weights / inputs / noise, with a per-evaluation trace.

Stated tolerances: single UNet forward PSNR >= 60 dB (peak = max|ref|) and rel-L2 <= 2e-3; decoded image after
50 steps >= 35 dB on [0,1] images.  Every measured value is also written to gpurun_out/parity_r03.json (copied to profiles/)."""
import json
import os
import time

import pytest
import torch

from oracle import configs as C
from oracle import models as M
from oracle import pipeline as P
from tests import util as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(key, value):
    U.record_parity(key, value)


@pytest.fixture(scope="module")
def full():
    import ladi_vton_amd as L
    torch.set_num_threads(U.cpu_quota_threads())
    ucfg, vcfg, ecfg = C.UNET_FULL, C.VAE_FULL, C.EMASC_FULL
    sd = dict(unet=C.synth_state_dict(C.unet_shapes(ucfg), "unet."), vae=C.synth_state_dict(C.vae_shapes(vcfg), "vae."),
              emasc=C.synth_state_dict(C.emasc_shapes(ecfg), "emasc."))
    mod = dict(unet=L.NativeUNet(ucfg, sd["unet"]), vae=L.NativeVAE(vcfg, sd["vae"]), emasc=L.NativeEMASC(ecfg, sd["emasc"]))
    return dict(ucfg=ucfg, vcfg=vcfg, ecfg=ecfg, sd=sd, mod=mod)


def test_full_unet_single_forward_stated_tolerance(full):
    """noise_pred of ONE full-size CFG evaluation at 64x48: the stated contract is PSNR >= 60 dB and rel-L2 <= 2e-3"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 31, 64, 48), generator=g).half().float()
    ehs = torch.randn((2, 77, 1024), generator=g).half().float()
    vals = {}
    for t in (981, 501, 1):
        ref = M.unet_forward(full["sd"]["unet"], full["ucfg"], x, t, ehs)
        got = full["mod"]["unet"](x.to(U.dev()), t, encoder_hidden_states=ehs.to(U.dev())).sample.float().cpu()
        vals[str(t)] = dict(psnr_db=round(U.psnr(got, ref), 2), rel_l2=U.rel_l2(got, ref))
    _record("unet_forward_full_64x48", vals)
    for t, v in vals.items():
        assert v["psnr_db"] >= 60.0 and v["rel_l2"] <= 2e-3, (t, v)


@pytest.mark.parametrize("sched", ["pndm", "ddim"])
def test_baseline_config_50_steps_vs_oracle(full, sched):
    """BASELINE configs[1] arithmetic at B = 1: 512x384, 50 scheduler steps (PNDM: 51 UNet evaluations), guidance 7.5, EMASC on"""
    import ladi_vton_amd as L
    B, H, W, steps = 1, 512, 384, 50
    inp = P.synthetic_inputs(B, H, W, L=77, D=1024)
    for k in ("prompt_embeds", "negative_prompt_embeds"):
        inp[k] = inp[k].half().float()
    trace = {}
    t0 = time.time()
    ref_img, ref_lat = P.tryon_pipeline(full["sd"]["unet"], full["ucfg"], full["sd"]["vae"], full["vcfg"], full["sd"]["emasc"], inp,
                                        num_inference_steps=steps, guidance_scale=7.5, scheduler=sched, trace=trace)
    cpu_s = time.time() - t0
    sch = L.DDIMScheduler() if sched == "ddim" else L.PNDMScheduler()
    pipe = L.StableDiffusionTryOnePipeline(vae=full["mod"]["vae"], text_encoder=None, tokenizer=None, unet=full["mod"]["unet"], scheduler=sch,
                                           emasc=full["mod"]["emasc"], emasc_int_layers=[1, 2, 3, 4, 5])
    evals = steps + 1 if sched == "pndm" else steps
    pipe.trace_evals = evals
    d = U.dev()
    out = pipe(image=inp["image"].to(d), mask_image=inp["mask_image"].clone().to(d), pose_map=inp["pose_map"].to(d),
               warped_cloth=inp["warped_cloth"].to(d), prompt_embeds=inp["prompt_embeds"].to(d),
               negative_prompt_embeds=inp["negative_prompt_embeds"].to(d), height=H, width=W, num_inference_steps=steps,
               guidance_scale=7.5, output_type="np", fused=True, use_graph=True,
               noise=(inp["noise_cloth"], inp["noise_latents"], inp["noise_masked"]))
    img = torch.from_numpy(out.images)
    lat = pipe.last_latents.float().cpu()
    tr = {k: v.cpu() for k, v in pipe.last_trace.items()}
    assert len(trace["noise_pred"]) == evals and tr["noise_pred"].shape[0] == evals
    eps_psnr = [round(U.psnr(tr["noise_pred"][i], trace["noise_pred"][i]), 2) for i in range(evals)]
    lat_psnr = [round(U.psnr(tr["latents"][i], trace["latents"][i]), 2) for i in range(evals)]
    u8a, u8b = (img * 255).round(), (ref_img * 255).round()
    res = dict(evals=evals, image_psnr_db=round(U.psnr(img, ref_img, 1.0), 2), final_latents_psnr_db=round(U.psnr(lat, ref_lat), 2),
               uint8_psnr_db=round(U.psnr(u8a, u8b, 255.0), 2), uint8_max_abs_diff=int((u8a - u8b).abs().max()),
               noise_pred_psnr_db_per_eval=eps_psnr, latents_psnr_db_per_eval=lat_psnr,
               noise_pred_psnr_db_min=min(eps_psnr), latents_psnr_db_min=min(lat_psnr), cpu_oracle_seconds=round(cpu_s, 1),
               cpu_threads=torch.get_num_threads())
    _record("tryon_512x384_50_%s_B1" % sched, res)
    assert img.shape == ref_img.shape == (B, H, W, 3)
    assert torch.equal(tr["latents"][-1], lat)                                     # the trace really is this run's trajectory
    # SURVEY.md §8d contract: full pipeline after 50 steps >= 35 dB.  Measured on MI355X (profiles/r02_parity.json, r03_parity.json): image
    # 63.5-64.3 dB, final latents 66.3-66.9 dB, guided noise_pred >= 55.4 dB at every evaluation, uint8 max abs diff 1 -> regression guards
    # 3-4 dB under the measured values:
    assert res["image_psnr_db"] >= 60.0, res
    assert res["final_latents_psnr_db"] >= 62.0 and res["noise_pred_psnr_db_min"] >= 52.0, res
    assert res["uint8_max_abs_diff"] <= 2, res
    assert eps_psnr[0] >= 55.0, eps_psnr[:3]                                       # first evaluation: no accumulated trajectory error yet


def test_config4_resolution_short_run_vs_oracle(full):
    """BASELINE configs[4] shapes (1024x768: 128x96 latents, 12 288-token self-attention, 12 288-token single-head VAE attention through the
    wide flash kernel, 1024x768 EMASC skips) at B = 1 with 4 DDIM steps -- the full 100-step run costs the CPU oracle ~20 min, the shapes
    are what this test is about.  Same contract as the 512x384 runs."""
    import ladi_vton_amd as L
    # LADI_TEST_CONFIG4_STEPS lengthens the compared trajectory (40 of the configuration's 100 DDIM steps were run once this way and are
    # recorded in profiles/r03_parity.json; the CPU oracle needs ~8.5 s per evaluation at this resolution)
    B, H, W, steps = 1, 1024, 768, int(os.environ.get("LADI_TEST_CONFIG4_STEPS", "4"))
    inp = P.synthetic_inputs(B, H, W, L=77, D=1024)
    for k in ("prompt_embeds", "negative_prompt_embeds"):
        inp[k] = inp[k].half().float()
    trace = {}
    t0 = time.time()
    ref_img, ref_lat = P.tryon_pipeline(full["sd"]["unet"], full["ucfg"], full["sd"]["vae"], full["vcfg"], full["sd"]["emasc"], inp,
                                        num_inference_steps=steps, guidance_scale=7.5, scheduler="ddim", trace=trace)
    cpu_s = time.time() - t0
    pipe = L.StableDiffusionTryOnePipeline(vae=full["mod"]["vae"], text_encoder=None, tokenizer=None, unet=full["mod"]["unet"],
                                           scheduler=L.DDIMScheduler(), emasc=full["mod"]["emasc"], emasc_int_layers=[1, 2, 3, 4, 5])
    pipe.trace_evals = steps
    d = U.dev()
    out = pipe(image=inp["image"].to(d), mask_image=inp["mask_image"].clone().to(d), pose_map=inp["pose_map"].to(d),
               warped_cloth=inp["warped_cloth"].to(d), prompt_embeds=inp["prompt_embeds"].to(d),
               negative_prompt_embeds=inp["negative_prompt_embeds"].to(d), height=H, width=W, num_inference_steps=steps,
               guidance_scale=7.5, output_type="np", fused=True, use_graph=True,
               noise=(inp["noise_cloth"], inp["noise_latents"], inp["noise_masked"]))
    img = torch.from_numpy(out.images)
    lat = pipe.last_latents.float().cpu()
    tr = {k: v.cpu() for k, v in pipe.last_trace.items()}
    eps_psnr = [round(U.psnr(tr["noise_pred"][i], trace["noise_pred"][i]), 2) for i in range(steps)]
    u8a, u8b = (img * 255).round(), (ref_img * 255).round()
    res = dict(evals=steps, image_psnr_db=round(U.psnr(img, ref_img, 1.0), 2), final_latents_psnr_db=round(U.psnr(lat, ref_lat), 2),
               uint8_max_abs_diff=int((u8a - u8b).abs().max()), noise_pred_psnr_db_per_eval=eps_psnr, cpu_oracle_seconds=round(cpu_s, 1))
    _record("tryon_1024x768_%d_ddim_B1" % steps, res)
    assert img.shape == ref_img.shape == (B, H, W, 3)
    # measured (profiles/r02_parity.json): image 58.2 dB, final latents 61.2 dB, guided noise_pred >= 55.0 dB at every evaluation
    assert res["image_psnr_db"] >= 55.0 and res["final_latents_psnr_db"] >= 58.0 and min(eps_psnr) >= 52.0, res


def test_baseline_batch8_matches_single_sample_runs(full):
    """BASELINE configs[1] AT ITS BATCH SIZE (B = 8, 512x384, 50 PNDM steps): the tile selections of the bench batch (the 320x256 / 256x256
    8-wave tiles, split-K choices, X-stationary linears with fused LayerNorm) differ from those of a B = 1 call, and the CPU oracle at B = 8
    would take ~13 min.  Every sample of the batch is therefore compared with the same sample run alone (B = 1: the configuration measured
    against the oracle above at 64 dB): a sample's trajectory may not depend on its batch, up to fp16 rounding of different tile shapes."""
    import ladi_vton_amd as L
    B, H, W, steps = 8, 512, 384, 50
    inp = P.synthetic_inputs(B, H, W, L=77, D=1024)
    d = U.dev()
    pipe = L.StableDiffusionTryOnePipeline(vae=full["mod"]["vae"], text_encoder=None, tokenizer=None, unet=full["mod"]["unet"],
                                           scheduler=L.PNDMScheduler(), emasc=full["mod"]["emasc"], emasc_int_layers=[1, 2, 3, 4, 5])

    def run(lo, hi):
        out = pipe(image=inp["image"][lo:hi].to(d), mask_image=inp["mask_image"][lo:hi].clone().to(d), pose_map=inp["pose_map"][lo:hi].to(d),
                   warped_cloth=inp["warped_cloth"][lo:hi].to(d), prompt_embeds=inp["prompt_embeds"][lo:hi].half().to(d),
                   negative_prompt_embeds=inp["negative_prompt_embeds"][lo:hi].half().to(d), height=H, width=W, num_inference_steps=steps,
                   guidance_scale=7.5, output_type="np", fused=True, use_graph=True,
                   noise=(inp["noise_cloth"][lo:hi], inp["noise_latents"][lo:hi], inp["noise_masked"][lo:hi]))
        return torch.from_numpy(out.images), pipe.last_latents.float().cpu()

    img8, lat8 = run(0, B)
    vals = []
    for i in (0, 3, 7):
        img1, lat1 = run(i, i + 1)
        vals.append(dict(sample=i, image_psnr_db=round(U.psnr(img8[i:i + 1], img1, 1.0), 2), latents_psnr_db=round(U.psnr(lat8[i:i + 1], lat1), 2),
                         uint8_max_abs_diff=int(((img8[i:i + 1] * 255).round() - (img1 * 255).round()).abs().max())))
    _record("tryon_512x384_50_pndm_B8_vs_B1", vals)
    for v in vals:      # measured 62.2-62.4 dB / 64.7-65.6 dB (two fp16 trajectories of different tile shapes against each other)
        assert v["image_psnr_db"] >= 58.0 and v["latents_psnr_db"] >= 60.0 and v["uint8_max_abs_diff"] <= 2, vals


def test_config2_batch32_matches_single_sample_runs(full):
    """BASELINE configs[2] batch (B = 32, 512x384, 50 PNDM steps): at n = 64 CFG samples the tuner picks other tiles than at B = 8 (full
    grids: the 256x256 8-wave pipeline carries most convolutions).  Samples 0, 13 and 31 of the batch are compared with the same samples
    run alone -- the configuration pinned to the oracle by test_baseline_config_50_steps_vs_oracle -- with the B = 8 thresholds."""
    import ladi_vton_amd as L
    B, H, W, steps = 32, 512, 384, 50
    inp = P.synthetic_inputs(B, H, W, L=77, D=1024)
    d = U.dev()
    pipe = L.StableDiffusionTryOnePipeline(vae=full["mod"]["vae"], text_encoder=None, tokenizer=None, unet=full["mod"]["unet"],
                                           scheduler=L.PNDMScheduler(), emasc=full["mod"]["emasc"], emasc_int_layers=[1, 2, 3, 4, 5])

    def run(lo, hi):
        out = pipe(image=inp["image"][lo:hi].to(d), mask_image=inp["mask_image"][lo:hi].clone().to(d), pose_map=inp["pose_map"][lo:hi].to(d),
                   warped_cloth=inp["warped_cloth"][lo:hi].to(d), prompt_embeds=inp["prompt_embeds"][lo:hi].half().to(d),
                   negative_prompt_embeds=inp["negative_prompt_embeds"][lo:hi].half().to(d), height=H, width=W, num_inference_steps=steps,
                   guidance_scale=7.5, output_type="np", fused=True, use_graph=True,
                   noise=(inp["noise_cloth"][lo:hi], inp["noise_latents"][lo:hi], inp["noise_masked"][lo:hi]))
        return torch.from_numpy(out.images), pipe.last_latents.float().cpu()

    img32, lat32 = run(0, B)
    vals = []
    for i in (0, 13, 31):
        img1, lat1 = run(i, i + 1)
        vals.append(dict(sample=i, image_psnr_db=round(U.psnr(img32[i:i + 1], img1, 1.0), 2), latents_psnr_db=round(U.psnr(lat32[i:i + 1], lat1), 2),
                         uint8_max_abs_diff=int(((img32[i:i + 1] * 255).round() - (img1 * 255).round()).abs().max())))
    _record("tryon_512x384_50_pndm_B32_vs_B1", vals)
    for v in vals:
        assert v["image_psnr_db"] >= 58.0 and v["latents_psnr_db"] >= 60.0 and v["uint8_max_abs_diff"] <= 2, vals


def test_batched_launch_vs_oracle_directly(full):
    """A BATCHED call against the oracle itself (the B = 8 / B = 32 tests above compare the HIP path with its own single-sample runs):
    B = 2, 512x384, 20 PNDM steps (21 evaluations; the oracle needs ~90 s for it), guidance 7.5, EMASC on; both samples, same thresholds
    as the 50-step B = 1 run."""
    import ladi_vton_amd as L
    B, H, W, steps = 2, 512, 384, 20
    inp = P.synthetic_inputs(B, H, W, L=77, D=1024)
    for k in ("prompt_embeds", "negative_prompt_embeds"):
        inp[k] = inp[k].half().float()
    t0 = time.time()
    ref_img, ref_lat = P.tryon_pipeline(full["sd"]["unet"], full["ucfg"], full["sd"]["vae"], full["vcfg"], full["sd"]["emasc"], inp,
                                        num_inference_steps=steps, guidance_scale=7.5, scheduler="pndm")
    cpu_s = time.time() - t0
    pipe = L.StableDiffusionTryOnePipeline(vae=full["mod"]["vae"], text_encoder=None, tokenizer=None, unet=full["mod"]["unet"],
                                           scheduler=L.PNDMScheduler(), emasc=full["mod"]["emasc"], emasc_int_layers=[1, 2, 3, 4, 5])
    d = U.dev()
    out = pipe(image=inp["image"].to(d), mask_image=inp["mask_image"].clone().to(d), pose_map=inp["pose_map"].to(d),
               warped_cloth=inp["warped_cloth"].to(d), prompt_embeds=inp["prompt_embeds"].to(d),
               negative_prompt_embeds=inp["negative_prompt_embeds"].to(d), height=H, width=W, num_inference_steps=steps,
               guidance_scale=7.5, output_type="np", fused=True, use_graph=True,
               noise=(inp["noise_cloth"], inp["noise_latents"], inp["noise_masked"]))
    img = torch.from_numpy(out.images)
    lat = pipe.last_latents.float().cpu()
    vals = [dict(sample=i, image_psnr_db=round(U.psnr(img[i:i + 1], ref_img[i:i + 1], 1.0), 2),
                 latents_psnr_db=round(U.psnr(lat[i:i + 1], ref_lat[i:i + 1]), 2),
                 uint8_max_abs_diff=int(((img[i:i + 1] * 255).round() - (ref_img[i:i + 1] * 255).round()).abs().max())) for i in range(B)]
    _record("tryon_512x384_20_pndm_B2_vs_oracle", dict(samples=vals, cpu_oracle_seconds=round(cpu_s, 1)))
    for v in vals:
        assert v["image_psnr_db"] >= 60.0 and v["latents_psnr_db"] >= 62.0 and v["uint8_max_abs_diff"] <= 2, vals
