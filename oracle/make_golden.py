"""Generate the golden fixtures under tests/golden/ by running the REAL reference code where it is importable in the build
container: src/models/emasc.py EMASC, src/utils/data_utils.py mask_features, and the installed
transformers CLIPEncoderLayer (the class src/models/inversion_adapter.py:2,9 wraps; v5 call convention `layer(x, None)`), and
src/utils/encode_text_word_embedding.py on the installed transformers CLIPTextModel (through a 4.27-layout facade).

Run here (needs /root/reference):  python -m oracle.make_golden
The GPU box has no /root/reference: tests only read the committed .safetensors files.
"""
import os
import sys

import numpy as np
import torch
from safetensors.torch import save_file

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.path.insert(0, REF)
    from src.models.emasc import EMASC  # noqa: E402  (reference code, imported not copied)
    from src.utils.data_utils import mask_features  # noqa: E402
    from ladi_vton_amd import configs as C

    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # ---- EMASC + mask_features (tiny channel config, 64x48 base resolution)
    ecfg = C.EMASC_TINY
    sd = C.synth_state_dict(C.emasc_shapes(ecfg), "emasc.")
    m = EMASC(list(ecfg["in_channels"]), list(ecfg["out_channels"]), kernel_size=3, padding=1, stride=1, type="nonlinear").eval()
    m.load_state_dict(sd, strict=True)
    B, H, W = 1, 32, 24
    g = torch.Generator().manual_seed(11)
    sizes = [(H, W), (H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    feats = [torch.randn((B, c, h, w), generator=g).half().float() for c, (h, w) in zip(ecfg["in_channels"], sizes)]
    mask = (torch.rand((B, 1, H, W), generator=g) > 0.6).float()
    with torch.no_grad():
        outs = m([f.clone() for f in feats])
        masked = mask_features([o.clone() for o in outs], mask.clone())
    blob = {}
    for i in range(5):
        blob["feat%d" % i] = feats[i].contiguous()
        blob["emasc%d" % i] = outs[i].contiguous()
        blob["masked%d" % i] = masked[i].contiguous()
    blob["mask"] = mask
    save_file(blob, os.path.join(OUT, "emasc_tiny.safetensors"))

    # ---- CLIPEncoderLayer (adapter's encoder layer), tiny vision config
    from transformers import CLIPVisionConfig
    from transformers.models.clip.modeling_clip import CLIPEncoderLayer
    acfg = C.ADAPTER_TINY
    vc = CLIPVisionConfig(hidden_size=acfg["hidden"], intermediate_size=acfg["mlp_dim"], num_attention_heads=acfg["heads"],
                          layer_norm_eps=acfg["layer_norm_eps"], hidden_act="gelu", attention_dropout=0.0)
    try:
        vc._attn_implementation = "eager"
    except Exception:
        pass
    layer = CLIPEncoderLayer(vc).eval()
    asd = C.synth_state_dict(C.adapter_shapes(acfg), "adapter.")
    lsd = {k[len("encoder_layers.0."):]: v for k, v in asd.items() if k.startswith("encoder_layers.0.")}
    layer.load_state_dict(lsd, strict=True)
    x = torch.randn((2, 17, acfg["hidden"]), generator=g).half().float()
    with torch.no_grad():
        y = layer(x, None)
        if isinstance(y, (tuple, list)):
            y = y[0]
    save_file({"x": x, "y": y.contiguous()}, os.path.join(OUT, "clip_encoder_layer_tiny.safetensors"))

    make_text_golden(g)
    make_vision_golden(g)
    make_warp_golden()
    make_warp_composed_golden()
    make_wiring_golden()
    make_wiring_lms_golden()
    make_dataset_golden()
    print("wrote", os.listdir(OUT))


class _Emb42:
    """transformers-4.27 view of CLIPTextEmbeddings: the attributes encode_text_word_embedding.py:26-37 reaches into"""
    def __init__(self, emb, T):
        self.token_embedding, self.position_embedding = emb.token_embedding, emb.position_embedding
        self.position_ids = torch.arange(T).unsqueeze(0)


class _Out42(tuple):
    hidden_states = None
    attentions = None


class _TextModel42:
    """transformers-4.27 `CLIPTextModel.text_model` facade over the installed (5.x, flattened) CLIPTextModel: same modules, same
    arithmetic; only attribute names / call conventions are adapted so that the reference function runs unmodified"""
    def __init__(self, m, T):
        self.m, self.embeddings, self.final_layer_norm = m, _Emb42(m.embeddings, T), m.final_layer_norm

    @staticmethod
    def _build_causal_attention_mask(bsz, seq_len, dtype):   # modeling_clip.py (4.27.3) CLIPTextTransformer._build_causal_attention_mask
        mask = torch.empty(bsz, seq_len, seq_len, dtype=dtype)
        mask.fill_(torch.finfo(dtype).min)
        mask.triu_(1)
        return mask.unsqueeze(1)

    def encoder(self, inputs_embeds, attention_mask, causal_attention_mask, output_attentions, output_hidden_states, return_dict):
        out = self.m.encoder(inputs_embeds=inputs_embeds, attention_mask=causal_attention_mask, is_causal=True)
        return _Out42((out.last_hidden_state,))


def warp_inputs():
    """deterministic inputs of the warp fixture (regenerated by the test, only outputs are stored)"""
    g = torch.Generator().manual_seed(77)
    F = torch.nn.functional
    def smooth(c, h, w):
        return F.interpolate(torch.rand((1, c, h // 8, w // 8), generator=g) * 2 - 1, size=(h, w), mode="bilinear", align_corners=False)
    cloth = smooth(3, 256, 192)
    agnostic = smooth(21, 256, 192)
    refine_in = smooth(24, 64, 48)
    return cloth, agnostic, refine_in


def make_warp_golden():
    """tests/golden/warp_modules.safetensors: the REAL reference ConvNet_TPS / UNetVanilla (hubconf.py:56-58 instantiation) with the
    deterministic synthetic checkpoint (BatchNorm running statistics included), eval mode.  ConvNet_TPS.forward cannot run without a GPU
    (its training-only regularisers hard-code .cuda(), ConvNet_TPS.py:210-214), so the inference data flow of :318-334 is driven through
    the module's own sub-modules: extractionA/B, l2norm, correlation, loc_net.regression, gridGen."""
    import contextlib
    import io
    from src.models.ConvNet_TPS import ConvNet_TPS
    from src.models.UNet import UNetVanilla
    from ladi_vton_amd import configs as C
    with contextlib.redirect_stdout(io.StringIO()):
        tps = ConvNet_TPS(256, 192, 21, 3).eval()
    ref = UNetVanilla(n_channels=24, n_classes=3, bilinear=True).eval()
    tsd = C.synth_state_dict(C.tps_shapes(C.TPS_FULL), "tps.", fp16_round=False)
    tsd["loc_net.regression.linear.weight"] = tsd["loc_net.regression.linear.weight"] * 0.2      # keep tanh out of saturation
    rsd = C.synth_state_dict(C.refine_shapes(C.REFINE_FULL), "refine.", fp16_round=False)
    missing, unexpected = tps.load_state_dict(tsd, strict=False)
    assert not unexpected and all(k.startswith("gridGen.") or k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    missing, unexpected = ref.load_state_dict(rsd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    cloth, agnostic, refine_in = warp_inputs()
    with torch.no_grad():
        fa = tps.l2norm(tps.extractionA(cloth))
        fb = tps.l2norm(tps.extractionB(agnostic))
        corr = tps.correlation(fa, fb)
        coor = tps.loc_net.regression(corr).view(1, -1, 2)
        grid = tps.gridGen(coor).view(1, 256, 192, 2)
        warped = torch.nn.functional.grid_sample(cloth, grid, padding_mode="border")
        refined = ref(refine_in)
    save_file({"corr_sample": corr[:, :, ::4, ::4].contiguous(), "coor": coor.contiguous(), "grid": grid.half().contiguous(),
               "warped": warped.half().contiguous(), "refined": refined.contiguous()}, os.path.join(OUT, "warp_modules.safetensors"))


def warp_composed_inputs(H=384, W=288):
    """deterministic inputs of the composed-warp fixture: cloth [1,3,H,W], im_mask [1,3,H,W], pose_map [1,18,H,W] in [-1, 1]"""
    g = torch.Generator().manual_seed(78)
    F = torch.nn.functional
    def smooth(c):
        return F.interpolate(torch.rand((1, c, H // 8, W // 8), generator=g) * 2 - 1, size=(H, W), mode="bilinear", align_corners=False)
    return smooth(3), smooth(3), smooth(18)


def warp_composed_weights(damp):
    """synthetic checkpoint of the warping module; damp scales the control-point regression (0.2 keeps tanh in its linear range, where an
    error in the regression input shows up undiminished; 1.0 = the raw synthetic weights, tanh partly saturated)"""
    from ladi_vton_amd import configs as C
    tsd = C.synth_state_dict(C.tps_shapes(C.TPS_FULL), "tps.", fp16_round=False)
    tsd["loc_net.regression.linear.weight"] = tsd["loc_net.regression.linear.weight"] * damp
    rsd = C.synth_state_dict(C.refine_shapes(C.REFINE_FULL), "refine.", fp16_round=False)
    return tsd, rsd


def make_warp_composed_golden():
    """tests/golden/warp_composed.safetensors: the warping stage of src/inference.py:239-266 END TO END on the REAL reference modules --
    antialiased bilinear down-sampling of cloth / im_mask / pose_map to 256x192, cat([low_im_mask, low_pose_map]), ConvNet_TPS (through its
    sub-modules: its forward() hard-codes .cuda()), antialiased up-sampling of the grid, F.grid_sample(padding_mode='border'),
    cat([im_mask, pose_map, warped_cloth]) -> UNetVanilla -> clamp(-1, 1) -- for the damped (0.2) and the raw (1.0) regression weights.
    torchvision.transforms.functional.resize(tensor, size, BILINEAR, antialias=True) is torch's interpolate(mode='bilinear',
    antialias=True, align_corners=False) (torchvision/transforms/_functional_tensor.py resize); torchvision is not installed here, so that
    call is made directly.  The reference up-samples the grid to its fixed (512, 384); here to the size of the fixture's cloth."""
    import contextlib
    import io
    sys.path.insert(0, REF)
    from src.models.ConvNet_TPS import ConvNet_TPS
    from src.models.UNet import UNetVanilla
    F = torch.nn.functional

    def resize(x, size):
        return F.interpolate(x, size=size, mode="bilinear", antialias=True, align_corners=False)

    cloth, im_mask, pose_map = warp_composed_inputs()
    H, W = cloth.shape[-2:]
    blob = {}
    for tag, damp in (("", 0.2), ("_raw", 1.0)):
        tsd, rsd = warp_composed_weights(damp)
        with contextlib.redirect_stdout(io.StringIO()):
            tps = ConvNet_TPS(256, 192, 21, 3).eval()
        refinement = UNetVanilla(n_channels=24, n_classes=3, bilinear=True).eval()
        missing, unexpected = tps.load_state_dict(tsd, strict=False)
        assert not unexpected and all(k.startswith("gridGen.") or k.endswith("num_batches_tracked") for k in missing)
        missing, unexpected = refinement.load_state_dict(rsd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
        with torch.no_grad():
            low_cloth = resize(cloth, (256, 192))                                                    # inference.py:242-244
            agnostic = torch.cat([resize(im_mask, (256, 192)), resize(pose_map, (256, 192))], 1)     # :245-251
            fa = tps.l2norm(tps.extractionA(low_cloth.to(torch.float32)))                            # :253 -> ConvNet_TPS.py:318-334
            fb = tps.l2norm(tps.extractionB(agnostic.to(torch.float32)))
            theta = tps.loc_net.regression(tps.correlation(fa, fb)).view(1, -1, 2)
            low_grid = tps.gridGen(theta).view(1, 256, 192, 2)
            highres_grid = resize(low_grid.permute(0, 3, 1, 2), (H, W)).permute(0, 2, 3, 1)           # :256-259
            warped = F.grid_sample(cloth.to(torch.float32), highres_grid.to(torch.float32), padding_mode="border")   # :260
            refined = refinement(torch.cat([im_mask, pose_map, warped], 1).to(torch.float32)).clamp(-1, 1)          # :263-265
        blob["theta" + tag] = theta.contiguous()
        blob["low_grid_sub" + tag] = low_grid[:, ::4, ::4].contiguous()
        blob["warped_sub" + tag] = warped[:, :, ::4, ::4].contiguous()
        blob["refined" + tag] = refined.half().contiguous()
    save_file(blob, os.path.join(OUT, "warp_composed.safetensors"))


def wiring_inputs(B=2, H=128, W=64, L=8, D=128):
    """deterministic inputs of the wiring fixtures (regenerated by the tests; only outputs are stored)"""
    from oracle import pipeline as P
    inp = P.synthetic_inputs(B, H, W, L=L, D=D)
    inp["mask_image"][:, :, 5:9, 3:7] = 0.3                   # non-binary mask values exercise the in-place binarisation
    inp["mask_image"][:, :, 60:66, 30:33] = 0.7
    g = torch.Generator().manual_seed(4242)
    sizes = [(H, W), (H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    return inp, g, sizes


WIRING_CASES = [   # (name, scheduler, steps, cloth_cond_rate, guidance, emasc, cloth_input_type, no_pose)
    ("ddim_full", "ddim", 4, 1.0, 7.5, True, "warped", False),
    ("pndm_full", "pndm", 4, 1.0, 7.5, True, "warped", False),          # 5 evaluations: the 5th sees zero cloth (tryon_pipe.py:718)
    ("pndm_ccr", "pndm", 5, 0.4, 7.5, True, "warped", False),           # cloth zeroed from evaluation i >= 5 - 3.0 = 2
    ("ddim_nocfg", "ddim", 3, 0.5, 1.0, False, "warped", True),         # no CFG, no EMASC, no_pose
]
WIRING_CASES_LMS = [   # separate fixture (ref_wiring_lms.safetensors): init_noise_sigma / scale_model_input placement (tryon_pipe.py:424,722)
    ("lms_full", "lms", 5, 0.5, 7.5, True, "warped", False),          # fractional timesteps 749.25 / 499.5 / 249.75; cloth zero from i = 3
]


def _run_wiring_cases(RH, cases, blob, ucfg, usd, vae, emasc, inp):
    """the REAL StableDiffusionTryOnePipeline.__call__ for each case; records images, every assembled UNet input, timesteps, prompt batch"""
    with torch.no_grad():
        for name, sched, steps, ccr, gscale, use_emasc, cit, no_pose in cases:
            unet = RH.OracleUNet(ucfg, usd)
            sch = {"ddim": RH.DDIMScheduler, "pndm": RH.PNDMScheduler, "lms": RH.LMSDiscreteScheduler}[sched]()
            pipe = RH.real_pipeline(unet, vae, sch, emasc if use_emasc else None, [1, 2, 3, 4, 5] if use_emasc else None)
            pipe.text_encoder = type("T", (), {"dtype": torch.float32})()
            gen = torch.Generator().manual_seed(1234)
            mask = inp["mask_image"].clone()
            res = pipe(image=inp["image"].clone(), mask_image=mask, pose_map=inp["pose_map"].clone(), warped_cloth=inp["warped_cloth"].clone(),
                       prompt_embeds=inp["prompt_embeds"].clone(), negative_prompt_embeds=inp["negative_prompt_embeds"].clone(),
                       height=128, width=64, num_inference_steps=steps, guidance_scale=gscale, generator=gen, output_type="np",
                       cloth_cond_rate=ccr, no_pose=no_pose, cloth_input_type=cit)
            blob["pipe.%s.images" % name] = torch.from_numpy(res.images)[:, ::2, ::2].contiguous()
            blob["pipe.%s.unet_in" % name] = torch.stack([c[0] for c in unet.calls]).contiguous()
            if sched == "lms":
                blob["pipe.%s.timesteps" % name] = torch.tensor([float(c[1]) for c in unet.calls], dtype=torch.float64)
            else:
                blob["pipe.%s.timesteps" % name] = torch.tensor([c[1] for c in unet.calls], dtype=torch.int32)
            blob["pipe.%s.ehs" % name] = unet.calls[0][2].contiguous()
            blob["pipe.%s.mask_after" % name] = mask.contiguous()                     # binarised in place by the reference


def make_wiring_golden():
    """tests/golden/ref_wiring.safetensors: the REAL reference `AutoencoderKL` (src/models/AutoencoderKL.py + src/models/vae.py
    Encoder / Decoder) and the REAL `StableDiffusionTryOnePipeline.__call__` (src/vto_pipelines/tryon_pipe.py), run on the CPU with the
    stand-in diffusers blocks of oracle/ref_harness.py and the tiny synthetic checkpoint.  Stored: encoder moments + the 6-entry
    feature list, decoder outputs for several `int_layers` selections, and for each pipeline case the decoded images, every UNet
    input the real pipeline assembled (31 channels, CFG order, cloth zeroing) and the prompt batch."""
    from oracle import ref_harness as RH
    from oracle import models as M
    from ladi_vton_amd import configs as C
    RH.install()
    from src.models.emasc import EMASC
    vcfg, ucfg = C.VAE_TINY, C.UNET_TINY
    ecfg = C.emasc_for_vae(vcfg)
    vsd = C.synth_state_dict(C.vae_shapes(vcfg), "vae.")
    usd = C.synth_state_dict(C.unet_shapes(ucfg), "unet.")
    esd = C.synth_state_dict(C.emasc_shapes(ecfg), "emasc.")
    vae = RH.real_autoencoder_kl(vcfg, vsd)
    emasc = EMASC(list(ecfg["in_channels"]), list(ecfg["out_channels"]), kernel_size=3, padding=1, stride=1, type="nonlinear").eval()
    emasc.load_state_dict(esd, strict=True)
    inp, g, sizes = wiring_inputs()
    blob = {}

    def sub(t):     # spatial 4x sub-sampling keeps the committed fixture small; the per-channel sums cover what it skips
        return t[:, :, ::4, ::4].contiguous()

    with torch.no_grad():
        # ---- Encoder.forward / AutoencoderKL.encode
        out, feats = vae.encode(inp["image"])
        assert len(feats) == 6 and feats[0] is inp["image"] or torch.equal(feats[0], inp["image"])
        blob["enc.moments"] = out.latent_dist.parameters.contiguous()
        for i in range(1, 6):
            blob["enc.feat%d" % i] = sub(feats[i])
            blob["enc.feat%d.sum" % i] = feats[i].double().sum(dim=(2, 3)).float()
        gs = torch.Generator().manual_seed(7)
        blob["enc.sample"] = out.latent_dist.sample(generator=gs).contiguous()       # randn(seed 7) drawn with the moments' shape
        # ---- Decoder.forward / AutoencoderKL.decode with intermediate features
        z = torch.randn((2, 4, 16, 8), generator=g)
        skips = [torch.randn((2, c, h, w), generator=g) * 0.5 for c, (h, w) in zip(ecfg["out_channels"], sizes)]
        blob["dec.plain"] = sub(vae.decode(z).sample)
        for name, layers in (("12345", [1, 2, 3, 4, 5]), ("2345", [2, 3, 4, 5]), ("345", [3, 4, 5])):
            lst = [skips[i - 1].clone() for i in layers]
            first = lst[0]
            blob["dec." + name] = sub(vae.decode(z.clone(), intermediate_features=lst, int_layers=layers).sample)
            assert lst[-1] is first, "Decoder.forward reverses the caller's list in place (vae.py:190)"
    _run_wiring_cases(RH, WIRING_CASES, blob, ucfg, usd, vae, emasc, inp)
    blob = {k: (v.half() if k.endswith("unet_in") else v) for k, v in blob.items()}   # keep the fixture small
    save_file(blob, os.path.join(OUT, "ref_wiring.safetensors"))


def make_wiring_lms_golden():
    """tests/golden/ref_wiring_lms.safetensors: the REAL pipeline driven with an LMSDiscreteScheduler (oracle LMS behind the diffusers
    interface): pins init_noise_sigma on the initial latents and scale_model_input on the 4 latent channels of every UNet input"""
    from oracle import ref_harness as RH
    from ladi_vton_amd import configs as C
    RH.install()
    from src.models.emasc import EMASC
    vcfg, ucfg = C.VAE_TINY, C.UNET_TINY
    ecfg = C.emasc_for_vae(vcfg)
    vae = RH.real_autoencoder_kl(vcfg, C.synth_state_dict(C.vae_shapes(vcfg), "vae."))
    emasc = EMASC(list(ecfg["in_channels"]), list(ecfg["out_channels"]), kernel_size=3, padding=1, stride=1, type="nonlinear").eval()
    emasc.load_state_dict(C.synth_state_dict(C.emasc_shapes(ecfg), "emasc."), strict=True)
    inp, _, _ = wiring_inputs()
    blob = {}
    _run_wiring_cases(RH, WIRING_CASES_LMS, blob, ucfg, C.synth_state_dict(C.unet_shapes(ucfg), "unet."), vae, emasc, inp)
    blob = {k: (v.half() if k.endswith("unet_in") else v) for k, v in blob.items()}
    save_file(blob, os.path.join(OUT, "ref_wiring_lms.safetensors"))


DATASET_KEYS = ("image", "cloth", "pose_map", "im_mask", "inpaint_mask", "parse_mask_total")


def make_dataset_golden():
    """tests/golden/dataset_ref.safetensors: the REAL reference dataset classes (src/dataset/vitonhd.py VitonHDDataset,
    src/dataset/dresscode.py DressCodeDataset; cv2 / torchvision stood in by oracle/ref_harness.py) on the synthetic trees of
    tests/util_data.py, working size 128 x 96.  Byte / integer outputs are stored whole (uint8), float maps as fp32 (images
    sub-sampled 2x: they come from identical PIL calls, the masks carry the logic)."""
    import tempfile
    from oracle import ref_harness as RH
    from tests import util_data as UD
    RH.install()
    from src.dataset.vitonhd import VitonHDDataset
    from src.dataset.dresscode import DressCodeDataset
    blob = {}
    with tempfile.TemporaryDirectory() as tmp:
        UD.make_vitonhd(os.path.join(tmp, "viton"), n=4)
        UD.make_dresscode(os.path.join(tmp, "dc"), per_category=2)
        sets = {"viton_unpaired": VitonHDDataset(os.path.join(tmp, "viton"), "test", order="unpaired", outputlist=DATASET_KEYS + ("c_name", "im_name", "category"), size=(128, 96)),
                "dc_paired": DressCodeDataset(os.path.join(tmp, "dc"), "test", order="paired", outputlist=DATASET_KEYS + ("c_name", "im_name", "category"), size=(128, 96))}
        for tag, ds in sets.items():
            blob["%s.len" % tag] = torch.tensor([len(ds)])
            for i in range(len(ds)):
                it = ds[i]
                for k in DATASET_KEYS:
                    v = it[k]
                    v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
                    if k in ("image", "cloth", "im_mask"):
                        v = v[:, ::2, ::2].float()
                    elif k == "pose_map":
                        blob["%s.%d.pose_map.sum" % (tag, i)] = v.double().sum(dim=(1, 2)).float()
                        v = v[:, ::4, ::4].float()
                    else:
                        v = v.to(torch.uint8)
                    blob["%s.%d.%s" % (tag, i, k)] = v.contiguous()
                blob["%s.%d.names" % (tag, i)] = torch.tensor(list((it["im_name"] + "|" + it["c_name"] + "|" + it["category"]).encode()), dtype=torch.uint8)
    save_file(blob, os.path.join(OUT, "dataset_ref.safetensors"))


def make_vision_golden(g):
    """tests/golden/clip_vision_tiny.safetensors: installed transformers CLIPVisionModel (the class behind inference.py:269-273's
    vision_encoder) with the deterministic tiny checkpoint: last_hidden_state (what the adapter consumes) and pooler_output"""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from ladi_vton_amd import configs as C
    vc = C.VISION_TINY
    hc = CLIPVisionConfig(hidden_size=vc["hidden"], intermediate_size=vc["mlp_dim"], num_hidden_layers=vc["layers"],
                          num_attention_heads=vc["heads"], image_size=vc["image_size"], patch_size=vc["patch_size"], hidden_act="gelu",
                          layer_norm_eps=vc["layer_norm_eps"], attention_dropout=0.0)
    try:
        hc._attn_implementation = "eager"
    except Exception:
        pass
    m = CLIPVisionModel(hc).eval()
    sd = C.synth_state_dict(C.vision_shapes(vc), "vision.")
    own = m.state_dict()
    flat = {(k[len("vision_model."):] if k[len("vision_model."):] in own else k): v for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(flat, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    x = (torch.randn((3, 3, vc["image_size"], vc["image_size"]), generator=g) * 1.2).half().float()   # CLIP-normalised pixel range
    with torch.no_grad():
        out = m(pixel_values=x)
    save_file({"pixel_values": x, "last_hidden_state": out.last_hidden_state.contiguous(), "pooler_output": out.pooler_output.contiguous()},
              os.path.join(OUT, "clip_vision_tiny.safetensors"))


def make_text_golden(g):
    """tests/golden/clip_text_tiny.safetensors: the reference's own encode_text_word_embedding (imported, not copied) on the installed
    transformers CLIPTextModel with the deterministic tiny checkpoint; rows with and without '$', num_vstar = 3"""
    from transformers import CLIPTextConfig, CLIPTextModel
    from src.utils.encode_text_word_embedding import encode_text_word_embedding  # noqa: E402  (reference code)
    from ladi_vton_amd import configs as C
    tc = C.TEXT_TINY
    hc = CLIPTextConfig(vocab_size=tc["vocab_size"], hidden_size=tc["hidden"], intermediate_size=tc["mlp_dim"], num_hidden_layers=tc["layers"],
                        num_attention_heads=tc["heads"], max_position_embeddings=tc["max_positions"], hidden_act="gelu",
                        layer_norm_eps=tc["layer_norm_eps"], attention_dropout=0.0, bos_token_id=1, eos_token_id=2)
    try:
        hc._attn_implementation = "eager"
    except Exception:
        pass
    m = CLIPTextModel(hc).eval()
    sd = C.synth_state_dict(C.text_shapes(tc), "text.")
    own = m.state_dict()
    flat = {(k[len("text_model."):] if k[len("text_model."):] in own else k): v for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(flat, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    B, T, NV = 4, tc["max_positions"], 3
    ids = torch.randint(3, 250, (B, T), generator=g)
    ids[:, 0] = 300                                  # "bos"-like high id is NOT the argmax ...
    eot = [20, 9, 40, 76]
    for b in range(B):
        ids[b, eot[b]] = 319                         # ... the highest id marks the pooled position (eot convention, :62-65)
        ids[b, eot[b] + 1:] = 0
    ids[0, 10:13] = 259                              # three consecutive '$' (the prompt template of inference.py:289)
    ids[2, 5] = 259; ids[2, 30] = 259                # only the FIRST '$' of a sentence anchors the splice
    # rows 1 and 3 have no '$': untouched
    we = torch.randn((B, NV, tc["hidden"]), generator=g).half().float()
    shim = type("TextEncoder42", (), {})()
    shim.text_model = _TextModel42(m, T)
    with torch.no_grad():
        out = encode_text_word_embedding(shim, ids.clone(), we.clone(), NV)
        plain = m(input_ids=ids)                     # sanity of the facade: without splice rows it must equal the model's own forward
        out_nosplice = encode_text_word_embedding(shim, ids[[1, 3]].clone(), we[[1, 3]].clone(), NV)
    assert torch.allclose(out_nosplice.last_hidden_state, plain.last_hidden_state[[1, 3]], atol=1e-5), "4.27 facade != installed forward"
    save_file({"input_ids": ids.int(), "word_embeddings": we, "last_hidden_state": out.last_hidden_state.contiguous(),
               "pooler_output": out.pooler_output.contiguous()}, os.path.join(OUT, "clip_text_tiny.safetensors"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "warp_composed":      # only this fixture (the others stay byte-identical)
        make_warp_composed_golden()
    else:
        main()
