"""End-to-end parity cases whose CPU-oracle side is too slow for the GPU box's test budget (test infrastructure, never imported by the
product).  Each case is a pure function of seeds: `*_inputs()` builds the inputs (used both by `oracle/make_golden_e2e.py`, which runs the
fp32 oracle HERE and commits the outputs under tests/golden/e2e_*.safetensors, and by tests/test_gpu_e2e_golden.py, which feeds the same
inputs to the HIP path on the GPU box and compares with the committed oracle outputs).

Cases:
  unet_n16        one CFG UNet forward at the BENCH batch (n = 16 samples at 64x48: the tile selections `bench.py` times)
  config2_chain   BASELINE configs[2] as benched, producers in the chain (src/inference.py:267-311): in-shop cloth -> CLIP ViT-H/14 ->
                  inversion adapter -> pseudo-word splice -> CLIP text encoder -> try-on pipeline; B = 2, 20 PNDM steps, 512x384
  tryon_b8        BASELINE configs[1] at ITS batch (B = 8, 50 PNDM steps = 51 evaluations, 512x384) against the oracle itself
  tryon_1024      BASELINE configs[4] shapes for the configuration's full 100 DDIM steps (B = 1, 1024x768)
"""
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def unet_n16_inputs():
    g = torch.Generator().manual_seed(77)
    x = torch.randn((16, 31, 64, 48), generator=g).half().float()
    ehs = torch.randn((16, 77, 1024), generator=g).half().float()
    return dict(x=x, ehs=ehs, t=501)


def config2_rows(B=2, H=512, W=384, L=77, D=1024, seed=1234):
    """rows [0, B) of bench.py's synthetic global batch (bench.make_rows), on the CPU as fp32 tensors holding fp16-representable values"""
    ys = torch.arange(H, dtype=torch.float32)[None, :, None]
    xs = torch.arange(W, dtype=torch.float32)[None, None, :]
    keys = ("image", "mask_image", "pose_map", "warped_cloth", "cloth", "prompt_embeds", "noise_cloth", "noise_latents", "noise_masked", "word_ids")
    rows = {k: [] for k in keys}
    for gidx in range(B):
        g = torch.Generator(device="cpu").manual_seed(seed * 1000003 + gidx)

        def smooth():
            low = torch.rand((1, 3, H // 8, W // 8), generator=g) * 2 - 1
            return F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False).clamp(-1, 1)[0]

        rows["image"].append(smooth()); rows["warped_cloth"].append(smooth()); rows["cloth"].append(smooth())
        m = torch.zeros(1, H, W); m[:, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
        rows["mask_image"].append(m)
        cy = torch.rand((18, 1, 1), generator=g) * H
        cx = torch.rand((18, 1, 1), generator=g) * W
        pose = torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 81.0)
        pose[7::9] = 0.0
        rows["pose_map"].append(pose)
        rows["prompt_embeds"].append(torch.randn((L, D), generator=g))
        for k in ("noise_cloth", "noise_latents", "noise_masked"):
            rows[k].append(torch.randn((4, H // 8, W // 8), generator=g))
        ids = torch.zeros(77, dtype=torch.int32)
        nw = 8 + gidx % 3
        ids[0] = 49406
        ids[1:1 + nw] = torch.randint(300, 40000, (nw,), generator=g).int()
        ids[1 + nw:1 + nw + 16] = 259
        ids[1 + nw + 16] = 49407
        rows["word_ids"].append(ids)
    out = {}
    for k, v in rows.items():
        t = torch.stack(v)
        out[k] = t if (k.startswith("noise") or k == "word_ids") else t.half().float()
    gneg = torch.Generator(device="cpu").manual_seed(seed + 4)
    out["negative_prompt_embeds"] = torch.randn((1, L, D), generator=gneg).expand(B, L, D).contiguous().half().float()
    return out


def clip_pixels(cloth):
    """src/inference.py:267-272: (cloth + 1) / 2 -> resize 224 (bilinear, antialias) -> CLIP mean / std normalisation; rounded to fp16, the
    dtype the vision encoder receives in the reference (`.to(weight_dtype)`)"""
    img = F.interpolate((cloth + 1) / 2, size=(224, 224), mode="bilinear", antialias=True, align_corners=False).clamp(0, 1)
    # `processor(images=input_image)` (transformers 4.27.3 CLIPImageProcessor): resize -> to_pil_image = (x * 255).astype(uint8), a truncation;
    # the 224x224 image passes the PIL resize / centre crop unchanged and is rescaled by 1 / 255 before the normalisation
    img = torch.floor(img * 255.0) / 255.0
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return ((img - mean) / std).half().float()
