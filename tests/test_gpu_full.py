"""Full-size (released architecture, deterministic random-init checkpoint) single-stage parity on the GPU vs the CPU fp32 oracle
at sizes the oracle finishes in seconds (batch 1), plus size-independent properties at the BASELINE batch.
Tolerances: PSNR (peak = max|ref|) >= 50 dB per stage — the same fp16-storage / fp32-accumulate bound as the tiny-model tests."""
import pytest
import torch

from oracle import configs as C
from oracle import models as M
from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _threads():
    torch.set_num_threads(U.cpu_quota_threads())


def test_full_unet_forward_vs_oracle():
    import ladi_vton_amd as L
    cfg = C.UNET_FULL
    sd = C.synth_state_dict(C.unet_shapes(cfg), "unet.")
    unet = L.NativeUNet(cfg, sd)
    g = torch.Generator().manual_seed(5)
    n, h, w = 1, 64, 48
    x = torch.randn((n, 31, h, w), generator=g).half().float()
    ehs = torch.randn((n, 77, 1024), generator=g).half().float()
    ref = M.unet_forward(sd, cfg, x, 741, ehs)
    got = unet(x.to(U.dev()), 741, encoder_hidden_states=ehs.to(U.dev())).sample.float().cpu()
    assert got.shape == ref.shape
    assert U.psnr(got, ref) >= 50.0, (U.psnr(got, ref), U.rel_l2(got, ref))
    # batch independence: sample i of a batch-4 forward equals the batch-1 forward (every op on the path is per-sample)
    xb = torch.cat([x, x.flip(-1), x * 0.5, x])
    eb = torch.cat([ehs, ehs.flip(1), ehs, ehs])
    gb = unet(xb.to(U.dev()), 741, encoder_hidden_states=eb.to(U.dev())).sample.float().cpu()
    assert U.psnr(gb[0:1], got) >= 55.0 and U.psnr(gb[3:4], got) >= 55.0


def test_full_vae_emasc_roundtrip_vs_oracle():
    import ladi_vton_amd as L
    from oracle import pipeline as P
    vcfg, ecfg = C.VAE_FULL, C.EMASC_FULL
    vsd = C.synth_state_dict(C.vae_shapes(vcfg), "vae.")
    esd = C.synth_state_dict(C.emasc_shapes(ecfg), "emasc.")
    vae, em = L.NativeVAE(vcfg, vsd), L.NativeEMASC(ecfg, esd)
    H, W = 256, 192     # full-width model, quarter-size image: keeps the CPU oracle to seconds
    inp = P.synthetic_inputs(1, H, W, L=4, D=8)
    x, mask = inp["image"], inp["mask_image"]
    mom_ref, feats_ref = M.vae_encode(vsd, vcfg, x)
    enc, feats = vae.encode(x.to(U.dev()))
    assert U.psnr(enc.latent_dist.parameters.float().cpu(), mom_ref) >= 50.0
    sk_ref = M.mask_features(M.emasc_forward(esd, feats_ref[1:6]), mask.clone())
    sk = em([f for f in feats[1:6]], mask=mask.to(U.dev()))
    for a, b in zip(sk, sk_ref):
        assert U.psnr(a.float().cpu(), b) >= 50.0
    z = torch.randn((1, 4, H // 8, W // 8), generator=torch.Generator().manual_seed(9))
    dec_ref = M.vae_decode(vsd, vcfg, z, [s.clone() for s in sk_ref], [1, 2, 3, 4, 5])
    dec = vae.decode(z.to(U.dev()), intermediate_features=list(sk), int_layers=[1, 2, 3, 4, 5]).sample.float().cpu()
    assert U.psnr(dec, dec_ref) >= 50.0, U.psnr(dec, dec_ref)


def test_full_adapter_vs_oracle():
    import ladi_vton_amd as L
    cfg = C.ADAPTER_FULL
    sd = C.synth_state_dict(C.adapter_shapes(cfg), "adapter.")
    ad = L.NativeInversionAdapter(cfg, sd)
    x = torch.randn((3, 257, 1280), generator=torch.Generator().manual_seed(11)).half().float()
    ref = M.adapter_forward(sd, cfg, x)
    got = ad(x.to(U.dev())).float().cpu()
    assert got.shape == (3, 16384)
    U.record_parity("inversion_adapter_full", dict(psnr_db=round(U.psnr(got, ref), 2), rel_l2=U.rel_l2(got, ref)))
    assert U.psnr(got, ref) >= 50.0, U.psnr(got, ref)


def test_full_text_encoder_vs_oracle():
    """SD2 text encoder size (23 layers, 1024-d, 340.4 M parameters), the prompt template of src/inference.py:289 with 16 pseudo-words"""
    import ladi_vton_amd as L
    from oracle import text as T
    cfg = C.TEXT_FULL
    sd = C.synth_state_dict(C.text_shapes(cfg), "text.")
    enc = L.NativeCLIPTextEncoder(cfg, sd)
    B, T_, NV = 2, 77, 16
    g = torch.Generator().manual_seed(21)
    ids = torch.zeros((B, T_), dtype=torch.int32)
    ids[:, 0] = 49406
    for b in range(B):
        n_words = 8 + b
        ids[b, 1:1 + n_words] = torch.randint(300, 40000, (n_words,), generator=g).int()
        ids[b, 1 + n_words:1 + n_words + NV] = 259
        ids[b, 1 + n_words + NV] = 49407
    we = torch.randn((B, NV, cfg["hidden"]), generator=g).half().float() * 0.05
    ref, ref_pooled = T.clip_text_forward(sd, cfg, ids, we, NV)
    out = L.encode_text_word_embedding(enc, ids, we.to(U.dev()), NV)
    torch.cuda.synchronize()
    got, pooled = out.last_hidden_state.float().cpu(), out.pooler_output.float().cpu()
    m = dict(hidden_psnr_db=round(U.psnr(got, ref), 2), hidden_rel_l2=U.rel_l2(got, ref), pooled_rel_l2=U.rel_l2(pooled, ref_pooled))
    U.record_parity("text_encoder_full_23_layers", m)
    # measured on MI355X: rel-L2 1.3e-3 on the hidden states (23 layers of fp16 storage); asserted at ~2x the measured value
    assert m["hidden_psnr_db"] >= 55.0 and m["hidden_rel_l2"] <= 3e-3 and m["pooled_rel_l2"] <= 1e-2, m


def test_full_vision_encoder_feeds_adapter_vs_oracle():
    """ViT-H/14 size (32 layers, 1280-d, 16 heads of 80, 257 tokens, 630.8 M parameters) -> inversion adapter (inference.py:269-277)"""
    import ladi_vton_amd as L
    from oracle import vision as V
    cfg = C.VISION_FULL
    sd = C.synth_state_dict(C.vision_shapes(cfg), "vision.")
    enc = L.NativeCLIPVisionEncoder(cfg, sd)
    px = (torch.randn((2, 3, 224, 224), generator=torch.Generator().manual_seed(31)) * 1.2).half().float()
    ref, ref_pooled = V.clip_vision_forward(sd, cfg, px)
    out = enc(px.to(U.dev()))
    torch.cuda.synchronize()
    got = out.last_hidden_state.float().cpu()
    assert got.shape == (2, 257, 1280)
    m = dict(hidden_psnr_db=round(U.psnr(got, ref), 2), hidden_rel_l2=U.rel_l2(got, ref),
             pooled_rel_l2=U.rel_l2(out.pooler_output.float().cpu(), ref_pooled))
    # measured on MI355X: rel-L2 1.4e-3 on the hidden states (32 layers); asserted at ~2x the measured value
    assert m["hidden_psnr_db"] >= 50.0 and m["hidden_rel_l2"] <= 3e-3 and m["pooled_rel_l2"] <= 1e-2, m
    acfg = C.ADAPTER_FULL
    asd = C.synth_state_dict(C.adapter_shapes(acfg), "adapter.")
    ad = L.NativeInversionAdapter(acfg, asd)
    we = ad(out.last_hidden_state).float().cpu()
    we_ref = M.adapter_forward(asd, acfg, ref)
    m["adapter_after_vision_psnr_db"] = round(U.psnr(we, we_ref), 2)
    U.record_parity("vision_encoder_full_32_layers_to_adapter", m)
    assert m["adapter_after_vision_psnr_db"] >= 40.0, m
