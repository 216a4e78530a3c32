// CLIP ViT-H/14 vision encoder: native counterpart of the transformers 4.27.3 CLIPVisionTransformer the reference calls at
// src/inference.py:269-273 (`vision_encoder(pixel_values).last_hidden_state`, consumed by the inversion adapter at :276).
// Patch embedding = patch-row gather + one GEMM whose epilogue adds the (class-token-augmented) position embeddings, pre_layrnorm,
// 32 pre-LN blocks (fused QKV GEMM, d = 80 MFMA attention, out-proj + residual, gelu MLP).  Once per batch, outside the denoising loop.
#include "runtime.h"
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <utility>

namespace ladi {

void VisionEncoder::load(const VisionCfg& c, const WeightStore& ws) {
    cfg = c;
    const int H = c.hidden, d = H / c.heads;
    if (H % 32 || c.heads * d != H || !(d == 64 || d == 80 || d == 96 || d == 128)) throw std::runtime_error("vision encoder: head dim must be 64/80/96/128");
    if (c.image % c.patch) throw std::runtime_error("vision encoder: image size must be a multiple of the patch size");
    const std::string pre = ws.has("vision_model.embeddings.class_embedding") ? "vision_model." : "";
    const HostTensor& pw = ws.get(pre + "embeddings.patch_embedding.weight");
    const HostTensor& ce = ws.get(pre + "embeddings.class_embedding");
    const HostTensor& pe = ws.get(pre + "embeddings.position_embedding.weight");
    const int nk = 3 * c.patch * c.patch, kp = (nk + 63) / 64 * 64, T = tokens();
    if (pw.numel() != (size_t)H * nk || ce.numel() != (size_t)H || pe.numel() != (size_t)T * H) throw std::runtime_error("vision encoder: embedding shapes");
    std::vector<float> w((size_t)H * kp, 0.f);
    for (int q = 0; q < H; ++q) std::memcpy(&w[(size_t)q * kp], &pw.data[(size_t)q * nk], (size_t)nk * sizeof(float));
    patch.w = pool.upload_h16(w); patch.b = nullptr; patch.cin = nk; patch.cin_pad = kp; patch.cout = H; patch.k = 1;
    std::vector<float> pc(pe.data);
    for (int i = 0; i < H; ++i) pc[i] += ce.data[i];          // the class token sits in row 0 (its patch row is zero)
    posc = pool.upload_h16(pc);
    pre_ln = load_norm(pool, ws, pre + "pre_layrnorm");      // (sic: the key is misspelled upstream)
    post_ln = load_norm(pool, ws, pre + "post_layernorm");
    layers.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        const std::string e = pre + "encoder.layers." + std::to_string(i);
        TextLayer& L = layers[i];
        L.ln1 = load_norm(pool, ws, e + ".layer_norm1");
        L.ln2 = load_norm(pool, ws, e + ".layer_norm2");
        L.qkv = load_linear_cat(pool, ws, {e + ".self_attn.q_proj", e + ".self_attn.k_proj", e + ".self_attn.v_proj"}, true);
        L.o = load_conv(pool, ws, e + ".self_attn.out_proj");
        L.fc1 = load_conv(pool, ws, e + ".mlp.fc1");
        L.fc2 = load_conv(pool, ws, e + ".mlp.fc2");
    }
}

int VisionEncoder::forward(const void* pixels, int in_f32, int B, h16* out_hidden, h16* out_pooled, hipStream_t st) {
    if (B <= 0) { set_error("vision encoder: bad batch"); return -1; }
    const int H = cfg.hidden, T = tokens(), d = H / cfg.heads;
    for (int pass = 0; pass < 2; ++pass) {
        arena.dry = (pass == 0);
        if (pass == 1) arena.reserve(arena.peak);
        arena.off = 0;
        Ctx c; c.st = st; c.ar = &arena;
        Act xa = c.new_act(B, T, 1, H), xb = c.new_act(B, T, 1, H);   // residual stream, ping-pong
        Act* cur = &xa; Act* nxt = &xb;
        {   // embeddings: patch rows -> GEMM (+ position / class embeddings as the residual, shared by every image) -> pre_layrnorm
            const size_t mk = c.ar->mark();
            Act pr = c.new_act(B, T, 1, patch.cin_pad);
            Act emb = c.new_act(B, T, 1, H);
            if (!c.dry()) {
                c.check(ladi_launch_patchify(pixels, in_f32, B, cfg.image, cfg.patch, patch.cin_pad, pr.p, st), "patchify");
                IGemmArgs g;
                std::memset(&g, 0, sizeof(g));
                g.src0 = pr.p; g.C0 = pr.c; g.ld0 = pr.ld; g.Hs = T; g.Ws = 1; g.Ho = T; g.Wo = 1; g.P = T;
                g.ksize = 1; g.stride = 1; g.pad = 0; g.W = patch.w; g.Q = H; g.K = patch.K();
                g.bs_src0 = (long long)T * pr.ld; g.bs_w = 0; g.bs_out = (long long)T * emb.ld; g.bs_res = 0;
                g.act = LADI_ACT_NONE; g.out_scale = 1.f; g.res0 = posc; g.ldr0 = H; g.out = emb.p; g.ldo = emb.ld;
                c.check(ladi_launch_igemm(g, B, 0, st), "patch embedding");
                c.check(ladi_launch_layernorm(emb.p, emb.ld, pre_ln.g, pre_ln.b, cfg.ln_eps, B * T, H, cur->p, cur->ld, st), "pre_layrnorm");
            }
            c.ar->release(mk);
        }
        for (const TextLayer& L : layers) {
            const size_t mk = c.ar->mark();
            Act a = layer_norm(c, L.ln1, *cur, cfg.ln_eps);
            ConvOpt op;
            Act qkv = conv2d(c, L.qkv, a, nullptr, op);            // [B*T][3H], bias fused
            Act ao = c.new_act(B, T, 1, H);
            if (!c.dry()) {
                AttnArgs aa;
                aa.q = qkv.p; aa.k = qkv.p + H; aa.v = qkv.p + 2 * H; aa.o = ao.p;
                aa.ldq = aa.ldk = aa.ldv = qkv.ld; aa.ldo = ao.ld;
                aa.sq = aa.sk = aa.sv = (long long)T * qkv.ld; aa.so = (long long)T * ao.ld;
                aa.n = B; aa.heads = cfg.heads; aa.Nq = T; aa.Nk = T; aa.scale = 1.f / std::sqrt((float)d);
                if (d == 64) c.check(ladi_launch_flash_attn64(aa, st), "vision attention");
                else c.check(ladi_launch_attn_generic(aa, d, st), "vision attention");
            }
            ConvOpt oo; oo.res0 = cur;
            Act h1 = conv2d(c, L.o, ao, nullptr, oo);              // x + out_proj(attn)
            Act a2 = layer_norm(c, L.ln2, h1, cfg.ln_eps);
            ConvOpt o1; o1.act = LADI_ACT_GELU;
            Act m = conv2d(c, L.fc1, a2, nullptr, o1);
            {
                IGemmArgs g;
                std::memset(&g, 0, sizeof(g));
                g.src0 = m.p; g.C0 = m.c; g.ld0 = m.ld; g.Hs = T; g.Ws = 1; g.Ho = T; g.Wo = 1; g.P = B * T;
                g.ksize = 1; g.stride = 1; g.pad = 0; g.W = L.fc2.w; g.Q = L.fc2.cout; g.K = L.fc2.K(); g.bias = L.fc2.b;
                g.act = LADI_ACT_NONE; g.out_scale = 1.f; g.res0 = h1.p; g.ldr0 = h1.ld;
                if (m.c != L.fc2.cin_pad) throw std::runtime_error("vision encoder: fc2 channel mismatch");
                // the last block writes straight into the caller's buffer
                Act dst = *nxt;
                if (&L == &layers.back()) { dst.p = out_hidden; dst.ld = H; }
                launch_conv_into(c, g, dst);
            }
            c.ar->release(mk);
            std::swap(cur, nxt);
        }
        if (!c.dry() && out_pooled)
            c.check(ladi_launch_layernorm(out_hidden, T * H, post_ln.g, post_ln.b, cfg.ln_eps, B, H, out_pooled, H, st), "post_layernorm");
    }
    return 0;
}

}  // namespace ladi
