// CLIP text encoder with the pseudo-word ('$') splice: native counterpart of the reference's encode_text_word_embedding
// (src/utils/encode_text_word_embedding.py:6-72) and of the transformers 4.27.3 CLIPTextTransformer it drives (pre-LN blocks, q scaled
// by head_dim^-0.5, additive causal mask, gelu MLP, final LayerNorm, pooled output at argmax(input_ids)).
// Runs once per batch, outside the denoising loop; reuses the hot path's kernels (igemm, LayerNorm, flash attention with a causal
// flag) plus a gather kernel for the embeddings.
#include "runtime.h"
#include <cmath>
#include <cstring>
#include <utility>
#include <stdexcept>

namespace ladi {

void TextEncoder::load(const TextCfg& c, const WeightStore& ws) {
    cfg = c;
    if (c.hidden % 64 || c.hidden / c.heads != 64) throw std::runtime_error("text encoder: head dim must be 64");
    // released checkpoints use the transformers-4.27 layout (text_model.*); the flattened 5.x layout is accepted too
    const std::string pre = ws.has("text_model.embeddings.token_embedding.weight") ? "text_model." : "";
    const HostTensor& te = ws.get(pre + "embeddings.token_embedding.weight");
    const HostTensor& pe = ws.get(pre + "embeddings.position_embedding.weight");
    if (te.shape.size() != 2 || te.shape[0] != c.vocab || te.shape[1] != c.hidden) throw std::runtime_error("text encoder: token_embedding shape");
    if (pe.shape.size() != 2 || pe.shape[0] != c.max_pos || pe.shape[1] != c.hidden) throw std::runtime_error("text encoder: position_embedding shape");
    tok = pool.upload_h16(te.data);
    pos = pool.upload_h16(pe.data);
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
    final_ln = load_norm(pool, ws, pre + "final_layer_norm");
}

TextEncoder::~TextEncoder() {
    if (h_done) { (void)hipEventSynchronize(h_done); (void)hipEventDestroy(h_done); }
    if (h_meta) (void)hipHostFree(h_meta);
    if (d_ids) (void)hipFree(d_ids);
}

int TextEncoder::forward(const int* ids, int ids_on_device, int B, int T, const h16* word_emb, int nv, h16* out_hidden, h16* out_pooled, hipStream_t st) {
    if (B <= 0 || T <= 0 || T > cfg.max_pos) { set_error("text encoder: bad batch / sequence length"); return -1; }
    const int H = cfg.hidden;
    const int nmeta = B * T + 2 * B;
    if (nmeta > ids_cap) {
        if (d_ids) (void)hipFree(d_ids);
        d_ids = nullptr; ids_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&d_ids), (size_t)nmeta * sizeof(int)) != hipSuccess) { set_error("text encoder: hipMalloc"); return -1; }
        ids_cap = nmeta;
    }
    if (ids_on_device) {
        // ids already on the device (the reference moves them there before the call, inference.py:291): first '$' and end-of-text row
        // are found by a kernel, nothing crosses the host
        if (hipMemcpyAsync(d_ids, ids, (size_t)B * T * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("text encoder: D2D"); return -1; }
        if (ladi_launch_text_meta(d_ids, B, T, cfg.vstar_id, word_emb ? 1 : 0, d_ids + (size_t)B * T, d_ids + (size_t)B * T + B, st)) { set_error("text encoder: meta"); return -1; }
    } else {
        // ---- host side of encode_text_word_embedding.py:12-19,62-65: first '$' per sentence, eot (= argmax id) row per sentence; staged
        //      through a pinned buffer owned by the handle (re-used only after the previous call's copy has executed), so the call
        //      does not synchronise the stream
        if (!h_done && hipEventCreateWithFlags(&h_done, hipEventDisableTiming) != hipSuccess) { set_error("text encoder: event"); return -1; }
        if (h_meta && hipEventSynchronize(h_done) != hipSuccess) { set_error("text encoder: sync"); return -1; }
        if (nmeta > h_cap) {
            if (h_meta) (void)hipHostFree(h_meta);
            h_meta = nullptr; h_cap = 0;
            if (hipHostMalloc(reinterpret_cast<void**>(&h_meta), (size_t)nmeta * sizeof(int), hipHostMallocDefault) != hipSuccess) { set_error("text encoder: hipHostMalloc"); return -1; }
            h_cap = nmeta;
        }
        int* meta = h_meta;
        for (int b = 0; b < B; ++b) {
            int first = -1, arg = 0;
            for (int t = 0; t < T; ++t) {
                const int id = ids[(size_t)b * T + t];
                if (id < 0 || id >= cfg.vocab) { set_error("text encoder: token id out of range"); return -1; }
                meta[(size_t)b * T + t] = id;
                if (id == cfg.vstar_id && first < 0) first = t;
                if (id > ids[(size_t)b * T + arg]) arg = t;     // first maximum, like torch.argmax
            }
            if (word_emb && first >= 0 && first + nv > T) {
                set_error("text encoder: pseudo-word slots run past the sequence end (the reference raises IndexError here)");
                return -1;
            }
            meta[(size_t)B * T + b] = word_emb ? first : -1;
            meta[(size_t)B * T + B + b] = b * T + arg;
        }
        if (hipMemcpyAsync(d_ids, meta, (size_t)nmeta * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) { set_error("text encoder: H2D"); return -1; }
        if (hipEventRecord(h_done, st) != hipSuccess) { set_error("text encoder: event record"); return -1; }
    }
    const int* d_first = d_ids + (size_t)B * T;
    const int* d_eot = d_first + B;

    for (int pass = 0; pass < 2; ++pass) {
        arena.dry = (pass == 0);
        if (pass == 1) arena.reserve(arena.peak);
        arena.off = 0;
        Ctx c; c.st = st; c.ar = &arena;
        Act xa = c.new_act(B, T, 1, H), xb = c.new_act(B, T, 1, H);   // residual stream, ping-pong
        Act* cur = &xa; Act* nxt = &xb;
        if (!c.dry()) c.check(ladi_launch_text_embed(d_ids, d_first, nv, tok, pos, word_emb, B, T, H, cfg.vocab, cur->p, st), "text_embed");
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
                aa.n = B; aa.heads = cfg.heads; aa.Nq = T; aa.Nk = T; aa.scale = 0.125f; aa.causal = 1;   // q * 64^-0.5, causal mask
                c.check(ladi_launch_flash_attn64(aa, st), "text attention");
            }
            ConvOpt oo; oo.res0 = cur;
            Act h1 = conv2d(c, L.o, ao, nullptr, oo);              // x + out_proj(attn)
            Act a2 = layer_norm(c, L.ln2, h1, cfg.ln_eps);
            ConvOpt o1; o1.act = LADI_ACT_GELU;
            Act m = conv2d(c, L.fc1, a2, nullptr, o1);
            {   // nxt = h1 + fc2(m), written into the other residual buffer
                IGemmArgs g;
                std::memset(&g, 0, sizeof(g));
                g.src0 = m.p; g.C0 = m.c; g.ld0 = m.ld; g.Hs = T; g.Ws = 1; g.Ho = T; g.Wo = 1; g.P = B * T;
                g.ksize = 1; g.stride = 1; g.pad = 0; g.W = L.fc2.w; g.Q = L.fc2.cout; g.K = L.fc2.K(); g.bias = L.fc2.b;
                g.act = LADI_ACT_NONE; g.out_scale = 1.f; g.res0 = h1.p; g.ldr0 = h1.ld;
                if (m.c != L.fc2.cin_pad) throw std::runtime_error("text encoder: fc2 channel mismatch");
                launch_conv_into(c, g, *nxt);
            }
            c.ar->release(mk);
            std::swap(cur, nxt);
        }
        const Act& x = *cur;
        Act y; y.p = out_hidden; y.n = B; y.h = T; y.w = 1; y.c = H; y.ld = H;
        if (!c.dry()) {
            c.check(ladi_launch_layernorm(x.p, x.ld, final_ln.g, final_ln.b, cfg.ln_eps, B * T, H, y.p, y.ld, st), "text final LN");
            if (out_pooled) c.check(ladi_launch_gather_rows(out_hidden, d_eot, B, H, out_pooled, st), "text pooled");
        }
        if (c.err) { set_error("text encoder forward launch failure"); return -1; }
    }
    return 0;
}

}  // namespace ladi
