// EMASC-aware SD VAE (encode with intermediate features, decode with learnable skip adds), EMASC and the
// inversion adapter on the native kernels.
// Reference semantics: src/models/AutoencoderKL.py:145-188, src/models/vae.py:99-119,183-212,329-348,
// src/models/emasc.py:11-40, src/utils/data_utils.py:4-16, src/models/inversion_adapter.py:5-28.
#include "runtime.h"
#include <stdexcept>
#include <cstring>
#include <cmath>

namespace ladi {

static ResBlock load_res_vae(DevPool& pool, const WeightStore& ws, const std::string& p) {
    ResBlock r;
    r.n1 = load_norm(pool, ws, p + ".norm1");
    r.c1 = load_conv(pool, ws, p + ".conv1");
    r.n2 = load_norm(pool, ws, p + ".norm2");
    r.c2 = load_conv(pool, ws, p + ".conv2");
    r.cin = r.c1.cin; r.cout = r.c1.cout;
    r.has_sc = ws.has(p + ".conv_shortcut.weight");
    if (r.has_sc) r.sc = load_conv(pool, ws, p + ".conv_shortcut");
    return r;
}
static VAEAttn load_vae_attn(DevPool& pool, const WeightStore& ws, const std::string& p) {
    VAEAttn a;
    a.gn = load_norm(pool, ws, p + ".group_norm");
    a.qk = load_linear_cat(pool, ws, {p + ".query", p + ".key"}, true);
    a.v = load_conv(pool, ws, p + ".value");
    a.proj = load_conv(pool, ws, p + ".proj_attn");
    a.C = a.v.cout;
    return a;
}

void VAE::load(const VAECfg& c, const WeightStore& ws) {
    cfg = c;
    const int L = c.layers_per_block;
    e_conv_in = load_conv(pool, ws, "encoder.conv_in", c.in_channels);
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L; ++j) e_res.push_back(load_res_vae(pool, ws, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
        if (i < 3) e_down[i] = load_conv(pool, ws, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv");
    }
    e_mid[0] = load_res_vae(pool, ws, "encoder.mid_block.resnets.0");
    e_attn = load_vae_attn(pool, ws, "encoder.mid_block.attentions.0");
    e_mid[1] = load_res_vae(pool, ws, "encoder.mid_block.resnets.1");
    e_norm_out = load_norm(pool, ws, "encoder.conv_norm_out");
    {
        // fold quant_conv (1x1, 2z->2z) into encoder.conv_out (3x3): both linear, no padding interaction
        const HostTensor& wc = ws.get("encoder.conv_out.weight");  // [2z][C][3][3]
        const HostTensor& bc = ws.get("encoder.conv_out.bias");
        const HostTensor& wq = ws.get("quant_conv.weight");        // [2z][2z][1][1]
        const HostTensor& bq = ws.get("quant_conv.bias");
        const int z2 = (int)wc.shape[0], C = (int)wc.shape[1];
        HostTensor wf, bf;
        wf.shape = wc.shape; wf.data.assign(wc.data.size(), 0.f);
        bf.shape = {z2}; bf.data.assign(z2, 0.f);
        const size_t per = (size_t)C * 9;
        for (int o = 0; o < z2; ++o) {
            double bb = bq.data[o];
            for (int m = 0; m < z2; ++m) {
                const float q = wq.data[(size_t)o * z2 + m];
                bb += (double)q * bc.data[m];
                for (size_t e = 0; e < per; ++e) wf.data[o * per + e] += q * wc.data[m * per + e];
            }
            bf.data[o] = (float)bb;
        }
        WeightStore tmp; tmp.m["f.weight"] = wf; tmp.m["f.bias"] = bf;
        e_conv_out = load_conv(pool, tmp, "f");
    }
    // decoder
    {
        const HostTensor& wp = ws.get("post_quant_conv.weight");
        const HostTensor& bp = ws.get("post_quant_conv.bias");
        if (wp.numel() != 16) throw std::runtime_error("post_quant_conv must be 4x4");
        std::vector<float> v(20);
        for (int i = 0; i < 16; ++i) { pq_w[i] = wp.data[i]; v[i] = wp.data[i]; }
        for (int i = 0; i < 4; ++i) { pq_b[i] = bp.data[i]; v[16 + i] = bp.data[i]; }
        d_pq = pool.upload_f32(v);
    }
    d_conv_in = load_conv(pool, ws, "decoder.conv_in", c.latent_channels);
    d_mid[0] = load_res_vae(pool, ws, "decoder.mid_block.resnets.0");
    d_attn = load_vae_attn(pool, ws, "decoder.mid_block.attentions.0");
    d_mid[1] = load_res_vae(pool, ws, "decoder.mid_block.resnets.1");
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L + 1; ++j) d_res.push_back(load_res_vae(pool, ws, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j)));
        if (i < 3) d_up[i] = load_conv(pool, ws, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv");
    }
    d_norm_out = load_norm(pool, ws, "decoder.conv_norm_out");
    d_conv_out = load_conv(pool, ws, "decoder.conv_out");
    d_bad = reinterpret_cast<int*>(pool.alloc(256));
    if (hipMemset(d_bad, 0, 256) != hipSuccess) throw std::runtime_error("VAE: overflow flag");
}

VAE::~VAE() {
    if (stats) (void)hipFree(stats);
    if (h_bad) (void)hipHostFree(h_bad);
    if (ev_bad) (void)hipEventDestroy(ev_bad);
}

void VAE::post_overflow_check(hipStream_t st) {
    if (!d_bad) return;
    if (!h_bad) {
        if (hipHostMalloc(reinterpret_cast<void**>(&h_bad), sizeof(int)) != hipSuccess || hipEventCreateWithFlags(&ev_bad, hipEventDisableTiming) != hipSuccess)
            throw std::runtime_error("VAE: allocating the overflow-flag mailbox failed");
        *h_bad = 0;
    }
    if (hipMemcpyAsync(h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipMemsetAsync(d_bad, 0, sizeof(int), st) != hipSuccess ||
        hipEventRecord(ev_bad, st) != hipSuccess)
        throw std::runtime_error("VAE: queueing the overflow-flag copy failed");
    bad_pending = true;
}

int VAE::poll_overflow() {
    if (!bad_pending) return 0;
    bad_pending = false;
    if (hipEventSynchronize(ev_bad) != hipSuccess) throw std::runtime_error("VAE: waiting for the overflow flag failed");
    if (!*h_bad) return 0;
    *h_bad = 0;
    if (range_shift < 0 && auto_shift < 8) auto_shift = auto_shift == 0 ? 4 : 8;     // the re-submitted run gets 2^4 (then 2^8) more head-room
    return 1;
}

namespace {

struct VF {
    Ctx& c; VAE& v;
    // the residual stream (x in, out of res / attn) is stored multiplied by `s` = 2^-shift (1 except under the decoder's fp16-range guard):
    //   GroupNorm(s x; eps s^2) == GroupNorm(x; eps) exactly, so everything behind a norm is at its true scale;
    //   a branch that ENDS in the stream leaves its producer's epilogue multiplied by s (out_scale);
    //   a convolution that READS the stream directly (shortcut, up-sampler) is linear: its output is already scaled, only its bias needs s.
    float s = 1.f;
    float eps_s() const { return v.cfg.eps * s * s; }
    // ResnetBlock2D without time embedding; optional extra residual (EMASC skip folded into the producer epilogue, pre-scaled by s)
    Act res(const ResBlock& r, const Act& x, const Act* extra) {
        Act out = new_act_with_stats(c, x.n, x.h, x.w, r.cout);
        const size_t mk = c.ar->mark();
        Act s1 = group_norm(c, r.n1, x, nullptr, v.cfg.groups, eps_s(), 1);
        ConvOpt o1; o1.stats = true;
        Act h1 = conv2d(c, r.c1, s1, nullptr, o1);
        Act s2 = group_norm(c, r.n2, h1, nullptr, v.cfg.groups, v.cfg.eps, 1);
        Act sc; const Act* resid = &x;
        if (r.has_sc) { ConvOpt os; os.bias_mul = s; sc = conv2d(c, r.sc, x, nullptr, os); resid = &sc; }
        {
            IGemmArgs a; std::memset(&a, 0, sizeof(a));
            a.src0 = s2.p; a.C0 = s2.c; a.ld0 = s2.ld;
            a.Hs = x.h; a.Ws = x.w; a.Ho = x.h; a.Wo = x.w; a.P = (int)x.pixels();
            a.ksize = 3; a.stride = 1; a.pad = 1;
            a.W = r.c2.w; a.Q = r.c2.cout; a.K = r.c2.K(); a.bias = r.c2.b; a.out_scale = s;
            a.res0 = resid->p; a.ldr0 = resid->ld;
            if (extra) { a.res1 = extra->p; a.ldr1 = extra->ld; }
            launch_conv_into(c, a, out);
        }
        c.ar->release(mk);
        return out;
    }
    // diffusers AttentionBlock (single head, d = C).  C = 128 / 256 / 512 (every released / test configuration): flash attention over
    // the wide head, scores never leave the chip (flash_attn_wide_kernel); other widths: materialised scores through batched GEMMs.
    Act attn(const VAEAttn& at, const Act& x) {
        const int n = x.n, T = x.h * x.w, C = at.C;
        const bool flash = (C == 128 || C == 256 || C == 512) && (T % 4 == 0);
        Act out = new_act_with_stats(c, x.n, x.h, x.w, C);
        const size_t mk = c.ar->mark();
        Act g = group_norm(c, at.gn, x, nullptr, v.cfg.groups, eps_s(), 0);
        Act tok = g; tok.h = T; tok.w = 1;
        ConvOpt op;
        Act qk = conv2d(c, at.qk, tok, nullptr, op);                 // [n*T][2C]
        h16* vt = c.alloc_h16((size_t)n * C * T);                    // V^T [n][C][T]
        float* S = flash ? nullptr : c.alloc_f32((size_t)n * T * T);
        h16* P = flash ? nullptr : c.alloc_h16((size_t)n * T * T);
        Act o = c.new_act(n, T, 1, C);
        if (!c.dry()) {
            if (!flash && (T % 64)) throw std::runtime_error("VAE attention: tokens must be a multiple of 64");
            IGemmArgs a;
            // V^T[b] = Wv * Xn[b]^T + bv   (pixel operand = Wv rows, weight operand = tokens)
            std::memset(&a, 0, sizeof(a));
            a.src0 = at.v.w; a.C0 = at.v.cin_pad; a.ld0 = at.v.cin_pad; a.Hs = C; a.Ws = 1; a.Ho = C; a.Wo = 1; a.P = C;
            a.ksize = 1; a.stride = 1; a.W = g.p; a.Q = T; a.K = C; a.ldw = g.ld; a.bs_w = (long long)T * g.ld;
            a.bias = at.v.b; a.bias_per_pixel = 1; a.out_scale = 1.f; a.out = vt; a.ldo = T; a.bs_out = (long long)C * T;
            c.check(ladi_launch_igemm(a, n, 0, c.st), "igemm(vT)");
            if (flash) {
                AttnArgs fa;
                fa.q = qk.p; fa.k = qk.p + C; fa.v = vt; fa.o = o.p;
                fa.ldq = 2 * C; fa.ldk = 2 * C; fa.ldv = T; fa.ldo = C;
                fa.sq = (long long)T * 2 * C; fa.sk = fa.sq; fa.sv = (long long)C * T; fa.so = (long long)T * C;
                fa.n = n; fa.heads = 1; fa.Nq = T; fa.Nk = T; fa.scale = 1.f / std::sqrt((float)C);
                c.check(ladi_launch_attn_wide(fa, C, c.st), "attn_wide");
            } else {
            // S[b] = Q[b] K[b]^T (fp32)
            std::memset(&a, 0, sizeof(a));
            a.src0 = qk.p; a.C0 = C; a.ld0 = 2 * C; a.bs_src0 = (long long)T * 2 * C; a.Hs = T; a.Ws = 1; a.Ho = T; a.Wo = 1; a.P = T;
            a.ksize = 1; a.stride = 1; a.W = qk.p + C; a.Q = T; a.K = C; a.ldw = 2 * C; a.bs_w = (long long)T * 2 * C;
            a.out_scale = 1.f; a.out = S; a.ldo = T; a.out_f32 = 1; a.bs_out = (long long)T * T;
            c.check(ladi_launch_igemm(a, n, 0, c.st), "igemm(S)");
            c.check(ladi_launch_softmax_rows(S, n * T, T, 1.f / std::sqrt((float)C), P, c.st), "softmax");
            // O[b] = P[b] V[b]
            std::memset(&a, 0, sizeof(a));
            a.src0 = P; a.C0 = T; a.ld0 = T; a.bs_src0 = (long long)T * T; a.Hs = T; a.Ws = 1; a.Ho = T; a.Wo = 1; a.P = T;
            a.ksize = 1; a.stride = 1; a.W = vt; a.Q = C; a.K = T; a.ldw = T; a.bs_w = (long long)C * T;
            a.out_scale = 1.f; a.out = o.p; a.ldo = C; a.bs_out = (long long)T * C;
            c.check(ladi_launch_igemm(a, n, 0, c.st), "igemm(PV)");
            }
            // proj_attn + residual
            std::memset(&a, 0, sizeof(a));
            a.src0 = o.p; a.C0 = C; a.ld0 = C; a.Hs = n * T; a.Ws = 1; a.Ho = n * T; a.Wo = 1; a.P = n * T;
            a.ksize = 1; a.stride = 1; a.W = at.proj.w; a.Q = C; a.K = at.proj.K(); a.bias = at.proj.b; a.out_scale = s;
            a.res0 = x.p; a.ldr0 = x.ld;
            launch_conv_into(c, a, out);
        }
        c.ar->release(mk);
        return out;
    }
};

}  // namespace

Act VAE::encode(Ctx& c, const Act& x, Act feats[5]) {
    VF f{c, *this};
    const int L = cfg.layers_per_block;
    ConvOpt o; o.stats = true;
    Act h = conv2d(c, e_conv_in, x, nullptr, o);
    feats[0] = h;  // idx1 (conv_in output)
    feats[1] = h;  // idx2 (input of down block 0) - same tensor (vae.py:104-109)
    int ri = 0;
    for (int i = 0; i < 4; ++i) {
        if (i > 0) feats[i + 1] = h;  // input of down block i
        for (int j = 0; j < L; ++j) h = f.res(e_res[ri++], h, nullptr);
        if (i < 3) {
            ConvOpt od; od.stats = true; od.stride = 2; od.pad = 0;  // F.pad(0,1,0,1) + stride-2 conv, pad 0: trailing zeros via bounds check
            h = conv2d(c, e_down[i], h, nullptr, od);
        }
    }
    h = f.res(e_mid[0], h, nullptr);
    h = f.attn(e_attn, h);
    h = f.res(e_mid[1], h, nullptr);
    Act g = group_norm(c, e_norm_out, h, nullptr, cfg.groups, cfg.eps, 1);
    ConvOpt oc; oc.out_ld = 8;
    return conv2d(c, e_conv_out, g, nullptr, oc);  // moments (quant_conv folded)
}

Act VAE::decode(Ctx& c, const Act& z, const Act* skips, int shift) {
    VF f{c, *this};
    f.s = std::ldexp(1.f, -shift);
    c.bad = d_bad;
    const int L = cfg.layers_per_block;
    // slot i = EMASC output for encoder feature idx i+1; a slot without a tensor is an int_layers selection that omits it (vae.py:190-205).
    // Skips that are added INTO the stream (slots 1..4) must carry the stream's scale: scaled copies under the range guard
    Act scaled[5];
    for (int i = 1; i < 5; ++i) {
        if (!(skips && skips[i].p) || shift == 0) continue;
        scaled[i] = c.new_act(skips[i].n, skips[i].h, skips[i].w, skips[i].c);
        if (!c.dry()) c.check(ladi_launch_scale_h16(skips[i].p, skips[i].ld, scaled[i].p, scaled[i].ld, skips[i].pixels(), skips[i].c, f.s, c.st), "skip scale");
    }
    auto sk = [&](int i) -> const Act* {
        if (!(skips && skips[i].p)) return nullptr;
        return (shift != 0 && i > 0) ? &scaled[i] : &skips[i];
    };
    ConvOpt o; o.stats = true; o.out_scale = f.s;
    Act h = conv2d(c, d_conv_in, z, nullptr, o);
    h = f.res(d_mid[0], h, nullptr);
    h = f.attn(d_attn, h);
    // vae.py:191-194: sample += reversed(feats)[i] before up_block i  -> folded into the producing epilogue
    h = f.res(d_mid[1], h, sk(4));
    int ri = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L + 1; ++j) h = f.res(d_res[ri++], h, nullptr);
        if (i < 3) {
            ConvOpt ou; ou.ups = 1; ou.stats = true; ou.bias_mul = f.s;     // reads the scaled stream: linear, only the bias needs s
            ou.res0 = sk(3 - i);
            h = conv2d(c, d_up[i], h, nullptr, ou);
        }
    }
    // vae.py:200-205: conv_norm_out -> SiLU -> (+ feats for int layer 1, true scale: it is added behind the norm) -> conv_out
    Act g = group_norm(c, d_norm_out, h, nullptr, cfg.groups, f.eps_s(), 1, sk(0));
    ConvOpt oc; oc.out_ld = 4;
    c.bad = nullptr;
    return conv2d(c, d_conv_out, g, nullptr, oc);
}

bool VAE::overflowed(hipStream_t st) {
    if (!d_bad) return false;
    int h = 0;
    if (hipMemcpyAsync(&h, d_bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        throw std::runtime_error("VAE: reading the overflow flag failed");
    if (h) (void)hipMemsetAsync(d_bad, 0, sizeof(int), st);
    return h != 0;
}

// ------------------------------------------------------------------------------------------------
void EMASC::load(const EMASCCfg& c, const WeightStore& ws) {
    cfg = c;
    for (int i = 0; i < c.n; ++i) {
        a[i] = load_conv(pool, ws, "conv." + std::to_string(i) + ".0", c.in_ch[i]);
        b[i] = load_conv(pool, ws, "conv." + std::to_string(i) + ".2", c.in_ch[i]);
        if (b[i].cout != c.out_ch[i]) throw std::runtime_error("EMASC out_channels mismatch");
    }
}

void EMASC::forward(Ctx& c, const Act* feats, const h16* const* masks, Act* outs, bool outs_preallocated) {
    for (int i = 0; i < cfg.n; ++i) {
        if (!outs_preallocated) outs[i] = c.new_act(feats[i].n, feats[i].h, feats[i].w, b[i].cout);
    }
    for (int i = 0; i < cfg.n; ++i) {
        const size_t mk = c.ar->mark();
        ConvOpt oa; oa.act = LADI_ACT_SILU;
        Act t = conv2d(c, a[i], feats[i], nullptr, oa);
        if (!c.dry()) {
            IGemmArgs g; std::memset(&g, 0, sizeof(g));
            g.src0 = t.p; g.C0 = t.c; g.ld0 = t.ld; g.Hs = t.h; g.Ws = t.w; g.Ho = t.h; g.Wo = t.w; g.P = (int)t.pixels();
            g.ksize = 3; g.stride = 1; g.pad = 1; g.W = b[i].w; g.Q = b[i].cout; g.K = b[i].K(); g.bias = b[i].b; g.out_scale = 1.f;
            g.mask = masks ? masks[i] : nullptr;  // mask_features fused: out *= (1 - mask)
            g.out = outs[i].p; g.ldo = outs[i].ld;
            c.check(ladi_launch_igemm(g, 1, 0, c.st), "igemm(emasc)");
        }
        c.ar->release(mk);
    }
}

// ------------------------------------------------------------------------------------------------
void Adapter::load(const AdapterCfg& c, const WeightStore& ws) {
    cfg = c;
    const std::string e = "encoder_layers.0";
    ln1 = load_norm(pool, ws, e + ".layer_norm1");
    ln2 = load_norm(pool, ws, e + ".layer_norm2");
    post_ln = load_norm(pool, ws, "post_layernorm");
    q = load_conv(pool, ws, e + ".self_attn.q_proj");
    kv = load_linear_cat(pool, ws, {e + ".self_attn.k_proj", e + ".self_attn.v_proj"}, true);
    o = load_conv(pool, ws, e + ".self_attn.out_proj");
    fc1 = load_conv(pool, ws, e + ".mlp.fc1");
    fc2 = load_conv(pool, ws, e + ".mlp.fc2");
    l0 = load_conv(pool, ws, "layers.0");
    l3 = load_conv(pool, ws, "layers.3");
    l6 = load_conv(pool, ws, "layers.6");
}

int Adapter::forward(const h16* x, int B, int T, h16* out, hipStream_t st) {
    // Only the CLS row of the encoder layer output is consumed (inversion_adapter.py:26): K/V need all T tokens,
    // everything downstream of the scores only row 0.
    const int H = cfg.hidden, d = H / cfg.heads;
    for (int pass = 0; pass < 2; ++pass) {
        arena.dry = (pass == 0);
        if (pass == 1) arena.reserve(arena.peak);
        arena.off = 0;
        Ctx c; c.st = st; c.ar = &arena;
        Act xin; xin.p = const_cast<h16*>(x); xin.n = B; xin.h = T; xin.w = 1; xin.c = H; xin.ld = H;
        Act a1 = layer_norm(c, ln1, xin, cfg.ln_eps);                         // [B][T][H]
        ConvOpt op;
        Act kvt = conv2d(c, kv, a1, nullptr, op);                             // [B*T][2H]
        h16* qv = c.alloc_h16((size_t)B * H);
        h16* ao = c.alloc_h16((size_t)B * H);
        h16* h1 = c.alloc_h16((size_t)B * H);
        h16* n2 = c.alloc_h16((size_t)B * H);
        h16* m1 = c.alloc_h16((size_t)B * cfg.mlp);
        h16* h2 = c.alloc_h16((size_t)B * H);
        h16* n3 = c.alloc_h16((size_t)B * H);
        h16* g1 = c.alloc_h16((size_t)B * cfg.head_hidden);
        h16* g2 = c.alloc_h16((size_t)B * cfg.head_hidden);
        if (c.dry()) continue;
        const int ldrow = T * H;  // CLS rows of [B][T][H]
        int rc = 0;
        // q = q_proj(LN1(x))[CLS]  (HF CLIPAttention scales q by d^-0.5 -> applied as the softmax scale)
        rc |= ladi_launch_small_linear(a1.p, 0, ldrow, q.w, q.b, nullptr, 0, B, H, q.cin_pad, LADI_ACT_NONE, 0, qv, 0, H, st);
        rc |= ladi_launch_attn_single_query(qv, H, kvt.p, 2 * H, kvt.p + H, 2 * H, ao, H, B, cfg.heads, d, T, (long long)T * 2 * H,
                                            (long long)T * 2 * H, 1.f / std::sqrt((float)d), st);
        // h1 = x[CLS] + out_proj(attn)
        rc |= ladi_launch_small_linear(ao, 0, H, o.w, o.b, x, ldrow, B, H, o.cin_pad, LADI_ACT_NONE, 0, h1, 0, H, st);
        rc |= ladi_launch_layernorm(h1, H, ln2.g, ln2.b, cfg.ln_eps, B, H, n2, H, st);
        rc |= ladi_launch_small_linear(n2, 0, H, fc1.w, fc1.b, nullptr, 0, B, cfg.mlp, fc1.cin_pad, LADI_ACT_GELU, 0, m1, 0, cfg.mlp, st);
        rc |= ladi_launch_small_linear(m1, 0, cfg.mlp, fc2.w, fc2.b, h1, H, B, H, fc2.cin_pad, LADI_ACT_NONE, 0, h2, 0, H, st);
        rc |= ladi_launch_layernorm(h2, H, post_ln.g, post_ln.b, cfg.ln_eps, B, H, n3, H, st);
        rc |= ladi_launch_small_linear(n3, 0, H, l0.w, l0.b, nullptr, 0, B, cfg.head_hidden, l0.cin_pad, LADI_ACT_GELU, 0, g1, 0, cfg.head_hidden, st);
        rc |= ladi_launch_small_linear(g1, 0, cfg.head_hidden, l3.w, l3.b, nullptr, 0, B, cfg.head_hidden, l3.cin_pad, LADI_ACT_GELU, 0, g2, 0, cfg.head_hidden, st);
        rc |= ladi_launch_small_linear(g2, 0, cfg.head_hidden, l6.w, l6.b, nullptr, 0, B, cfg.out_dim, l6.cin_pad, LADI_ACT_NONE, 0, out, 0, cfg.out_dim, st);
        if (rc) { set_error("adapter forward launch failure"); return -1; }
    }
    return 0;
}

}  // namespace ladi
