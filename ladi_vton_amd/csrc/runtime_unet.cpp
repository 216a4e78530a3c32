// Extended SD2-inpainting UNet (31 input channels) forward on the native kernels.
// Architecture restated in SURVEY.md App. A.1-A.3 (diffusers 0.14.0 UNet2DConditionModel as configured by the
// reference at hubconf.py:31-39 and called at src/vto_pipelines/tryon_pipe.py:732).
#include "runtime.h"
#include <stdexcept>
#include <cstring>
#include <new>
#include <cstdlib>

namespace ladi {

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

static ResBlock load_res(DevPool& pool, const WeightStore& ws, const std::string& p, bool temb) {
    ResBlock r;
    r.n1 = load_norm(pool, ws, p + ".norm1");
    r.c1 = load_conv(pool, ws, p + ".conv1");
    r.n2 = load_norm(pool, ws, p + ".norm2");
    r.c2 = load_conv(pool, ws, p + ".conv2");
    r.cin = r.c1.cin; r.cout = r.c1.cout;
    r.has_sc = ws.has(p + ".conv_shortcut.weight");
    if (r.has_sc) r.sc = load_conv(pool, ws, p + ".conv_shortcut");
    (void)temb;
    return r;
}

static XfBlock load_xf(DevPool& pool, const WeightStore& ws, const std::string& p, int heads, bool pack_fused) {
    XfBlock x;
    x.gn = load_norm(pool, ws, p + ".norm");
    x.proj_in = load_conv(pool, ws, p + ".proj_in");
    const std::string b = p + ".transformer_blocks.0";
    x.ln1 = load_norm(pool, ws, b + ".norm1");
    x.ln2 = load_norm(pool, ws, b + ".norm2");
    x.ln3 = load_norm(pool, ws, b + ".norm3");
    x.qkv = load_linear_cat(pool, ws, {b + ".attn1.to_q", b + ".attn1.to_k", b + ".attn1.to_v"}, false);
    x.o1 = load_conv(pool, ws, b + ".attn1.to_out.0");
    x.q2 = load_conv(pool, ws, b + ".attn2.to_q");
    x.kv2 = load_linear_cat(pool, ws, {b + ".attn2.to_k", b + ".attn2.to_v"}, false);
    x.o2 = load_conv(pool, ws, b + ".attn2.to_out.0");
    x.ff1 = load_geglu(pool, ws, b + ".ff.net.0.proj");
    x.ff2 = load_conv(pool, ws, b + ".ff.net.2");
    x.proj_out = load_conv(pool, ws, p + ".proj_out");
    x.C = x.proj_in.cout;
    x.heads = heads;
    if (x.C != heads * 64) throw std::runtime_error(p + ": head_dim must be 64");
    // the C = 320 level CAN run attn2 and the feed-forward as fused kernels (xf_fused.hip, default off): pack to_out / ff.net.2 per head / per
    // hidden block -- only when the path is switched on (UNet::xf_fuse): the default path reads none of it
    if (pack_fused && ladi_xf_fused_eligible(x.C, heads, 128, 1) && x.o2.cin_pad == x.C && x.ff2.cin_pad == 4 * x.C && x.ff1.cout == 8 * x.C) {
        x.o2_packed = reinterpret_cast<h16*>(pool.alloc(ladi_xf_wo_packed_elems() * sizeof(h16)));
        x.ff2_packed = reinterpret_cast<h16*>(pool.alloc(ladi_xf_w2_packed_elems() * sizeof(h16)));
        if (ladi_launch_pack_wo(x.o2.w, x.o2_packed, nullptr) || ladi_launch_pack_w2(x.ff2.w, x.ff2_packed, nullptr) || hipDeviceSynchronize() != hipSuccess)
            throw std::runtime_error(p + ": packing the fused-block operands failed");
    }
    return x;
}

void UNet::load(const UNetCfg& c, const WeightStore& ws) {
    cfg = c;
    // LADI_XF_FUSE (0 / 1 / 2: none / attn2 only / attn2 + feed-forward as fused kernels) is latched HERE, once per UNet: the value selects
    // different arena allocations in the forward, so a planning pass, its real pass, a later re-plan and the lanes of one forward must all
    // see the same one (it used to be a getenv in every block of every pass)
    { const char* e = getenv("LADI_XF_FUSE"); xf_fuse = e ? atoi(e) : 0; }
    const int L = c.layers_per_block;
    conv_in = load_conv(pool, ws, "conv_in", c.in_channels);
    time_l1 = load_conv(pool, ws, "time_embedding.linear_1");
    time_l2 = load_conv(pool, ws, "time_embedding.linear_2");
    std::vector<std::string> temb_names;
    auto add_res = [&](std::vector<ResBlock>* vec, ResBlock* single, const std::string& p) {
        ResBlock r = load_res(pool, ws, p, true);
        r.temb_off = temb_total;
        temb_total += r.cout;
        temb_names.push_back(p + ".time_emb_proj");
        if (vec) vec->push_back(r); else *single = r;
    };
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L; ++j) {
            const std::string p = "down_blocks." + std::to_string(i);
            add_res(&down_res, nullptr, p + ".resnets." + std::to_string(j));
            if (i < 3) down_xf.push_back(load_xf(pool, ws, p + ".attentions." + std::to_string(j), c.heads[i], xf_fuse > 0));
        }
        if (i < 3) down_samp[i] = load_conv(pool, ws, "down_blocks." + std::to_string(i) + ".downsamplers.0.conv");
    }
    add_res(nullptr, &mid_res[0], "mid_block.resnets.0");
    mid_xf = load_xf(pool, ws, "mid_block.attentions.0", c.heads[3], xf_fuse > 0);
    add_res(nullptr, &mid_res[1], "mid_block.resnets.1");
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L + 1; ++j) {
            const std::string p = "up_blocks." + std::to_string(i);
            add_res(&up_res, nullptr, p + ".resnets." + std::to_string(j));
            if (i > 0) up_xf.push_back(load_xf(pool, ws, p + ".attentions." + std::to_string(j), c.heads[3 - i], xf_fuse > 0));
        }
        if (i < 3) up_samp[i] = load_conv(pool, ws, "up_blocks." + std::to_string(i) + ".upsamplers.0.conv");
    }
    norm_out = load_norm(pool, ws, "conv_norm_out");
    conv_out = load_conv(pool, ws, "conv_out");
    temb_all = load_linear_cat(pool, ws, temb_names, true);
}

UNet::~UNet() {
    if (temb_table) (void)hipFree(temb_table);
    if (stats) (void)hipFree(stats);
    if (in_buf) (void)hipFree(in_buf);
}

int UNet::set_context(const h16* ehs, int n, int L, hipStream_t st) {
    // cross-attention K/V of every transformer block depend only on encoder_hidden_states: hoisted out of the
    // denoising loop (SURVEY.md §8d "step-invariant work")
    std::vector<XfBlock*> all;
    for (auto& x : down_xf) all.push_back(&x);
    all.push_back(&mid_xf);
    for (auto& x : up_xf) all.push_back(&x);
    if (n * L > ctx_cap_n || n > ctx_cap_samples) {
        ctx_pool.reset(new DevPool());   // frees the previous (smaller) K/V cache
        for (auto* x : all) {
            x->kv_cache = reinterpret_cast<h16*>(ctx_pool->alloc((size_t)n * L * 2 * x->C * sizeof(h16)));
            if (x->o2_packed) {
                x->kp_tiles = reinterpret_cast<h16*>(ctx_pool->alloc(ladi_xf_kp_elems(n) * sizeof(h16)));
                x->vt_tiles = reinterpret_cast<h16*>(ctx_pool->alloc(ladi_xf_vt_elems(n) * sizeof(h16)));
            }
        }
        ctx_cap_n = n * L; ctx_cap_samples = n;
    }
    ctx_n = n; ctx_L = L;
    for (auto* x : all) {
        IGemmArgs a; std::memset(&a, 0, sizeof(a));
        a.src0 = ehs; a.C0 = cfg.cross_dim; a.ld0 = cfg.cross_dim;
        a.Hs = n * L; a.Ws = 1; a.Ho = n * L; a.Wo = 1; a.P = n * L;
        a.ksize = 1; a.stride = 1; a.pad = 0;
        a.W = x->kv2.w; a.Q = x->kv2.cout; a.K = x->kv2.K();
        a.out = x->kv_cache; a.ldo = 2 * x->C; a.out_scale = 1.f;
        int rc = ladi_launch_igemm(a, 1, 0, st);
        if (rc) { set_error("set_context igemm rc=" + std::to_string(rc)); return rc; }
        if (x->kp_tiles && ladi_xf_fused_eligible(x->C, x->heads, 128, L)) {      // the fused attn2 reads per-(sample, head) tiles
            rc = ladi_launch_pack_kv_tiles(x->kv_cache, n, L, x->C, x->kp_tiles, x->vt_tiles, st);
            if (rc) { set_error("set_context pack_kv_tiles rc=" + std::to_string(rc)); return rc; }
        }
    }
    return 0;
}

int UNet::compute_temb(const float* ts_host, int count, hipStream_t st) {
    const int tdim = time_l1.cout;   // 1280
    const int edim = time_l1.cin;    // 320
    if (count > temb_rows_cap) {
        if (temb_table) (void)hipFree(temb_table);
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&temb_table), (size_t)count * (temb_total + 2 * tdim + edim + 1) * sizeof(float)));
        temb_rows_cap = count;
    }
    float* table = temb_table;                               // [count][temb_total]
    float* t_dev = table + (size_t)temb_rows_cap * temb_total;  // [count]
    float* emb0 = t_dev + temb_rows_cap;                     // [count][edim]
    float* h1 = emb0 + (size_t)temb_rows_cap * edim;         // [count][tdim]
    float* h2 = h1 + (size_t)temb_rows_cap * tdim;           // [count][tdim]
    HIP_OK(hipMemcpyAsync(t_dev, ts_host, (size_t)count * sizeof(float), hipMemcpyHostToDevice, st));
    int rc = ladi_launch_timestep_embedding(t_dev, count, edim, emb0, st);
    // reference casts the sinusoid to the model dtype (fp16) before linear_1; keep fp32 (more accurate)
    if (!rc) rc = ladi_launch_small_linear(emb0, 1, edim, time_l1.w, time_l1.b, nullptr, 0, count, tdim, time_l1.cin_pad, LADI_ACT_SILU, 0, h1, 1, tdim, st);
    if (!rc) rc = ladi_launch_small_linear(h1, 1, tdim, time_l2.w, time_l2.b, nullptr, 0, count, tdim, time_l2.cin_pad, LADI_ACT_NONE, 0, h2, 1, tdim, st);
    if (!rc) rc = ladi_launch_small_linear(h2, 1, tdim, temb_all.w, temb_all.b, nullptr, 0, count, temb_total, temb_all.cin_pad, LADI_ACT_NONE, 1, table, 1, temb_total, st);
    temb_rows = count;
    if (rc) set_error("compute_temb rc=" + std::to_string(rc));
    return rc;
}

namespace {

struct Fwd {
    Ctx& c; UNet& u; const float* temb; const int* tidx; int sample0;
    Act res(const ResBlock& r, const Act& x, const Act* x2) {
        Act out;
        // block output is allocated first so temporaries can be released (stack discipline)
        out = new_act_with_stats(c, x.n, x.h, x.w, r.cout);
        const size_t mk = c.ar->mark();
        Act s1 = group_norm(c, r.n1, x, x2, u.cfg.groups, u.cfg.eps, 1);
        ConvOpt o1; o1.stats = true; o1.rowadd = temb + r.temb_off; o1.rowadd_idx = tidx; o1.rowadd_stride = u.temb_total;
        Act h1 = conv2d(c, r.c1, s1, nullptr, o1);
        Act s2 = group_norm(c, r.n2, h1, nullptr, u.cfg.groups, u.cfg.eps, 1);
        Act sc;
        const Act* resid = &x;
        if (r.has_sc) { ConvOpt os; sc = conv2d(c, r.sc, x, x2, os); resid = &sc; }
        else if (x2) throw std::runtime_error("resnet: concat input requires conv_shortcut");
        ConvOpt o2; o2.res0 = resid;
        Act y = conv2d_into(r.c2, s2, o2, out);
        c.ar->release(mk);
        return y;
    }
    // conv writing into a pre-allocated output
    Act conv2d_into(const DConv& cv, const Act& x, ConvOpt o, Act out) {
        IGemmArgs a; std::memset(&a, 0, sizeof(a));
        a.src0 = x.p; a.C0 = x.c; a.ld0 = x.ld;
        a.Hs = x.h; a.Ws = x.w; a.Ho = out.h; a.Wo = out.w; a.P = out.n * out.h * out.w;
        a.ksize = cv.k; a.stride = 1; a.pad = cv.k / 2; a.ups = 0;
        a.W = cv.w; a.Q = cv.cout; a.K = cv.K();
        a.bias = cv.b; a.act = o.act; a.out_scale = 1.f;
        if (o.res0) { a.res0 = o.res0->p; a.ldr0 = o.res0->ld; }
        if (o.res1) { a.res1 = o.res1->p; a.ldr1 = o.res1->ld; }
        launch_conv_into(c, a, out);
        return out;
    }
    Act attn(const h16* q, int ldq, long long sq, const h16* k, const h16* v, int ldkv, long long skv, int n, int T, int Nk,
             int heads) {
        Act o = c.new_act(n, T, 1, heads * 64);
        if (c.dry()) return o;
        AttnArgs a;
        a.q = q; a.k = k; a.v = v; a.o = o.p;
        a.ldq = ldq; a.ldk = ldkv; a.ldv = ldkv; a.ldo = o.ld;
        a.sq = sq; a.sk = skv; a.sv = skv; a.so = (long long)T * o.ld;
        a.n = n; a.heads = heads; a.Nq = T; a.Nk = Nk; a.scale = 0.125f;
        c.check(ladi_launch_flash_attn64(a, c.st), "flash_attn64");
        return o;
    }
    Act xf(const XfBlock& b, const Act& x) {
        const int n = x.n, T = x.h * x.w, C = b.C;
        Act out = new_act_with_stats(c, x.n, x.h, x.w, C);
        const size_t mk = c.ar->mark();
        // GroupNorm (no activation) -> proj_in: the normalisation's affine is applied to proj_in's register panel where the X-stationary
        // kernel carries the projection (64x48 / 32x24 levels), else as its own pass
        ConvOpt op;
        Act tok;
        if (gn_fusable(x, b.proj_in.cout)) {
            op.gn_ss = gn_scale_shift(c, b.gn, x, nullptr, u.cfg.groups, 1e-6f);
            tok = x;
        } else tok = group_norm(c, b.gn, x, nullptr, u.cfg.groups, 1e-6f, 0);
        tok.h = T; tok.w = 1;  // tokens view [n][T][C]
        tok.st_part = nullptr; tok.st_px = 0;
        Act t0 = conv2d(c, b.proj_in, tok, nullptr, op);
        ConvOpt oq; oq.ln = &b.ln1;                  // LayerNorm fused into the K = 320 / 640 projections (X-stationary kernel), else a launch
        Act qkv = conv2d(c, b.qkv, t0, nullptr, oq);
        Act o1 = attn(qkv.p, 3 * C, (long long)T * 3 * C, qkv.p + C, qkv.p + 2 * C, 3 * C, (long long)T * 3 * C, n, T, T, b.heads);
        ConvOpt or1; or1.res0 = &t0;
        Act t1 = conv2d(c, b.o1, o1, nullptr, or1);
        // attn2 and the feed-forward as ONE kernel each on the C = 320 level (xf_fused.hip; LADI_XF_FUSE=0 / =1 / =2: none / attn2 only / both,
        // latched at UNet::load), else the chain of projections.  Measured on
        // MI355X (profiles/r05_xf_fused_ab.txt): parity-green and bit-reproducible, but at one wave per SIMD the fused chains expose the latency two
        // waves per SIMD hide in the separate launches -- attn2 85.7 us against the ~76 us of the three launches it replaces (forward
        // +0.05 ms), feed-forward 275 us against 109 + 57 us (forward +0.45 ms) -- so the DEFAULT IS OFF; the kernels stay selectable.
        int fuse = 0;
        if (b.o2_packed && b.kp_tiles && t1.ld == C && ladi_xf_fused_eligible(C, b.heads, T, u.ctx_L)) fuse = u.xf_fuse;
        Act t2;
        if (fuse >= 1) {
            t2 = c.new_act(n, T, 1, C);
            if (!c.dry()) {
                XAttnBlockArgs xa;
                xa.x = t1.p; xa.ln_g = b.ln2.g; xa.ln_b = b.ln2.b; xa.ln_eps = 1e-5f;
                xa.Wq = b.q2.w;
                xa.Kp = b.kp_tiles + ladi_xf_kp_elems(sample0); xa.Vt = b.vt_tiles + ladi_xf_vt_elems(sample0);
                xa.Wo = b.o2_packed; xa.bo = b.o2.b; xa.res = t1.p; xa.out = t2.p;
                xa.P = n * T; xa.T = T; xa.nk = u.ctx_L; xa.scale = 0.125f;
                c.check(ladi_launch_xattn_block(xa, c.st), "xattn_block");
            }
        } else {
            ConvOpt oq2; oq2.ln = &b.ln2;
            Act q2 = conv2d(c, b.q2, t1, nullptr, oq2);
            const h16* kv = b.kv_cache + (size_t)sample0 * u.ctx_L * 2 * C;     // this lane's samples of the cached cross-attention K / V
            Act o2 = attn(q2.p, C, (long long)T * C, kv, kv + C, 2 * C, (long long)u.ctx_L * 2 * C, n, T, u.ctx_L, b.heads);
            ConvOpt or2; or2.res0 = &t1;
            t2 = conv2d(c, b.o2, o2, nullptr, or2);
        }
        Act t3;
        if (fuse >= 2) {
            t3 = c.new_act(n, T, 1, C);
            if (!c.dry()) {
                FFBlockArgs fa;
                fa.x = t2.p; fa.ln_g = b.ln3.g; fa.ln_b = b.ln3.b; fa.ln_eps = 1e-5f;
                fa.W1 = b.ff1.w; fa.b1 = b.ff1.b; fa.W2 = b.ff2_packed; fa.bo = b.ff2.b; fa.res = t2.p; fa.out = t3.p; fa.P = n * T;
                c.check(ladi_launch_ff_block(fa, c.st), "ff_block");
            }
        } else {
            ConvOpt og; og.act = LADI_ACT_GEGLU; og.ln = &b.ln3;
            Act gg = conv2d(c, b.ff1, t2, nullptr, og);
            ConvOpt or3; or3.res0 = &t2;
            t3 = conv2d(c, b.ff2, gg, nullptr, or3);
        }
        Act xin = x; xin.h = T; xin.w = 1;
        Act outv = out; outv.h = T; outv.w = 1;
        ConvOpt oo; oo.res0 = &xin;
        outv = conv2d_into(b.proj_out, t3, oo, outv);
        out.st_part = outv.st_part; out.st_px = outv.st_px;
        c.ar->release(mk);
        return out;
    }
};

}  // namespace

Act UNet::forward(Ctx& c, const Act& x, const float* temb_row, const int* temb_idx, const Act* eps_out, int sample0) {
    if ((eps_out ? sample0 + x.n > ctx_n : ctx_n != x.n) && !c.dry()) throw std::runtime_error("UNet::forward: set_context batch mismatch");
    Fwd f{c, *this, temb_row, temb_idx, sample0};
    const int L = cfg.layers_per_block;
    std::vector<Act> skips;
    ConvOpt o; o.stats = true;
    Act h = conv2d(c, conv_in, x, nullptr, o);
    skips.push_back(h);
    int ri = 0, xi = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L; ++j) {
            h = f.res(down_res[ri++], h, nullptr);
            if (i < 3) h = f.xf(down_xf[xi++], h);
            skips.push_back(h);
        }
        if (i < 3) {
            ConvOpt od; od.stride = 2; od.pad = 1; od.stats = true;
            h = conv2d(c, down_samp[i], h, nullptr, od);
            skips.push_back(h);
        }
    }
    h = f.res(mid_res[0], h, nullptr);
    h = f.xf(mid_xf, h);
    h = f.res(mid_res[1], h, nullptr);
    ri = 0; xi = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L + 1; ++j) {
            Act sk = skips.back(); skips.pop_back();
            h = f.res(up_res[ri++], h, &sk);
            if (i > 0) h = f.xf(up_xf[xi++], h);
        }
        if (i < 3) {
            ConvOpt ou; ou.ups = 1; ou.stats = true;
            h = conv2d(c, up_samp[i], h, nullptr, ou);
        }
    }
    Act g = group_norm(c, norm_out, h, nullptr, cfg.groups, cfg.eps, 1);
    ConvOpt oc; oc.out_ld = 4;
    if (cfg.out_channels > 4) oc.out_ld = (cfg.out_channels + 3) / 4 * 4;
    if (eps_out) return f.conv2d_into(conv_out, g, oc, *eps_out);     // a sample-group lane writes its rows of the shared output
    return conv2d(c, conv_out, g, nullptr, oc);
}

// ------------------------------------------------------------------------------------------------
// Sample-group lanes (runtime.h): G forwards of n / G samples on G streams
// ------------------------------------------------------------------------------------------------
#define HIP_OK_L(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

int UNetLanes::pick(int n) {
    int want = 1;
    if (const char* e = getenv("LADI_UNET_LANES")) want = atoi(e);
    if (want < 1) want = 1;
    if (want > MAXG) want = MAXG;
    while (want > 1 && (n % want)) --want;
    return want;
}

void UNetLanes::reset() {
    for (int i = 0; i < MAXG; ++i) {
        if (st[i]) (void)hipStreamSynchronize(st[i]);
        arena[i].~Arena(); new (&arena[i]) Arena();
        peak[i] = 0;
    }
}

void UNetLanes::configure(int n, int g) {
    const int want = g > 0 ? g : pick(n);
    if (want != G && arena[0].cap) reset();     // another lane count than the one the arenas were planned for: start from a clean plan
    G = want;
    if (G < 1 || G > MAXG || (n % G)) throw std::runtime_error("UNetLanes: the lane count must divide the sample count");
    if (!fork) HIP_OK_L(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int i = 1; i < G; ++i) {
        if (!st[i]) HIP_OK_L(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        if (!join[i]) HIP_OK_L(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
    }
}

void UNetLanes::forward(UNet& u, hipStream_t main_st, bool dry, bool concurrent, const Act& x, const Act& eps, const float* temb, const int* tidx) {
    const int ng = x.n / G;
    const bool par = concurrent && !dry && G > 1;
    if (par) HIP_OK_L(hipEventRecord(fork, main_st));
    for (int g = 0; g < G; ++g) {
        hipStream_t sg = (par && g > 0) ? st[g] : main_st;
        if (par && g > 0) HIP_OK_L(hipStreamWaitEvent(sg, fork, 0));
        arena[g].dry = dry; arena[g].off = 0;
        Ctx c; c.st = sg; c.ar = &arena[g]; c.stats = stats[g]; c.stats_cap = stats_cap[g]; c.sk_cnt = sk_cnt[g];
        if (!dry && stats_cap[g]) HIP_OK_L(hipMemsetAsync(stats[g], 0, stats_cap[g] * sizeof(float), sg));
        Act xg = x; xg.n = ng; xg.p = x.p + (size_t)g * ng * x.h * x.w * x.ld;
        Act eg = eps; eg.n = ng; eg.p = eps.p + (size_t)g * ng * eps.h * eps.w * eps.ld; eg.st_part = nullptr; eg.st_px = 0;
        (void)u.forward(c, xg, temb, tidx, &eg, g * ng);
        if (dry) { peak[g] = arena[g].peak; stats_peak[g] = c.stats_peak; }
        if (par && g > 0) { HIP_OK_L(hipEventRecord(join[g], sg)); HIP_OK_L(hipStreamWaitEvent(main_st, join[g], 0)); }
    }
}

void UNetLanes::commit_plan() {
    for (int g = 0; g < G; ++g) {
        if (!sk_cnt[g]) {
            HIP_OK_L(hipMalloc(reinterpret_cast<void**>(&sk_cnt[g]), 1024 * sizeof(int)));
            HIP_OK_L(hipMemset(sk_cnt[g], 0, 1024 * sizeof(int)));
        }
        arena[g].reserve(peak[g] + 4096);
        if (stats_peak[g] > stats_cap[g]) {
            if (stats[g]) (void)hipFree(stats[g]);
            stats[g] = nullptr;
            HIP_OK_L(hipMalloc(reinterpret_cast<void**>(&stats[g]), stats_peak[g] * sizeof(float)));
            stats_cap[g] = stats_peak[g];
        }
    }
}

unsigned long long UNetLanes::key() const {
    unsigned long long h = 0x9e3779b97f4a7c15ULL * (unsigned long long)G;
    for (int g = 0; g < G; ++g) {
        h ^= (unsigned long long)(uintptr_t)arena[g].base + (h << 6) + (h >> 2);
        h ^= (unsigned long long)(uintptr_t)stats[g] + (h << 6) + (h >> 2);
    }
    return h;
}

UNetLanes::~UNetLanes() {
    for (int i = 0; i < MAXG; ++i) {
        if (stats[i]) (void)hipFree(stats[i]);
        if (sk_cnt[i]) (void)hipFree(sk_cnt[i]);
        if (join[i]) (void)hipEventDestroy(join[i]);
        if (st[i]) (void)hipStreamDestroy(st[i]);
    }
    if (fork) (void)hipEventDestroy(fork);
}

}  // namespace ladi
