// TPS geometric-matching network: native counterpart of the inference data flow of src/models/ConvNet_TPS.py ConvNet_TPS.forward
// (:315-337): FeatureExtraction A / B (:28-56: conv4x4 s2 -> ReLU -> BatchNorm, x (1 + n_layers), then conv3x3 -> ReLU -> BatchNorm ->
// conv3x3 -> ReLU), FeatureL2Norm (:59-66), FeatureCorrelation (:69-81), BoundedGridLocNet's FeatureRegression (:92-127, 197-200: convs
// with BatchNorm -> ReLU, linear, tanh) and TPSGridGen (:130-185).  The training-only regularisers of loc_net.forward (:201-224, hard-coded
// .cuda()) are not part of the inference path and are not computed.
// BatchNorm runs in inference mode: folded into the conv where it directly follows it (regression), a per-channel affine kernel where a
// ReLU sits in between (extraction).  The 4x4 stride-2 convs run on the igemm's generic tap loop; the correlation is one batched GEMM.
#include "runtime.h"
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace ladi {

namespace {

void bn_affine(DevPool& pool, const WeightStore& ws, const std::string& bn, float eps, float*& scale, float*& shift) {
    const HostTensor& g = ws.get(bn + ".weight");
    const HostTensor& b = ws.get(bn + ".bias");
    const HostTensor& m = ws.get(bn + ".running_mean");
    const HostTensor& v = ws.get(bn + ".running_var");
    std::vector<float> s(g.numel()), t(g.numel());
    for (size_t i = 0; i < s.size(); ++i) { s[i] = g.data[i] / std::sqrt(v.data[i] + eps); t[i] = b.data[i] - m.data[i] * s[i]; }
    scale = pool.upload_f32(s); shift = pool.upload_f32(t);
}

DConv conv_bn_folded(DevPool& pool, const WeightStore& ws, const std::string& conv, const std::string& bn, float eps) {
    const HostTensor& w = ws.get(conv + ".weight");
    const HostTensor& cb = ws.get(conv + ".bias");
    const HostTensor& g = ws.get(bn + ".weight");
    const HostTensor& b = ws.get(bn + ".bias");
    const HostTensor& m = ws.get(bn + ".running_mean");
    const HostTensor& v = ws.get(bn + ".running_var");
    const size_t cout = (size_t)w.shape[0], per = w.numel() / cout;
    WeightStore t;
    HostTensor& fw = t.m["f.weight"]; fw.shape = w.shape; fw.data.resize(w.numel());
    HostTensor& fb = t.m["f.bias"]; fb.shape = {(int64_t)cout}; fb.data.resize(cout);
    for (size_t q = 0; q < cout; ++q) {
        const float s = g.data[q] / std::sqrt(v.data[q] + eps);
        for (size_t i = 0; i < per; ++i) fw.data[q * per + i] = w.data[q * per + i] * s;
        fb.data[q] = (cb.data[q] - m.data[q]) * s + b.data[q];
    }
    return load_conv(pool, t, "f");
}

void load_extract(DevPool& pool, const WeightStore& ws, const std::string& p, int n_layers, float eps, TpsExtract& e, TpsExtractF& ef) {
    const int nconv = n_layers + 3;                       // (1 + n_layers) stride-2 convs + two 3x3 convs
    int idx = 0;
    for (int i = 0; i < nconv; ++i) {
        e.conv.push_back(load_conv(pool, ws, p + ".model." + std::to_string(idx)));
        ef.conv.push_back(load_conv_f32(pool, ws.get(p + ".model." + std::to_string(idx) + ".weight"), &ws.get(p + ".model." + std::to_string(idx) + ".bias")));
        if (i + 1 < nconv) {
            float *s = nullptr, *t = nullptr;
            bn_affine(pool, ws, p + ".model." + std::to_string(idx + 2), eps, s, t);
            e.bn_scale.push_back(s); e.bn_shift.push_back(t);
        }
        idx += 3;
    }
}

// symmetric-free Gauss-Jordan inverse in double (the (N+3)^2 TPS kernel matrix, N = 25)
std::vector<double> invert(std::vector<double> a, int n) {
    std::vector<double> inv((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a[(size_t)r * n + c]) > std::fabs(a[(size_t)piv * n + c])) piv = r;
        if (std::fabs(a[(size_t)piv * n + c]) < 1e-12) throw std::runtime_error("TPS kernel matrix is singular");
        if (piv != c) for (int k = 0; k < n; ++k) { std::swap(a[(size_t)piv * n + k], a[(size_t)c * n + k]); std::swap(inv[(size_t)piv * n + k], inv[(size_t)c * n + k]); }
        const double d = 1.0 / a[(size_t)c * n + c];
        for (int k = 0; k < n; ++k) { a[(size_t)c * n + k] *= d; inv[(size_t)c * n + k] *= d; }
        for (int r = 0; r < n; ++r) if (r != c) {
            const double f = a[(size_t)r * n + c];
            if (f != 0.0) for (int k = 0; k < n; ++k) { a[(size_t)r * n + k] -= f * a[(size_t)c * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k]; }
        }
    }
    return inv;
}

}  // namespace

void Tps::load(const TpsCfg& c, const WeightStore& ws) {
    cfg = c;
    if ((c.height % 64) || (c.width % 64) || c.grid * c.grid > 32) throw std::runtime_error("TPS: height / width must be multiples of 64, grid <= 5");
    load_extract(pool, ws, "extractionA", c.n_layers, c.bn_eps, ea, eaf);
    load_extract(pool, ws, "extractionB", c.n_layers, c.bn_eps, eb, ebf);
    const std::string r = "loc_net.regression.conv.";
    for (int i = 0; i < 4; ++i) {
        reg[i] = conv_bn_folded(pool, ws, r + std::to_string(3 * i), r + std::to_string(3 * i + 1), c.bn_eps);
        HostTensor fw, fb;
        fold_conv_bn(ws, r + std::to_string(3 * i), r + std::to_string(3 * i + 1), c.bn_eps, fw, fb);
        regf[i] = load_conv_f32(pool, fw, &fb);
    }
    {   // linear over flatten(NCHW [64, h, w]) -> columns re-ordered to our NHWC flatten ((h, w), c)
        const HostTensor& w = ws.get("loc_net.regression.linear.weight");
        const HostTensor& b = ws.get("loc_net.regression.linear.bias");
        const int h = c.height / 64, wd = c.width / 64, ch = reg[3].cout, nout = (int)w.shape[0];
        if ((int)w.shape[1] != ch * h * wd || nout != 2 * c.grid * c.grid) throw std::runtime_error("TPS: regression.linear shape");
        std::vector<float> p((size_t)nout * ch * h * wd);
        for (int o = 0; o < nout; ++o)
            for (int cc = 0; cc < ch; ++cc)
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < wd; ++x)
                        p[((size_t)o * h * wd + (size_t)y * wd + x) * ch + cc] = w.data[((size_t)o * ch + cc) * h * wd + (size_t)y * wd + x];
        lin.w = pool.upload_h16(p); lin.b = pool.upload_h16(b.data); lin.cin = lin.cin_pad = ch * h * wd; lin.cout = nout; lin.k = 1;
        linf_w = pool.upload_f32(p); linf_b = pool.upload_f32(b.data);
    }
    // TPSGridGen.__init__ (:132-170): control lattice of range 0.9 (ConvNet_TPS.__init__ :291-306), padded kernel matrix, its inverse
    const int G = c.grid, N = G * G, M = N + 3;
    std::vector<float> ctrl((size_t)N * 2);
    for (int iy = 0; iy < G; ++iy)
        for (int ix = 0; ix < G; ++ix) {
            ctrl[((size_t)iy * G + ix) * 2 + 0] = (float)(-0.9 + (2.0 * 0.9 / (G - 1)) * ix);
            ctrl[((size_t)iy * G + ix) * 2 + 1] = (float)(-0.9 + (2.0 * 0.9 / (G - 1)) * iy);
        }
    std::vector<double> K((size_t)M * M, 0.0);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) {
            const float dx = ctrl[i * 2] - ctrl[j * 2], dy = ctrl[i * 2 + 1] - ctrl[j * 2 + 1];
            const float r2 = dx * dx + dy * dy;
            K[(size_t)i * M + j] = r2 > 0.f ? 0.5f * r2 * std::log(r2) : 0.f;
        }
        K[(size_t)i * M + N] = 1.0; K[(size_t)N * M + i] = 1.0;
        K[(size_t)i * M + N + 1] = ctrl[i * 2]; K[(size_t)i * M + N + 2] = ctrl[i * 2 + 1];
        K[(size_t)(N + 1) * M + i] = ctrl[i * 2]; K[(size_t)(N + 2) * M + i] = ctrl[i * 2 + 1];
    }
    const std::vector<double> inv = invert(K, M);
    std::vector<float> invf(inv.begin(), inv.end());
    d_inv = pool.upload_f32(invf);
    d_ctrl = pool.upload_f32(ctrl);
}

Tps::~Tps() {
    if (d_perm) (void)hipFree(d_perm);
}

int Tps::forward(const void* a, const void* b, int in_f32, int B, float* grid, float* coor, hipStream_t st) {
    if (B <= 0 || !grid) { set_error("TPS: bad arguments"); return -1; }
    const int H = cfg.height, W = cfg.width, fh = H / 16, fw = W / 16, hw = fh * fw, N = cfg.grid * cfg.grid;
    // correlation row order of feature A: channel k <-> A position (y = k % fh, x = k / fh)  (ConvNet_TPS.py:76 transposes h and w)
    if (B * hw > perm_cap) {
        if (d_perm) (void)hipFree(d_perm);
        d_perm = nullptr; perm_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&d_perm), (size_t)B * hw * sizeof(int)) != hipSuccess) { set_error("TPS: hipMalloc"); return -1; }
        std::vector<int> p((size_t)B * hw);
        for (int bb = 0; bb < B; ++bb)
            for (int k = 0; k < hw; ++k) p[(size_t)bb * hw + k] = bb * hw + (k % fh) * fw + k / fh;
        if (hipMemcpy(d_perm, p.data(), p.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { set_error("TPS: H2D"); return -1; }
        perm_cap = B * hw;
    }
    // fp32 tensors from the caller (what src/inference.py:253 passes): the whole network runs in fp32 (runtime_f32.cpp)
    if (in_f32) return forward_f32(a, b, B, grid, coor, st);
    for (int pass = 0; pass < 2; ++pass) {
        arena.dry = (pass == 0);
        if (pass == 1) arena.reserve(arena.peak);
        arena.off = 0;
        Ctx c; c.st = st; c.ar = &arena;
        auto extract = [&](const TpsExtract& e, const void* src, int cin) -> Act {
            Act x = c.new_act(B, H, W, e.conv[0].cin_pad);
            if (!c.dry()) {
                if (hipMemsetAsync(x.p, 0, x.pixels() * (size_t)x.ld * sizeof(h16), st) != hipSuccess) throw std::runtime_error("TPS: memset");
                c.check(ladi_launch_nchw_to_nhwc(src, in_f32, B, cin, H, W, x.p, x.ld, st), "nchw_to_nhwc");
            }
            const int nconv = (int)e.conv.size();
            for (int i = 0; i < nconv; ++i) {
                ConvOpt o; o.act = LADI_ACT_RELU;
                if (e.conv[i].k == 4) { o.stride = 2; o.pad = 1; }
                x = conv2d(c, e.conv[i], x, nullptr, o);
                if (i + 1 < nconv && !c.dry())
                    c.check(ladi_launch_channel_affine(x.p, x.ld, x.pixels(), x.c, e.bn_scale[i], e.bn_shift[i], x.p, x.ld, st), "batchnorm");
            }
            if (!c.dry()) c.check(ladi_launch_l2norm_rows(x.p, x.ld, (int)x.pixels(), x.c, x.p, x.ld, st), "l2norm");
            return x;
        };
        Act fa = extract(ea, a, 3);
        Act fb = extract(eb, b, cfg.input_nc);
        if (fa.h != fh || fa.w != fw) throw std::runtime_error("TPS: unexpected feature size");
        const int C = fa.c;
        Act fap = c.new_act(B, fh, fw, C);
        Act corr = c.new_act(B, fh, fw, hw);
        if (!c.dry()) {
            c.check(ladi_launch_gather_rows(fa.p, d_perm, B * hw, C, fap.p, st), "correlation row order");
            IGemmArgs g;
            std::memset(&g, 0, sizeof(g));
            g.src0 = fb.p; g.C0 = C; g.ld0 = fb.ld; g.Hs = fh; g.Ws = fw; g.Ho = fh; g.Wo = fw; g.P = hw;
            g.ksize = 1; g.stride = 1; g.pad = 0; g.W = fap.p; g.Q = hw; g.K = C; g.ldw = C;
            g.bs_src0 = (long long)hw * fb.ld; g.bs_w = (long long)hw * C; g.bs_out = (long long)hw * corr.ld;
            g.act = LADI_ACT_NONE; g.out_scale = 1.f; g.out = corr.p; g.ldo = corr.ld;
            c.check(ladi_launch_igemm(g, B, 0, st), "correlation");
        }
        Act x = corr;
        for (int i = 0; i < 4; ++i) {
            ConvOpt o; o.act = LADI_ACT_RELU;
            if (reg[i].k == 4) { o.stride = 2; o.pad = 1; }
            x = conv2d(c, reg[i], x, nullptr, o);
        }
        float* co = coor ? coor : c.alloc_f32((size_t)B * N * 2);
        if (!c.dry()) {
            const int feat = (int)((size_t)x.h * x.w * x.c);
            if (feat != lin.cin) throw std::runtime_error("TPS: regression feature size mismatch");
            c.check(ladi_launch_small_linear(x.p, 0, feat, lin.w, lin.b, nullptr, 0, B, lin.cout, lin.cin, LADI_ACT_TANH, 0, co, 1, lin.cout, st), "regression linear");
            c.check(ladi_launch_tps_grid(co, d_inv, d_ctrl, N, B, H, W, grid, st), "tps grid");
        }
    }
    return 0;
}

}  // namespace ladi
