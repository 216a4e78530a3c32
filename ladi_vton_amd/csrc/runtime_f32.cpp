// fp32 graphs of the warping module: the TPS matching network and the refinement UNet with fp32 weights, fp32 NHWC activations and
// fp32 accumulation (f32path.hip), selected when the caller passes fp32 tensors -- which is what the reference does
// (src/inference.py:253 `tps(low_cloth.to(torch.float32), agnostic.to(torch.float32))`, :264 `refinement(warped_cloth.to(torch.float32))`).
// Same data flow as runtime_tps.cpp / runtime_refine.cpp (ConvNet_TPS.py:315-337, UNet.py:23-34); BatchNorm is inference-mode.
#include "runtime.h"
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace ladi {

static int pad8(int c) { return (c + 7) / 8 * 8; }

DConvF load_conv_f32(DevPool& pool, const HostTensor& w, const HostTensor* bias) {
    DConvF d;
    if (w.shape.size() == 4) { d.cout = (int)w.shape[0]; d.cin = (int)w.shape[1]; d.k = (int)w.shape[2]; }
    else if (w.shape.size() == 2) { d.cout = (int)w.shape[0]; d.cin = (int)w.shape[1]; d.k = 1; }
    else throw std::runtime_error("unsupported weight rank");
    d.cin_pad = pad8(d.cin);
    const int taps = d.k * d.k;
    std::vector<float> r((size_t)d.cout * taps * d.cin_pad, 0.f);
    for (int o = 0; o < d.cout; ++o)
        for (int i = 0; i < d.cin; ++i)
            for (int t = 0; t < taps; ++t)
                r[((size_t)o * taps + t) * d.cin_pad + i] = w.data[((size_t)o * d.cin + i) * taps + t];
    d.w = pool.upload_f32(r);
    if (bias) d.b = pool.upload_f32(bias->data);
    return d;
}

void fold_conv_bn(const WeightStore& ws, const std::string& conv, const std::string& bn, float eps, HostTensor& fw, HostTensor& fb) {
    const HostTensor& w = ws.get(conv + ".weight");
    const HostTensor& g = ws.get(bn + ".weight");
    const HostTensor& b = ws.get(bn + ".bias");
    const HostTensor& m = ws.get(bn + ".running_mean");
    const HostTensor& v = ws.get(bn + ".running_var");
    const size_t cout = (size_t)w.shape[0], per = w.numel() / cout;
    if (g.numel() != cout || b.numel() != cout || m.numel() != cout || v.numel() != cout) throw std::runtime_error(bn + ": BatchNorm size mismatch");
    const bool has_b = ws.has(conv + ".bias");
    fw.shape = w.shape; fw.data.resize(w.numel());
    fb.shape = {(int64_t)cout}; fb.data.resize(cout);
    for (size_t q = 0; q < cout; ++q) {
        const float s = g.data[q] / std::sqrt(v.data[q] + eps);
        for (size_t i = 0; i < per; ++i) fw.data[q * per + i] = w.data[q * per + i] * s;
        fb.data[q] = ((has_b ? ws.get(conv + ".bias").data[q] : 0.f) - m.data[q]) * s + b.data[q];
    }
}

ActF new_act_f32(Ctx& c, int n, int h, int w, int cc) {
    ActF a; a.n = n; a.h = h; a.w = w; a.c = cc; a.ld = cc;
    a.p = c.alloc_f32(a.pixels() * (size_t)a.ld);
    return a;
}

ActF conv2d_f32(Ctx& c, const DConvF& cv, const ActF& x, const ActF* x2, int stride, int pad, int act) {
    const int Ho = stride == 1 ? x.h : x.h / 2, Wo = stride == 1 ? x.w : x.w / 2;
    ActF out = new_act_f32(c, x.n, Ho, Wo, pad8(cv.cout));
    if (c.dry()) return out;
    const int c1 = x2 ? x2->c : 0;
    if (x.c + c1 != cv.cin_pad) throw std::runtime_error("conv2d_f32: channel mismatch");
    if (out.ld != cv.cout)      // padded output channels must read as zeros downstream
        if (hipMemsetAsync(out.p, 0, out.pixels() * (size_t)out.ld * sizeof(float), c.st) != hipSuccess) throw std::runtime_error("conv2d_f32: memset");
    ConvF32Args a;
    std::memset(&a, 0, sizeof(a));
    a.src0 = x.p; a.C0 = x.c; a.ld0 = x.ld;
    if (x2) { a.src1 = x2->p; a.C1 = x2->c; a.ld1 = x2->ld; }
    a.Hs = x.h; a.Ws = x.w; a.Ho = Ho; a.Wo = Wo; a.P = x.n * Ho * Wo;
    a.ksize = cv.k; a.stride = stride; a.pad = pad;
    a.W = cv.w; a.Q = cv.cout; a.K = cv.K(); a.ldw = 0;
    a.bias = cv.b; a.act = act; a.out = out.p; a.ldo = out.ld;
    c.check(ladi_launch_conv_f32(a, 1, c.st), "conv_f32");
    return out;
}

// ------------------------------------------------------------------------------------------------ refinement UNet
static ActF double_conv_f32(Ctx& c, const DoubleConvWF& d, const ActF& x, const ActF* x2) {
    ActF m = conv2d_f32(c, d.c1, x, x2, 1, 1, LADI_ACT_RELU);
    return conv2d_f32(c, d.c2, m, nullptr, 1, 1, LADI_ACT_RELU);
}
static ActF pool2_f32(Ctx& c, const ActF& x) {
    ActF o = new_act_f32(c, x.n, x.h / 2, x.w / 2, x.c);
    if (!c.dry()) c.check(ladi_launch_maxpool2_f32(x.p, x.ld, x.n, x.h, x.w, x.c, o.p, o.ld, c.st), "maxpool2_f32");
    return o;
}
static ActF up2_f32(Ctx& c, const ActF& x) {
    ActF o = new_act_f32(c, x.n, x.h * 2, x.w * 2, x.c);
    if (!c.dry()) c.check(ladi_launch_upsample2x_bilinear_ac_f32(x.p, x.ld, x.n, x.h, x.w, x.c, o.p, o.ld, c.st), "upsample2x_f32");
    return o;
}

int Refine::forward_f32(const void* x, int B, int H, int W, void* out, int out_f32, hipStream_t st) {
    for (int pass = 0; pass < 2; ++pass) {
        arena.dry = (pass == 0);
        if (pass == 1) arena.reserve(arena.peak);
        arena.off = 0;
        Ctx c; c.st = st; c.ar = &arena;
        ActF x0 = new_act_f32(c, B, H, W, incf.c1.cin_pad);
        if (!c.dry()) c.check(ladi_launch_nchw_to_nhwc_f32(x, 1, B, cfg.in_ch, H, W, x0.p, x0.ld, st), "nchw_to_nhwc_f32");
        ActF x1 = double_conv_f32(c, incf, x0, nullptr);
        ActF x2 = double_conv_f32(c, downf[0], pool2_f32(c, x1), nullptr);
        ActF x3 = double_conv_f32(c, downf[1], pool2_f32(c, x2), nullptr);
        ActF x4 = double_conv_f32(c, downf[2], pool2_f32(c, x3), nullptr);
        ActF x5 = double_conv_f32(c, downf[3], pool2_f32(c, x4), nullptr);
        ActF u1 = up2_f32(c, x5); ActF y = double_conv_f32(c, upf[0], x4, &u1);      // cat([skip, upsampled]) (unet_parts.py:62)
        ActF u2 = up2_f32(c, y); y = double_conv_f32(c, upf[1], x3, &u2);
        ActF u3 = up2_f32(c, y); y = double_conv_f32(c, upf[2], x2, &u3);
        ActF u4 = up2_f32(c, y); y = double_conv_f32(c, upf[3], x1, &u4);
        ActF lg = conv2d_f32(c, outcf, y, nullptr, 1, 0, LADI_ACT_NONE);
        if (!c.dry()) c.check(ladi_launch_nhwc_to_nchw_f32(lg.p, lg.ld, B, cfg.out_ch, H, W, out, out_f32, st), "nhwc_to_nchw_f32");
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ TPS matching network
int Tps::forward_f32(const void* a, const void* b, int B, float* grid, float* coor, hipStream_t st) {
    const int H = cfg.height, W = cfg.width, fh = H / 16, fw = W / 16, hw = fh * fw, N = cfg.grid * cfg.grid;
    for (int pass = 0; pass < 2; ++pass) {
        arena.dry = (pass == 0);
        if (pass == 1) arena.reserve(arena.peak);
        arena.off = 0;
        Ctx c; c.st = st; c.ar = &arena;
        auto extract = [&](const TpsExtractF& e, const TpsExtract& e16, const void* src, int cin) -> ActF {
            ActF x = new_act_f32(c, B, H, W, e.conv[0].cin_pad);
            if (!c.dry()) c.check(ladi_launch_nchw_to_nhwc_f32(src, 1, B, cin, H, W, x.p, x.ld, st), "nchw_to_nhwc_f32");
            const int nconv = (int)e.conv.size();
            for (int i = 0; i < nconv; ++i) {
                const bool s2 = e.conv[i].k == 4;
                x = conv2d_f32(c, e.conv[i], x, nullptr, s2 ? 2 : 1, 1, LADI_ACT_RELU);            // conv -> ReLU
                if (i + 1 < nconv && !c.dry())                                                      // -> BatchNorm (ConvNet_TPS.py:38-50)
                    c.check(ladi_launch_channel_affine_f32(x.p, x.ld, x.pixels(), x.c, e16.bn_scale[i], e16.bn_shift[i], st), "batchnorm_f32");
            }
            if (!c.dry()) c.check(ladi_launch_l2norm_rows_f32(x.p, x.ld, (int)x.pixels(), x.c, st), "l2norm_f32");
            return x;
        };
        ActF fa = extract(eaf, ea, a, 3);
        ActF fb = extract(ebf, eb, b, cfg.input_nc);
        if (fa.h != fh || fa.w != fw) throw std::runtime_error("TPS: unexpected feature size");
        const int C = fa.c;
        ActF fap = new_act_f32(c, B, fh, fw, C);
        ActF corr = new_act_f32(c, B, fh, fw, hw);
        if (!c.dry()) {
            c.check(ladi_launch_gather_rows_f32(fa.p, d_perm, B * hw, C, fap.p, st), "correlation row order");
            ConvF32Args g;
            std::memset(&g, 0, sizeof(g));
            g.src0 = fb.p; g.C0 = C; g.ld0 = fb.ld; g.Hs = fh; g.Ws = fw; g.Ho = fh; g.Wo = fw; g.P = hw;
            g.ksize = 1; g.stride = 1; g.pad = 0; g.W = fap.p; g.Q = hw; g.K = C; g.ldw = C;
            g.bs_src0 = (long long)hw * fb.ld; g.bs_w = (long long)hw * C; g.bs_out = (long long)hw * corr.ld;
            g.act = LADI_ACT_NONE; g.out = corr.p; g.ldo = corr.ld;
            c.check(ladi_launch_conv_f32(g, B, st), "correlation_f32");
        }
        ActF x = corr;
        for (int i = 0; i < 4; ++i) {
            const bool s2 = regf[i].k == 4;
            x = conv2d_f32(c, regf[i], x, nullptr, s2 ? 2 : 1, 1, LADI_ACT_RELU);
        }
        float* co = coor ? coor : c.alloc_f32((size_t)B * N * 2);
        if (!c.dry()) {
            const int feat = (int)((size_t)x.h * x.w * x.ld);
            c.check(ladi_launch_linear_f32(x.p, feat, linf_w, linf_b, B, 2 * N, feat, LADI_ACT_TANH, co, 2 * N, st), "regression linear f32");
            c.check(ladi_launch_tps_grid(co, d_inv, d_ctrl, N, B, H, W, grid, st), "tps grid");
        }
    }
    return 0;
}

}  // namespace ladi
