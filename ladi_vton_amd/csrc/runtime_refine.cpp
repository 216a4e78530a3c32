// Refinement UNet of the warping module: native counterpart of src/models/UNet.py UNetVanilla.forward (:23-34) with the parts of
// src/models/unet_parts.py (DoubleConv :10-25 = (conv3x3 no bias -> BatchNorm -> ReLU) x 2, Down :28-39 = MaxPool2d(2) + DoubleConv,
// Up :42-66 = bilinear x2 upsample (align_corners=True) + cat([skip, up]) + DoubleConv, OutConv :69-75), as instantiated by
// hubconf.py:57 (n_channels=24, n_classes=3, bilinear=True) and called at src/inference.py:264 on cat([im_mask, pose_map, warped_cloth]).
// BatchNorm runs in inference mode and is folded into the preceding conv at load time; the concat is the two-source K loop of the igemm;
// ReLU is an igemm epilogue.
#include "runtime.h"
#include <cmath>
#include <stdexcept>

namespace ladi {

namespace {

// conv (no bias) + BatchNorm(eval) -> conv with bias:  w' = w * g / sqrt(var + eps),  b' = beta - mean * g / sqrt(var + eps)
DConv load_conv_bn(DevPool& pool, const WeightStore& ws, const std::string& conv, const std::string& bn, float eps) {
    const HostTensor& w = ws.get(conv + ".weight");
    const HostTensor& g = ws.get(bn + ".weight");
    const HostTensor& b = ws.get(bn + ".bias");
    const HostTensor& m = ws.get(bn + ".running_mean");
    const HostTensor& v = ws.get(bn + ".running_var");
    const size_t cout = (size_t)w.shape[0], per = w.numel() / cout;
    if (g.numel() != cout || b.numel() != cout || m.numel() != cout || v.numel() != cout) throw std::runtime_error(bn + ": BatchNorm size mismatch");
    WeightStore t;
    HostTensor& fw = t.m["f.weight"]; fw.shape = w.shape; fw.data.resize(w.numel());
    HostTensor& fb = t.m["f.bias"]; fb.shape = {(int64_t)cout}; fb.data.resize(cout);
    for (size_t q = 0; q < cout; ++q) {
        const float s = g.data[q] / std::sqrt(v.data[q] + eps);
        for (size_t i = 0; i < per; ++i) fw.data[q * per + i] = w.data[q * per + i] * s;
        fb.data[q] = b.data[q] - m.data[q] * s + (ws.has(conv + ".bias") ? ws.get(conv + ".bias").data[q] * s : 0.f);
    }
    return load_conv(pool, t, "f");
}

DoubleConvW load_double(DevPool& pool, const WeightStore& ws, const std::string& p, float eps) {
    DoubleConvW d;
    d.c1 = load_conv_bn(pool, ws, p + ".double_conv.0", p + ".double_conv.1", eps);
    d.c2 = load_conv_bn(pool, ws, p + ".double_conv.3", p + ".double_conv.4", eps);
    return d;
}
DoubleConvWF load_double_f32(DevPool& pool, const WeightStore& ws, const std::string& p, float eps) {
    DoubleConvWF d;
    HostTensor fw, fb;
    fold_conv_bn(ws, p + ".double_conv.0", p + ".double_conv.1", eps, fw, fb); d.c1 = load_conv_f32(pool, fw, &fb);
    fold_conv_bn(ws, p + ".double_conv.3", p + ".double_conv.4", eps, fw, fb); d.c2 = load_conv_f32(pool, fw, &fb);
    return d;
}

Act double_conv(Ctx& c, const DoubleConvW& d, const Act& x, const Act* x2) {
    ConvOpt o; o.act = LADI_ACT_RELU;
    Act m = conv2d(c, d.c1, x, x2, o);
    return conv2d(c, d.c2, m, nullptr, o);
}

Act pool2(Ctx& c, const Act& x) {
    Act o = c.new_act(x.n, x.h / 2, x.w / 2, x.c);
    if (!c.dry()) c.check(ladi_launch_maxpool2(x.p, x.ld, x.n, x.h, x.w, x.c, o.p, o.ld, c.st), "maxpool2");
    return o;
}

Act up2(Ctx& c, const Act& x) {
    Act o = c.new_act(x.n, x.h * 2, x.w * 2, x.c);
    if (!c.dry()) c.check(ladi_launch_upsample2x_bilinear_ac(x.p, x.ld, x.n, x.h, x.w, x.c, o.p, o.ld, c.st), "upsample2x");
    return o;
}

}  // namespace

void Refine::load(const RefineCfg& c, const WeightStore& ws) {
    cfg = c;
    if (c.base % 64) throw std::runtime_error("refinement UNet: base width must be a multiple of 64 (two-source K loop)");
    inc = load_double(pool, ws, "inc", c.bn_eps);
    for (int i = 0; i < 4; ++i) down[i] = load_double(pool, ws, "down" + std::to_string(i + 1) + ".maxpool_conv.1", c.bn_eps);
    for (int i = 0; i < 4; ++i) up[i] = load_double(pool, ws, "up" + std::to_string(i + 1) + ".conv", c.bn_eps);
    outc = load_conv(pool, ws, "outc.conv");
    incf = load_double_f32(pool, ws, "inc", c.bn_eps);
    for (int i = 0; i < 4; ++i) downf[i] = load_double_f32(pool, ws, "down" + std::to_string(i + 1) + ".maxpool_conv.1", c.bn_eps);
    for (int i = 0; i < 4; ++i) upf[i] = load_double_f32(pool, ws, "up" + std::to_string(i + 1) + ".conv", c.bn_eps);
    outcf = load_conv_f32(pool, ws.get("outc.conv.weight"), ws.has("outc.conv.bias") ? &ws.get("outc.conv.bias") : nullptr);
    if (inc.c1.cin != c.in_ch || outc.cout != c.out_ch) throw std::runtime_error("refinement UNet: channel counts do not match the config");
}

int Refine::forward(const void* x, int in_f32, int B, int H, int W, void* out, int out_f32, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || (H % 16) || (W % 16)) { set_error("refinement UNet: H and W must be positive multiples of 16"); return -1; }
    if (in_f32) return forward_f32(x, B, H, W, out, out_f32, st);      // fp32 caller (inference.py:264): fp32 network (runtime_f32.cpp)
    for (int pass = 0; pass < 2; ++pass) {
        arena.dry = (pass == 0);
        if (pass == 1) arena.reserve(arena.peak);
        arena.off = 0;
        Ctx c; c.st = st; c.ar = &arena;
        Act x0 = c.new_act(B, H, W, inc.c1.cin_pad);
        if (!c.dry()) {
            if (hipMemsetAsync(x0.p, 0, x0.pixels() * (size_t)x0.ld * sizeof(h16), st) != hipSuccess) { set_error("refinement UNet: memset"); return -1; }
            c.check(ladi_launch_nchw_to_nhwc(x, in_f32, B, cfg.in_ch, H, W, x0.p, x0.ld, st), "nchw_to_nhwc");
        }
        Act x1 = double_conv(c, inc, x0, nullptr);
        Act x2 = double_conv(c, down[0], pool2(c, x1), nullptr);
        Act x3 = double_conv(c, down[1], pool2(c, x2), nullptr);
        Act x4 = double_conv(c, down[2], pool2(c, x3), nullptr);
        Act x5 = double_conv(c, down[3], pool2(c, x4), nullptr);
        // Up: cat([skip, upsampled]) -> DoubleConv; sizes match exactly (H, W multiples of 16), so F.pad (unet_parts.py:57-61) is a no-op
        Act u1 = up2(c, x5); Act y = double_conv(c, up[0], x4, &u1);
        Act u2 = up2(c, y); y = double_conv(c, up[1], x3, &u2);
        Act u3 = up2(c, y); y = double_conv(c, up[2], x2, &u3);
        Act u4 = up2(c, y); y = double_conv(c, up[3], x1, &u4);
        ConvOpt oc; oc.out_ld = 8;
        Act lg = conv2d(c, outc, y, nullptr, oc);
        if (!c.dry()) c.check(ladi_launch_nhwc_to_nchw(lg.p, lg.ld, B, cfg.out_ch, H, W, out, out_f32, st), "nhwc_to_nchw");
    }
    return 0;
}

}  // namespace ladi
