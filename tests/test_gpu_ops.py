"""Kernel-level parity (HIP through the C ABI vs plain torch fp32 on the CPU). fp16 storage / fp32 accumulation:
tolerance rel-L2 <= 2e-3 unless stated."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from ladi_vton_amd import _lib
from ladi_vton_amd._lib import ptr, stream_ptr
from tests import util as U

pytestmark = pytest.mark.gpu
TOL = 2e-3


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().float()


# --------------------------------------------------------------------------------------------------------------- igemm
NEW_RING = [39, 40, 41, 42, 43, 44, 45, 47, 48]      # round 3: one-wave-per-SIMD tiles / deep rings / interleaved DMA issue (igemm_tiles.h)
NEW_IGEMM8 = [54, 55, 56, 57, 58, 62, 63, 64, 65, 66, 67, 68]   # + the loader / consumer kernel (igemm_lc.hip)                      # round 3: more shapes of the phase-staggered 8-wave pipeline


HALO = [74, 75, 76, 77, 78, 84, 85]                            # round 3: halo-resident 3x3 convolution (igemm_halo.hip)


@pytest.mark.parametrize("cfg", [19, 20, 21, 22, 32, 33] + NEW_RING + NEW_IGEMM8 + HALO)
def test_conv3x3_eight_wave_tiles(cfg):
    """large workgroup tile shapes (8 waves, or 4 waves with one wave per SIMD), ragged pixel count, residual + statistics-free epilogue"""
    N, cin, cout, h, w = 3, 128, 320, 20, 13
    x, wt, b = _rand((N, cin, h, w), 70), _rand((cout, cin, 3, 3), 71, 1 / math.sqrt(9 * cin)), _rand((cout,), 72, 0.1)
    res = _rand((N, cout, h, w), 73)
    ref = F.conv2d(x, wt, b, padding=1) + res
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cout, bias=b, res0=U.nhwc16(res), cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL


@pytest.mark.parametrize("cfg,C,Q,hw", [(23, 320, 320, (32, 16)), (24, 320, 960, (32, 24)), (25, 320, 320, (16, 24)), (26, 640, 640, (16, 16)),
                                        (27, 640, 1920, (8, 16)), (25, 640, 96, (16, 8)), (27, 320, 160, (16, 16)),
                                        (93, 320, 320, (16, 24)), (94, 320, 960, (32, 24)), (95, 320, 160, (16, 16))])   # 93-95: two-slot ring, three workgroups per CU
def test_linear_x_stationary(cfg, C, Q, hw):
    """X-stationary linear kernel (pixel panel in registers, weights streamed): asymmetric weights catch any fragment /
    swizzle / channel-slice mix-up, random weights + bias check the arithmetic"""
    N, (H, W) = 1, hw
    x = _rand((N, C, H, W), 80)
    w = _rand((Q, C, 1, 1), 81, 1 / math.sqrt(C))
    for q in range(Q):
        w[q, (q * 7 + 3) % C, 0, 0] += 1.0 + (q % 5)
    b = _rand((Q,), 82, 0.1)
    ref = F.conv2d(x, w, b)
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(w), Q, ksize=1, bias=b, cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL


@pytest.mark.parametrize("cfg,C,Q,hw", [(25, 320, 320, (16, 16)), (26, 320, 640, (16, 24)), (27, 640, 640, (16, 8)), (25, 640, 64, (8, 16)),
                                        (27, 320, 160, (16, 8))])
def test_linear_x_stationary_residual(cfg, C, Q, hw):
    """residual tile arriving by LDS-DMA one channel block ahead (double-buffered landing patch, counted vmcnt hand-over);
    must agree BIT-EXACTLY with the tiled igemm epilogue (same rounding points)"""
    N, (H, W) = 1, hw
    x = _rand((N, C, H, W), 90)
    w = _rand((Q, C, 1, 1), 91, 1 / math.sqrt(C))
    for q in range(Q):
        w[q, (q * 11 + 5) % C, 0, 0] += 1.0 + (q % 3)
    b, res = _rand((Q,), 92, 0.1), _rand((N, Q, H, W), 93)
    ref = F.conv2d(x, w, b) + res
    xs, ws, rs = U.nhwc16(x), U.pack_conv_weight(w), U.nhwc16(res)
    y = U.igemm(xs, ws, Q, ksize=1, bias=b, res0=rs, cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    y_tiled = U.igemm(xs, ws, Q, ksize=1, bias=b, res0=rs, cfg=3)
    assert torch.equal(y, y_tiled)


@pytest.mark.parametrize("cfg,C,Qh,T", [(25, 320, 256, 256), (26, 320, 1280, 128), (27, 640, 320, 384), (25, 640, 32, 128),
                                        (93, 320, 256, 256), (94, 320, 1280, 128), (95, 320, 1280, 3072)])
def test_linear_x_stationary_geglu(cfg, C, Qh, T):
    """GEGLU up-projection: interleaved u | g weight blocks, out = (u + bu) * gelu(g + bg)"""
    x, w, b = _rand((1, C, T, 1), 94), _rand((2 * Qh, C), 95, 1 / math.sqrt(C)), _rand((2 * Qh,), 96, 0.1)
    t = x[0, :, :, 0].t()
    u, g = F.linear(t, w, b).chunk(2, -1)
    ref = u * F.gelu(g)
    wi, bi = torch.zeros_like(w), torch.zeros_like(b)
    for j in range(Qh):
        blk, i = divmod(j, 32)
        wi[blk * 64 + i], wi[blk * 64 + 32 + i] = w[j], w[Qh + j]
        bi[blk * 64 + i], bi[blk * 64 + 32 + i] = b[j], b[Qh + j]
    y = U.igemm(U.nhwc16(x), wi.half().contiguous().to(U.dev()), 2 * Qh, ksize=1, bias=bi, act="geglu", cfg=cfg)
    assert U.rel_l2(y.float().cpu().reshape(T, Qh), ref) < TOL


@pytest.mark.parametrize("n,T,L", [(2, 256, 77), (1, 3072, 77), (3, 128, 96), (2, 128, 5)])
def test_fused_cross_attention_block(n, T, L):
    """attn2 of a BasicTransformerBlock on the 320-channel level as ONE launch (xf_fused.hip: LayerNorm -> to_q -> attention over the L
    context rows -> to_out + bias + residual; the accumulator blocks of each product are the B operand of the next) against torch; repeated
    launches bit-equal (counted waits of the four-slot weight ring)."""
    lib = _lib.load()
    C, H = 320, 5
    x = _rand((n * T, C), 400, 1.5)
    gamma, beta = 1.0 + _rand((C,), 401, 0.2), _rand((C,), 402, 0.2)
    wq, wk, wv = (_rand((C, C), 403 + i, 1 / math.sqrt(C)) for i in range(3))
    wo, bo = _rand((C, C), 406, 1 / math.sqrt(C)), _rand((C,), 407, 0.2)
    ctx = _rand((n, L, 1024), 408)
    wkc, wvc = _rand((C, 1024), 409, 1 / 32.0), _rand((C, 1024), 410, 1 / 32.0)
    k, v = F.linear(ctx, wkc).half().float(), F.linear(ctx, wvc).half().float()          # what the cached kv projection holds (fp16)
    xn = F.layer_norm(x, (C,), gamma.half().float(), beta.half().float(), 1e-5).half().float()
    q = F.linear(xn, wq).half().float().view(n, T, H, 64).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k.view(n, L, H, 64).transpose(1, 2), v.view(n, L, H, 64).transpose(1, 2))
    o = o.transpose(1, 2).reshape(n * T, C).half().float()
    ref = x + F.linear(o, wo, bo)
    dev = U.dev()
    d = lambda t: t.half().contiguous().to(dev)
    X, KV = d(x), d(torch.cat([k, v], -1))
    G, B, WQ, WO, BO = d(gamma), d(beta), d(wq), d(wo), d(bo)
    out = torch.empty_like(X)

    def run():
        rc = lib.ladi_op_xattn_block(ptr(X), ptr(G), ptr(B), 1e-5, ptr(WQ), ptr(KV), L, ptr(WO), ptr(BO), n, T, ptr(out), stream_ptr())
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        return out.clone()
    y = run()
    assert U.rel_l2(y.float().cpu(), ref) < TOL, (n, T, L, U.rel_l2(y.float().cpu(), ref))
    for _ in range(3):
        assert torch.equal(run(), y)
    # unsupported shapes are refused
    assert lib.ladi_op_xattn_block(ptr(X), ptr(G), ptr(B), 1e-5, ptr(WQ), ptr(KV), 97, ptr(WO), ptr(BO), n, T, ptr(out), stream_ptr()) != 0
    assert lib.ladi_op_xattn_block(ptr(X), ptr(G), ptr(B), 1e-5, ptr(WQ), ptr(KV), L, ptr(WO), ptr(BO), n, 96, ptr(out), stream_ptr()) != 0


@pytest.mark.parametrize("P,pipe", [(256, "1"), (3072, "1"), (128, "0"), (1024, "0")])
def test_fused_feed_forward_block(P, pipe, monkeypatch):
    """the feed-forward of a BasicTransformerBlock on the 320-channel level as ONE launch (LayerNorm -> GEGLU 320 -> 2 x 1280 -> 1280 -> 320 + bias
    + residual; the hidden tensor never exists) against torch, both loop forms (LADI_FF_PIPE, read per launch)."""
    monkeypatch.setenv("LADI_FF_PIPE", pipe)
    lib = _lib.load()
    C, HID = 320, 1280
    x = _rand((P, C), 420, 1.5)
    gamma, beta = 1.0 + _rand((C,), 421, 0.2), _rand((C,), 422, 0.2)
    w1, b1 = _rand((2 * HID, C), 423, 1 / math.sqrt(C)), _rand((2 * HID,), 424, 0.1)
    w2, bo = _rand((C, HID), 425, 1 / math.sqrt(HID)), _rand((C,), 426, 0.2)
    xn = F.layer_norm(x, (C,), gamma.half().float(), beta.half().float(), 1e-5).half().float()
    u, g = F.linear(xn, w1, b1).chunk(2, -1)
    h = (u * F.gelu(g)).half().float()
    ref = x + F.linear(h, w2, bo)
    wi, bi = torch.zeros_like(w1), torch.zeros_like(b1)
    for j in range(HID):
        blk, i = divmod(j, 32)
        wi[blk * 64 + i], wi[blk * 64 + 32 + i] = w1[j], w1[HID + j]
        bi[blk * 64 + i], bi[blk * 64 + 32 + i] = b1[j], b1[HID + j]
    dev = U.dev()
    d = lambda t: t.half().contiguous().to(dev)
    X, G, B, W1, B1, W2, BO = d(x), d(gamma), d(beta), d(wi), d(bi), d(w2), d(bo)
    out = torch.empty_like(X)

    def run():
        rc = lib.ladi_op_ff_block(ptr(X), ptr(G), ptr(B), 1e-5, ptr(W1), ptr(B1), ptr(W2), ptr(BO), P, ptr(out), stream_ptr())
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        return out.clone()
    y = run()
    assert U.rel_l2(y.float().cpu(), ref) < TOL, (P, pipe, U.rel_l2(y.float().cpu(), ref))
    for _ in range(3):
        assert torch.equal(run(), y)
    assert lib.ladi_op_ff_block(ptr(X), ptr(G), ptr(B), 1e-5, ptr(W1), ptr(B1), ptr(W2), ptr(BO), 100, ptr(out), stream_ptr()) != 0


def test_linear_x_stationary_rejects_unsupported():
    """explicitly requested on a shape / epilogue it does not cover -> error, never a silent wrong answer"""
    x = _rand((1, 320, 8, 8), 83)            # 64 pixels: not a whole 128-pixel panel
    w = _rand((320, 320, 1, 1), 84, 0.05)
    with pytest.raises(Exception):
        U.igemm(U.nhwc16(x), U.pack_conv_weight(w), 320, ksize=1, cfg=25)
    x = _rand((1, 320, 16, 16), 85)
    with pytest.raises(Exception):                          # residual only with 32 pixels per wave
        U.igemm(U.nhwc16(x), U.pack_conv_weight(w), 320, ksize=1, res0=U.nhwc16(x), cfg=23)
    with pytest.raises(Exception):                          # no fused activation other than GEGLU
        U.igemm(U.nhwc16(x), U.pack_conv_weight(w), 320, ksize=1, act="silu", cfg=25)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 9, 16, 17, 18, 19, 20, 21, 22, 32, 33] + NEW_RING + NEW_IGEMM8)
def test_mfma_layout_asymmetric(cfg):
    """transpose-detecting check of the MFMA fragment / accumulator mapping: 1x1 'conv' with an asymmetric weight."""
    N, H, W, C, Q = 1, 16, 24, 64, 192
    x = _rand((N, C, H, W), 1)
    w = torch.zeros(Q, C, 1, 1)
    for q in range(Q):
        w[q, (q * 7) % C, 0, 0] = 1.0 + (q % 5)      # each output channel picks one (scaled) input channel
        w[q, (q * 3 + 1) % C, 0, 0] += 0.5
    ref = F.conv2d(x, w)
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(w), Q, ksize=1, cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < 1e-3


@pytest.mark.parametrize("cin,cout,h,w,cfg", [(64, 64, 16, 12, 0), (128, 320, 24, 16, 0), (320, 128, 20, 12, 1), (192, 64, 32, 24, 2),
                                              (64, 192, 9, 7, 3), (128, 128, 13, 5, 4), (64, 320, 17, 9, 2), (128, 256, 11, 7, 6), (192, 128, 16, 20, 5)])
def test_conv3x3_bias_res_temb(cin, cout, h, w, cfg):
    N = 2
    x, wt, b = _rand((N, cin, h, w), 2), _rand((cout, cin, 3, 3), 3, 1 / math.sqrt(9 * cin)), _rand((cout,), 4, 0.1)
    temb, res = _rand((cout,), 5), _rand((N, cout, h, w), 6)
    ref = F.conv2d(x, wt, b, padding=1) + temb[None, :, None, None] + res
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cout, bias=b, rowadd=temb, res0=U.nhwc16(res), cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL


@pytest.mark.parametrize("cfg", [11, 12, 13, 14, 15, 34, 35, 36, 37, 38, 46, 49, 50, 51, 52, 53, 59, 60, 61, 69, 70, 71, 72, 73, 79, 80, 81, 82, 83, 86, 87, 90, 91])
def test_conv3x3_split_k(cfg):
    """split-K variants on a few-tile / deep-K problem, in both forms: the in-launch combine (round 4: slices publish fp32 slabs with
    write-through stores, the last-arriving slice of a tile sums them in slice order and runs the fused epilogue) and the separate reduce
    pass of rounds 1-3.  Both against fp32 torch; the in-launch form must be bit-identical over repeated launches whichever slice arrives
    last (race screen of the ticket hand-off)."""
    lib = _lib.load()
    N, cin, cout, h, w = 2, 512, 192, 8, 6
    x, wt, b = _rand((N, cin, h, w), 60), _rand((cout, cin, 3, 3), 61, 1 / math.sqrt(9 * cin)), _rand((cout,), 62, 0.1)
    temb, res = _rand((cout,), 63), _rand((N, cout, h, w), 64)
    ref = F.silu(F.conv2d(x, wt, b, padding=1) + temb[None, :, None, None]) + res
    X, Wp, R = U.nhwc16(x), U.pack_conv_weight(wt), U.nhwc16(res)
    try:
        lib.ladi_igemm_set_splitk_two_pass(1)
        y2 = U.igemm(X, Wp, cout, bias=b, rowadd=temb, act="silu", res0=R, cfg=cfg)
        assert U.rel_l2(U.to_nchw(y2), ref) < TOL
    finally:
        lib.ladi_igemm_set_splitk_two_pass(0)
    y = U.igemm(X, Wp, cout, bias=b, rowadd=temb, act="silu", res0=R, cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    assert U.rel_l2(y.float().cpu(), y2.float().cpu()) < 1e-3
    for _ in range(8):
        assert torch.equal(U.igemm(X, Wp, cout, bias=b, rowadd=temb, act="silu", res0=R, cfg=cfg), y)


@pytest.mark.parametrize("cfg", [87, 91, 109, 83, 80, 13, 38])
def test_split_k_in_launch_combine_at_the_8x6_level(cfg):
    """the launch population the in-launch combine was built for: the 1280 -> 1280 3x3 convolution of the 8x6 level at the bench batch
    (768 pixels: 60-120 tiles x 4-8 K slices spread over every XCD), residual epilogue, 30 launches back to back -- every output must equal
    the first (a stale slab or a counter that was not re-armed shows up here) and the two-pass form within fp16 rounding"""
    lib = _lib.load()
    N, cin, cout, h, w = 16, 1280, 1280, 8, 6
    x, wt, b = _rand((N, cin, h, w), 160), _rand((cout, cin, 3, 3), 161, 1 / math.sqrt(9 * cin)), _rand((cout,), 162, 0.1)
    res = _rand((N, cout, h, w), 164)
    X, Wp, R = U.nhwc16(x), U.pack_conv_weight(wt), U.nhwc16(res)
    y = U.igemm(X, Wp, cout, bias=b, res0=R, cfg=cfg)
    ref = F.conv2d(x[:2], wt, b, padding=1) + res[:2]
    assert U.rel_l2(U.to_nchw(y[:2]), ref) < TOL
    for _ in range(30):
        assert torch.equal(U.igemm(X, Wp, cout, bias=b, res0=R, cfg=cfg), y)
    try:
        lib.ladi_igemm_set_splitk_two_pass(1)
        y2 = U.igemm(X, Wp, cout, bias=b, res0=R, cfg=cfg)
    finally:
        lib.ladi_igemm_set_splitk_two_pass(0)
    assert U.rel_l2(y.float().cpu(), y2.float().cpu()) < 1e-3


@pytest.mark.parametrize("cfg", [32, 33] + NEW_IGEMM8 + [3, 7, 9, 39, 40, 42, 45, 47])
def test_igemm8_staggered_pipeline_shapes(cfg):
    """the phase-staggered large-tile kernel (igemm8.hip) -- and the ring kernel with its hoisted DMA addressing (uniform soffset per tap,
    halo validity masks) -- on every gather variant and K-loop length it has to pipeline: a single K tile
    (prologue only), odd / even K-tile counts, two-source concat, stride 2, folded 2x upsample, ragged pixel and channel tiles"""
    # 1x1, K = 64: one K tile; K = 192: three
    for cin, cout, hw in ((64, 96, (9, 7)), (192, 320, (24, 16)), (128, 700, (40, 13))):
        x, w, b = _rand((2, cin) + hw, 80), _rand((cout, cin, 1, 1), 81, 1 / math.sqrt(cin)), _rand((cout,), 82, 0.1)
        y = U.igemm(U.nhwc16(x), U.pack_conv_weight(w), cout, ksize=1, bias=b, cfg=cfg)
        assert U.rel_l2(U.to_nchw(y), F.conv2d(x, w, b)) < TOL, (cin, cout)
    # two-source concat 3x3 (K tiles walk src0 then src1 inside every tap), SiLU epilogue, residual
    N, c0, c1, cout, h, w_ = 2, 128, 64, 320, 20, 14
    xa, xb = _rand((N, c0, h, w_), 83), _rand((N, c1, h, w_), 84)
    wt, b = _rand((cout, c0 + c1, 3, 3), 85, 1 / math.sqrt(9 * (c0 + c1))), _rand((cout,), 86, 0.1)
    res = _rand((N, cout, h, w_), 87)
    ref = F.silu(F.conv2d(torch.cat([xa, xb], 1), wt, b, padding=1)) + res
    y = U.igemm(U.nhwc16(xa), U.pack_conv_weight(wt), cout, x2=U.nhwc16(xb), bias=b, act="silu", res0=U.nhwc16(res), cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    # stride 2 (pad 1) and the folded nearest-2x upsample
    x, wt = _rand((2, 128, 18, 12), 88), _rand((256, 128, 3, 3), 89, 1 / math.sqrt(9 * 128))
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), 256, stride=2, pad=1, cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), F.conv2d(x, wt, stride=2, padding=1)) < TOL
    if cfg in range(62, 74):      # the loader / consumer kernel does not implement the folded upsample: it must refuse, not mis-compute
        with pytest.raises(AssertionError):
            U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), 256, ups=1, cfg=cfg)
        return
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), 256, ups=1, cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, padding=1)) < TOL


@pytest.mark.parametrize("cfg", [32, 33, 35, 54, 56, 57, 39, 40, 42, 44, 47, 62, 63, 64, 66, 68])
def test_igemm8_is_race_free_and_deterministic(cfg):
    """race screen for the counted-vmcnt / staggered-barrier pipeline: a UNet-sized conv (many workgroups, 45 K tiles) repeated 25 times
    must give bitwise identical outputs, and match the plain-tile kernel to fp16 rounding"""
    N, cin, cout, h, w_ = 4, 320, 320, 32, 24
    x, wt, b = _rand((N, cin, h, w_), 90), _rand((cout, cin, 3, 3), 91, 1 / math.sqrt(9 * cin)), _rand((cout,), 92, 0.1)
    X, Wp = U.nhwc16(x), U.pack_conv_weight(wt)
    first = U.igemm(X, Wp, cout, bias=b, cfg=cfg)
    for _ in range(24):
        assert torch.equal(U.igemm(X, Wp, cout, bias=b, cfg=cfg), first)
    base = U.igemm(X, Wp, cout, bias=b, cfg=7)
    assert U.rel_l2(first.float().cpu(), base.float().cpu()) < 1e-3
    assert U.rel_l2(U.to_nchw(first), F.conv2d(x, wt, b, padding=1)) < TOL


@pytest.mark.parametrize("cfg", [7, 9, 32, 39, 40, 45, 54, 56, 62, 64, 65, 68, 74, 76, 77, 84, 85, 92, 96])
def test_fused_output_statistics(cfg):
    """per-channel partial statistics of the output (sum, sum of squares of the fp16-rounded values) written by the epilogue for the
    consuming GroupNorm: rows of [Q][2] per (TP*32)-pixel block -- 96-pixel blocks for the 320x192 / 256x192 tiles.  Summed over all rows
    they must equal the per-channel totals of the stored output."""
    lib = _lib.load()
    N, cin, cout, h, w_ = 2, 128, 320, 24, 16                   # 384 pixels per sample: a multiple of 64, 96 and 128
    x, wt, b = _rand((N, cin, h, w_), 95), _rand((cout, cin, 3, 3), 96, 1 / math.sqrt(9 * cin)), _rand((cout,), 97, 0.1)
    X, Wp, B16 = U.nhwc16(x), U.pack_conv_weight(wt), b.half().to(U.dev())
    out = torch.zeros((N, h, w_, cout), dtype=torch.float16, device=U.dev())
    rows = N * h * w_ // 32
    stats = torch.zeros((rows, cout, 2), dtype=torch.float32, device=U.dev())
    d = _lib.IGemmDesc()
    d.src0, d.C0, d.ld0 = X.data_ptr(), cin, cin
    d.Hs, d.Ws, d.Ho, d.Wo, d.P = h, w_, h, w_, N * h * w_
    d.ksize, d.stride, d.pad, d.ups = 3, 1, 1, 0
    d.W, d.Q, d.K, d.ldw = Wp.data_ptr(), cout, 9 * cin, 0
    d.bias, d.act, d.out_scale = B16.data_ptr(), U.ACT["silu"], 1.0
    d.out, d.ldo, d.stats = out.data_ptr(), cout, stats.data_ptr()
    assert lib.ladi_op_igemm(ctypes.byref(d), 1, cfg, stream_ptr()) == 0, _lib.last_error()
    torch.cuda.synchronize()
    ref = F.silu(F.conv2d(x, wt, b, padding=1))
    assert U.rel_l2(U.to_nchw(out), ref) < TOL
    o = out.float().reshape(-1, cout)
    tot = stats.sum(0).cpu()
    assert U.rel_l2(tot[:, 0], o.sum(0).cpu()) < 1e-4 and U.rel_l2(tot[:, 1], (o * o).sum(0).cpu()) < 1e-4


HALO_W24 = [88, 89, 90, 91, 97, 98, 99, 109]     # halo buffer sized for rows of <= 24 pixels (third weight slot at two workgroups per CU)


@pytest.mark.parametrize("cfg", HALO + [79, 80, 86] + HALO_W24 + [92, 96])
def test_conv3x3_halo_resident(cfg):
    """the halo-resident 3x3 convolution: the nine taps are applied when the B fragments are read from ONE staged pixel range, so what
    must be right is the tap shift / validity logic at image edges (left / right wrap-around, top / bottom rows, sample boundaries), the
    two-source K loop, ragged tiles, several channel chunks (double-buffered halo tile) and the widest supported image (W = 48).
    Repeated runs must be bit-identical (race screen of the counted waits)."""
    lib = _lib.load()
    for (N, c0, c1, cout, h, w_, act, seed) in ((3, 128, 0, 320, 20, 13, "none", 70), (2, 192, 64, 192, 9, 48, "silu", 74), (5, 64, 0, 96, 6, 6, "none", 78),
                                               (1, 320, 0, 320, 64, 48, "none", 82), (2, 128, 128, 320, 32, 24, "silu", 86)):
        if cfg in HALO_W24 and w_ > 24:
            continue
        xa = _rand((N, c0, h, w_), seed)
        xb = _rand((N, c1, h, w_), seed + 1) if c1 else None
        wt, b = _rand((cout, c0 + c1, 3, 3), seed + 2, 1 / math.sqrt(9 * (c0 + c1))), _rand((cout,), seed + 3, 0.1)
        res = _rand((N, cout, h, w_), seed + 4)
        xin = torch.cat([xa, xb], 1) if c1 else xa
        ref = F.conv2d(xin, wt, b, padding=1)
        ref = (F.silu(ref) if act == "silu" else ref) + res
        args = dict(bias=b, act=act, res0=U.nhwc16(res), cfg=cfg)
        if c1:
            args["x2"] = U.nhwc16(xb)
        Xa, Wp = U.nhwc16(xa), U.pack_conv_weight(wt)
        y = U.igemm(Xa, Wp, cout, **args)
        assert U.rel_l2(U.to_nchw(y), ref) < TOL, (cfg, N, c0, c1, cout, h, w_)
        for _ in range(5):
            assert torch.equal(U.igemm(Xa, Wp, cout, **args), y)
    # not a 3x3 stride-1 convolution / too wide an image: refused, never mis-computed
    x, w1 = _rand((1, 64, 8, 8), 90), _rand((64, 64, 1, 1), 91)
    with pytest.raises(AssertionError):
        U.igemm(U.nhwc16(x), U.pack_conv_weight(w1), 64, ksize=1, cfg=cfg)
    xw, w3 = _rand((1, 64, 4, 64), 92), _rand((64, 64, 3, 3), 93)
    with pytest.raises(AssertionError):
        U.igemm(U.nhwc16(xw), U.pack_conv_weight(w3), 64, cfg=cfg)
    if cfg in HALO_W24:
        xw = _rand((1, 64, 4, 32), 94)
        with pytest.raises(AssertionError):
            U.igemm(U.nhwc16(xw), U.pack_conv_weight(w3), 64, cfg=cfg)


HALO2D = [100, 101, 102, 103]     # round 5: 2-D blocked halo tiles (th x 32 pixel blocks) for images wider than 48 pixels


@pytest.mark.parametrize("cfg", HALO2D)
def test_conv3x3_halo_2d_blocked(cfg):
    """the 2-D blocked halo form (igemm_halo.hip G2D): blocks of th image rows x 32 columns staged with their one-pixel frame, taps as linear
    shifts inside the (th + 2) x 34 block, sub-tiles written one IMAGE ROW apart.  What must be right: block -> (sample, row, column) mapping
    over several samples, the zero frame at all four image edges and between samples, the two-source K loop over several channel chunks,
    residual / mask / activation in the row-strided epilogue, the output statistics (rows per 64 pixels), bit-equal repeats; shapes that are
    not whole blocks are refused."""
    lib = _lib.load()
    th = 4 if cfg == 103 else 8
    for (N, c0, c1, cout, h, w_, act, seed) in ((2, 128, 0, 128, 16, 64, "none", 170), (1, 64, 64, 320, 8, 96, "silu", 174), (3, 256, 0, 256, 24, 32, "none", 178),
                                               (2, 64, 0, 96, 2 * th, 160, "silu", 182)):
        xa = _rand((N, c0, h, w_), seed)
        xb = _rand((N, c1, h, w_), seed + 1) if c1 else None
        wt, b = _rand((cout, c0 + c1, 3, 3), seed + 2, 1 / math.sqrt(9 * (c0 + c1))), _rand((cout,), seed + 3, 0.1)
        res = _rand((N, cout, h, w_), seed + 4)
        mask = (torch.rand((N, 1, h, w_), generator=torch.Generator().manual_seed(seed + 5)) > 0.5).float()
        xin = torch.cat([xa, xb], 1) if c1 else xa
        ref = F.conv2d(xin, wt, b, padding=1)
        ref = ((F.silu(ref) if act == "silu" else ref) + res) * (1.0 - mask)
        M = mask.permute(0, 2, 3, 1).reshape(-1).half().contiguous().to(U.dev())
        args = dict(bias=b, act=act, res0=U.nhwc16(res), mask=M, cfg=cfg)
        if c1:
            args["x2"] = U.nhwc16(xb)
        Xa, Wp = U.nhwc16(xa), U.pack_conv_weight(wt)
        y = U.igemm(Xa, Wp, cout, **args)
        assert U.rel_l2(U.to_nchw(y)[:, :cout], ref) < TOL, (cfg, N, c0, c1, cout, h, w_)
        for _ in range(3):
            assert torch.equal(U.igemm(Xa, Wp, cout, **args), y)
    # output statistics through the row-strided epilogue: all rows of a sample sum to the per-channel totals of what was stored
    N, cin, cout, h, w_ = 2, 128, 128, 16, 64
    x, wt, b = _rand((N, cin, h, w_), 190), _rand((cout, cin, 3, 3), 191, 1 / math.sqrt(9 * cin)), _rand((cout,), 192, 0.1)
    X, Wp, B16 = U.nhwc16(x), U.pack_conv_weight(wt), b.half().to(U.dev())
    out = torch.zeros((N, h, w_, cout), dtype=torch.float16, device=U.dev())
    rows = N * h * w_ // 32
    stats = torch.zeros((rows, cout, 2), dtype=torch.float32, device=U.dev())
    d = _lib.IGemmDesc()
    d.src0, d.C0, d.ld0 = X.data_ptr(), cin, cin
    d.Hs, d.Ws, d.Ho, d.Wo, d.P = h, w_, h, w_, N * h * w_
    d.ksize, d.stride, d.pad, d.ups = 3, 1, 1, 0
    d.W, d.Q, d.K, d.ldw = Wp.data_ptr(), cout, 9 * cin, 0
    d.bias, d.act, d.out_scale = B16.data_ptr(), U.ACT["none"], 1.0
    d.out, d.ldo, d.stats = out.data_ptr(), cout, stats.data_ptr()
    assert lib.ladi_op_igemm(ctypes.byref(d), 1, cfg, stream_ptr()) == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert U.rel_l2(U.to_nchw(out), F.conv2d(x, wt, b, padding=1)) < TOL
    o = out.float().reshape(N, -1, cout)
    rps = h * w_ // 64                                   # one statistics row per wave = two 32-pixel row segments
    st = stats[:N * rps].reshape(N, rps, cout, 2).sum(1).cpu()
    assert U.rel_l2(st[..., 0], o.sum(1).cpu()) < 1e-4 and U.rel_l2(st[..., 1], (o * o).sum(1).cpu()) < 1e-4
    # not whole blocks (W % 32, H % th) / not a stride-1 3x3: refused, never mis-computed
    w3 = _rand((64, 64, 3, 3), 193)
    for shp in ((1, 64, th, 48), (1, 64, th + 1, 32)):
        with pytest.raises(AssertionError):
            U.igemm(U.nhwc16(_rand(shp, 194)), U.pack_conv_weight(w3), 64, cfg=cfg)
    with pytest.raises(AssertionError):
        U.igemm(U.nhwc16(_rand((1, 64, 8, 32), 195)), U.pack_conv_weight(_rand((64, 64, 1, 1), 196)), 64, ksize=1, cfg=cfg)


def test_conv3x3_stride2_pad1_and_asym():
    N, cin, cout, h, w = 2, 64, 128, 16, 12
    x, wt, b = _rand((N, cin, h, w), 7), _rand((cout, cin, 3, 3), 8, 0.05), _rand((cout,), 9, 0.1)
    ref = F.conv2d(x, wt, b, stride=2, padding=1)                       # UNet Downsample2D
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cout, stride=2, pad=1, bias=b)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, b, stride=2, padding=0)  # VAE Downsample2D(padding=0)
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cout, stride=2, pad=0, bias=b)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL


def test_conv3x3_upsample_fold():
    N, cin, cout, h, w = 1, 128, 64, 8, 6
    x, wt, b = _rand((N, cin, h, w), 10), _rand((cout, cin, 3, 3), 11, 0.05), _rand((cout,), 12, 0.1)
    skip = _rand((N, cout, 2 * h, 2 * w), 13)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, b, padding=1) + skip
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cout, ups=1, bias=b, res0=U.nhwc16(skip))
    assert U.rel_l2(U.to_nchw(y), ref) < TOL


@pytest.mark.parametrize("cfg,N,cin,cout,hw", [(104, 2, 128, 128, (8, 12)),      # 128x192 tiles: 16x24 output = 384 pixels per sample, two tiles each
                                               (104, 1, 192, 320, (12, 24)),     # 24x48 output rows (the widest the form takes), ragged channel tile
                                               (106, 2, 128, 64, (16, 12)),      # 128x128 tiles, tile boundaries in the middle of image rows
                                               (106, 3, 64, 128, (4, 8)),        # 8x16 output = exactly one tile per sample
                                               (105, 1, 128, 320, (12, 8)),      # twelve-wave 320x192 tile, 24x16 output = two tiles
                                               (107, 2, 1152, 128, (8, 12)),     # cfg 104 + split-K 2 (K = 10 368: 162 steps, slices enter chunks mid-way)
                                               (108, 2, 1280, 128, (8, 8))])     # cfg 106 + split-K 2
def test_conv3x3_upsample_fold_halo_forms(cfg, N, cin, cout, hw):
    """round 6: the halo kernel's folded-upsample forms (igemm_halo_kernel.h UPS = 1: the LOW-resolution rows a tile touches are staged once per
    channel chunk, the tap is a per-lane (row, column) table look-up) against interpolate(nearest, x2) -> conv2d, with bias + residual; first /
    last rows and columns (out-of-image taps), tiles that start in the middle of an image row, several samples, split-K; repeat launches bit-equal"""
    h, w = hw
    x, wt, b = _rand((N, cin, h, w), 110), _rand((cout, cin, 3, 3), 111, 0.05), _rand((cout,), 112, 0.1)
    skip = _rand((N, cout, 2 * h, 2 * w), 113)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, b, padding=1) + skip
    xs, ws, rs = U.nhwc16(x), U.pack_conv_weight(wt), U.nhwc16(skip)
    y = U.igemm(xs, ws, cout, ups=1, bias=b, res0=rs, cfg=cfg)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    y2 = U.igemm(xs, ws, cout, ups=1, bias=b, res0=rs, cfg=cfg)
    assert torch.equal(y, y2)
    y_ring = U.igemm(xs, ws, cout, ups=1, bias=b, res0=rs, cfg=7)      # the ring kernel's per-lane gather of the same layer
    assert U.rel_l2(U.to_nchw(y), U.to_nchw(y_ring).float()) < TOL


def test_conv3x3_upsample_fold_halo_refuses_what_it_cannot_do():
    """two sources, an output row wider than 48 pixels, or a sample that is not whole tiles: the folded-upsample halo forms must refuse (rc != 0)"""
    lib = __import__("ladi_vton_amd")._lib.load()
    x, wt = _rand((1, 128, 12, 32), 114), _rand((64, 128, 3, 3), 115, 0.05)
    with pytest.raises(AssertionError):
        U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), 64, ups=1, cfg=104)             # 64-pixel output rows
    x = _rand((1, 128, 6, 6), 116)
    with pytest.raises(AssertionError):
        U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), 64, ups=1, cfg=106)             # 144 output pixels per sample: not whole 128-pixel tiles
    del lib


def test_conv_concat_two_sources_and_1x1():
    N, c0, c1, cout, h, w = 2, 128, 64, 128, 12, 8
    a, b2 = _rand((N, c0, h, w), 14), _rand((N, c1, h, w), 15)
    wt, b = _rand((cout, c0 + c1, 3, 3), 16, 0.03), _rand((cout,), 17, 0.1)
    ref = F.conv2d(torch.cat([a, b2], 1), wt, b, padding=1)
    y = U.igemm(U.nhwc16(a), U.pack_conv_weight(wt), cout, x2=U.nhwc16(b2), bias=b)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    w1 = _rand((cout, c0 + c1, 1, 1), 18, 0.08)
    ref = F.conv2d(torch.cat([a, b2], 1), w1, b)
    y = U.igemm(U.nhwc16(a), U.pack_conv_weight(w1), cout, ksize=1, x2=U.nhwc16(b2), bias=b)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL


def test_conv_small_cin_cout_and_mask_silu():
    # conv_in-like (31 -> 64, zero-padded input channels) and conv_out-like (64 -> 3 with ldo 4), SiLU and (1-mask) epilogues
    N, h, w = 1, 16, 12
    x, wt, b = _rand((N, 31, h, w), 19), _rand((64, 31, 3, 3), 20, 0.06), _rand((64,), 21, 0.1)
    ref = F.silu(F.conv2d(x, wt, b, padding=1))
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), 64, bias=b, act="silu")
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    x2, w2, b2 = _rand((N, 64, h, w), 22), _rand((3, 64, 3, 3), 23, 0.05), _rand((3,), 24, 0.1)
    mask = (torch.rand((N, 1, h, w), generator=torch.Generator().manual_seed(25)) > 0.5).float()
    ref = F.conv2d(x2, w2, b2, padding=1) * (1 - mask)
    y = U.igemm(U.nhwc16(x2), U.pack_conv_weight(w2), 3, bias=b2, mask=mask.reshape(-1).half().to(U.dev()), out_ld=4)
    assert U.rel_l2(U.to_nchw(y, 3), ref) < TOL


@pytest.mark.parametrize("cfg", [0, 1, 3, 4, 6, 33])
def test_linear_geglu(cfg):
    T, C = 200, 128
    x, w, b = _rand((1, C, T, 1), 26), _rand((8 * C, C), 27, 1 / math.sqrt(C)), _rand((8 * C,), 28, 0.1)
    t = x[0, :, :, 0].t()
    u, g = F.linear(t, w, b).chunk(2, -1)
    ref = u * F.gelu(g)
    half = 4 * C
    wi, bi = torch.zeros_like(w), torch.zeros_like(b)
    for j in range(half):
        blk, i = divmod(j, 32)
        wi[blk * 64 + i], wi[blk * 64 + 32 + i] = w[j], w[half + j]
        bi[blk * 64 + i], bi[blk * 64 + 32 + i] = b[j], b[half + j]
    y = U.igemm(U.nhwc16(x), wi.half().contiguous().to(U.dev()), 8 * C, ksize=1, bias=bi, act="geglu", cfg=cfg)
    assert U.rel_l2(y.float().cpu().reshape(T, half), ref) < TOL


def test_gemm_f32_out():
    T, C = 192, 128
    q, k = _rand((1, C, T, 1), 29), _rand((T, C), 30)
    ref = q[0, :, :, 0].t() @ k.t()
    y = U.igemm(U.nhwc16(q), k.half().to(U.dev()), T, ksize=1, out_f32=True)
    assert U.rel_l2(y.cpu().reshape(T, T), ref) < 1e-3


# --------------------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("c0,c1,silu,hw", [(64, 0, 1, (12, 10)), (320, 0, 0, (12, 10)), (128, 64, 1, (12, 10)), (640, 320, 1, (12, 10)),
                                           (320, 0, 1, (64, 48)), (1280, 1280, 1, (16, 12)), (128, 0, 1, (96, 64)), (256, 128, 0, (96, 64)),
                                           (160, 0, 1, (16, 12)), (96, 32, 0, (8, 6)), (640, 320, 1, (32, 24)), (1920, 640, 1, (8, 6))])
@pytest.mark.parametrize("onepass", ["1", "0"])
def test_group_norm(lib, c0, c1, silu, hw, onepass, monkeypatch):
    """partial rows -> (per-channel scale / shift ->) apply (+ SiLU, + add) over the virtual concat (c0 | c1), at UNet-like (64x48 ... 8x6)
    and VAE-like (96x64, many partial rows) shapes.  onepass = 1: the one-launch form (every block finalises the groups of its own 64-channel
    chunk -- groups that straddle chunk and source boundaries, a ragged last chunk) wherever it is eligible, 0: finalize + apply."""
    monkeypatch.setenv("LADI_GN_ONEPASS", onepass)
    N, (h, w), G = 2, hw, 32
    a = _rand((N, c0, h, w), 31, 2.0) + 0.5
    b2 = _rand((N, c1, h, w), 32) if c1 else None
    gam, bet = _rand((c0 + c1,), 33, 0.1) + 1, _rand((c0 + c1,), 34, 0.1)
    add = _rand((N, c0 + c1, h, w), 35)
    xcat = torch.cat([a, b2], 1) if c1 else a
    ref = F.group_norm(xcat, G, gam, bet, 1e-5)
    ref = (F.silu(ref) if silu else ref) + add
    exact = lambda t: t.permute(0, 2, 3, 1).half().contiguous().to(U.dev())       # NHWC with EXACTLY the tensor's channels (ld = C)
    A, B2, AD = exact(a), (exact(b2) if c1 else None), exact(add)
    out = torch.empty((N, h, w, c0 + c1), dtype=torch.float16, device=U.dev())
    stats = torch.empty((N * G * 2,), dtype=torch.float32, device=U.dev())
    g16, b16 = gam.half().to(U.dev()), bet.half().to(U.dev())
    rc = lib.ladi_op_group_norm(ptr(A), c0, ptr(B2), c1, N, h * w, G, ptr(g16), ptr(b16), 1e-5, silu, ptr(AD), ptr(out), ptr(stats), stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert U.rel_l2(U.to_nchw(out), ref) < TOL
    first = out.clone()
    assert lib.ladi_op_group_norm(ptr(A), c0, ptr(B2), c1, N, h * w, G, ptr(g16), ptr(b16), 1e-5, silu, ptr(AD), ptr(out), ptr(stats), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, first)           # atomics-free, fixed summation order


@pytest.mark.parametrize("C", [64, 320, 1280])
def test_layer_norm(lib, C):
    rows = 77
    x, g, b = _rand((rows, C), 36, 3.0) + 1, _rand((C,), 37, 0.1) + 1, _rand((C,), 38, 0.1)
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    X, G, B = x.half().to(U.dev()), g.half().to(U.dev()), b.half().to(U.dev())
    out = torch.empty_like(X)
    assert lib.ladi_op_layer_norm(ptr(X), ptr(G), ptr(B), 1e-5, rows, C, ptr(out), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert U.rel_l2(out.float().cpu(), ref) < TOL


def test_softmax_rows(lib):
    rows, cols = 130, 192
    s = _rand((rows, cols), 39, 4.0)
    ref = torch.softmax(s * 0.3, -1)
    S = s.to(U.dev())
    P = torch.empty((rows, cols), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_softmax_rows(ptr(S), rows, cols, 0.3, ptr(P), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert U.rel_l2(P.float().cpu(), ref) < TOL


# --------------------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("n,heads,Nq,Nk", [(2, 2, 192, 192), (1, 5, 300, 77), (2, 1, 48, 48), (1, 3, 768, 768), (1, 2, 33, 130),
                                            (4, 10, 2600, 300)])   # last: two query blocks per wave (>= 400 tiles of 256 queries), ragged both ways
def test_flash_attention(lib, n, heads, Nq, Nk):
    C = heads * 64
    q, k, v = _rand((n, Nq, C), 40), _rand((n, Nk, C), 41), _rand((n, Nk, C), 42)
    qh = q.view(n, Nq, heads, 64).transpose(1, 2)
    kh = k.view(n, Nk, heads, 64).transpose(1, 2)
    vh = v.view(n, Nk, heads, 64).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(n, Nq, C)
    Q, K, V = q.half().to(U.dev()), k.half().to(U.dev()), v.half().to(U.dev())
    O = torch.empty((n, Nq, C), dtype=torch.float16, device=U.dev())
    rc = lib.ladi_op_attention(ptr(Q), ptr(K), ptr(V), ptr(O), C, C, C, C, Nq * C, Nk * C, Nk * C, Nq * C, n, heads, Nq, Nk, 0.125, stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert U.rel_l2(O.float().cpu(), ref) < 3e-3


@pytest.mark.parametrize("n,heads,T", [(2, 2, 77), (1, 3, 200), (4, 10, 2560)])   # last: 64-queries-per-wave variant, 20 key stages
def test_flash_attention_causal(lib, n, heads, T):
    """causal mask of the CLIP text encoder: query i sees keys <= i"""
    C = heads * 64
    q, k, v = _rand((n, T, C), 240), _rand((n, T, C), 241), _rand((n, T, C), 242)
    qh, kh, vh = (t.view(n, T, heads, 64).transpose(1, 2) for t in (q, k, v))
    mask = torch.full((T, T), float("-inf")).triu_(1)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125 + mask, -1) @ vh).transpose(1, 2).reshape(n, T, C)
    Q, K, V = q.half().to(U.dev()), k.half().to(U.dev()), v.half().to(U.dev())
    O = torch.empty((n, T, C), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_attention_causal(ptr(Q), ptr(K), ptr(V), ptr(O), C, C, C, C, T * C, T * C, T * C, T * C, n, heads, T, T, 0.125, 1, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert U.rel_l2(O.float().cpu(), ref) < 3e-3
    assert U.rel_l2(O[:, :3].float().cpu(), ref[:, :3]) < 3e-3      # the first queries see 1-3 keys only


@pytest.mark.parametrize("n,heads,hd,Nq,Nk", [(2, 2, 80, 257, 257), (1, 3, 80, 17, 17), (1, 2, 64, 100, 70), (1, 1, 128, 65, 129), (2, 1, 96, 33, 200)])
def test_attention_generic_head_dim(lib, n, heads, hd, Nq, Nk):
    """MFMA attention for head dims other than 64 (CLIP ViT-H vision tower: 16 heads of 80, 257 tokens); ragged query / key tiles"""
    C = heads * hd
    q, k, v = _rand((n, Nq, C), 250), _rand((n, Nk, C), 251), _rand((n, Nk, C), 252)
    k[0, Nk - 1, :hd] = q[0, 3, :hd] * 4.0      # a late spike exercises the rescale branch
    qh = q.view(n, Nq, heads, hd).transpose(1, 2)
    kh = k.view(n, Nk, heads, hd).transpose(1, 2)
    vh = v.view(n, Nk, heads, hd).transpose(1, 2)
    sc = hd ** -0.5
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * sc, -1) @ vh).transpose(1, 2).reshape(n, Nq, C)
    Q, K, V = q.half().to(U.dev()), k.half().to(U.dev()), v.half().to(U.dev())
    O = torch.full((n, Nq, C), 7.0, dtype=torch.float16, device=U.dev())
    rc = lib.ladi_op_attention_generic(ptr(Q), ptr(K), ptr(V), ptr(O), C, C, C, C, Nq * C, Nk * C, Nk * C, Nq * C, n, heads, hd, Nq, Nk, sc, stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert U.rel_l2(O.float().cpu(), ref) < 3e-3


def test_flash_attention_forced_rescale(lib):
    """online-softmax rescale branch: one key row spikes late in the sequence (guide §5.4 rule 26)"""
    n, heads, Nq, Nk, C = 1, 1, 64, 256, 64
    q, k, v = _rand((n, Nq, C), 43), _rand((n, Nk, C), 44), _rand((n, Nk, C), 45)
    k[0, 200] = q[0, 5] * 6.0
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
    Q, K, V = q.half().to(U.dev()), k.half().to(U.dev()), v.half().to(U.dev())
    O = torch.empty((n, Nq, C), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_attention(ptr(Q), ptr(K), ptr(V), ptr(O), C, C, C, C, Nq * C, Nk * C, Nk * C, Nq * C, n, heads, Nq, Nk, 0.125, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert U.rel_l2(O.float().cpu(), ref) < 3e-3


def test_flash_attention_forced_rescale_two_query_blocks(lib):
    """same as above on the 64-queries-per-wave variant; the spike hits only the second query block of one wave"""
    n, heads, Nq, Nk, C = 8, 5, 2560, 320, 320
    q, k, v = _rand((n, Nq, C), 143), _rand((n, Nk, C), 144), _rand((n, Nk, C), 145)
    k[3, 300, 64:128] = q[3, 40, 64:128] * 6.0     # head 1, query 40 (second 32-query block of wave 0), key 300 (last stage)
    qh, kh, vh = (t.view(n, -1, heads, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(n, Nq, C)
    Q, K, V = q.half().to(U.dev()), k.half().to(U.dev()), v.half().to(U.dev())
    O = torch.empty((n, Nq, C), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_attention(ptr(Q), ptr(K), ptr(V), ptr(O), C, C, C, C, Nq * C, Nk * C, Nk * C, Nq * C, n, heads, Nq, Nk, 0.125, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert U.rel_l2(O.float().cpu(), ref) < 3e-3
    assert U.rel_l2(O[3, 32:64, 64:128].float().cpu(), ref[3, 32:64, 64:128]) < 3e-3


@pytest.mark.parametrize("mode", ["wide", "negative", "positive"])
@pytest.mark.parametrize("n,heads,Nq,Nk", [(1, 2, 96, 200), (8, 5, 2560, 384)])   # short-sequence form / 64-queries-per-wave form
def test_flash_attention_large_logits(lib, mode, n, heads, Nq, Nk):
    """the running reference of the online softmax enters the score accumulators as an exact fp16 pair (-1024 a, -b) through one MFMA
    k-step (attention.hip): scores of +-10^3 and beyond (log2 units, far outside anything the UNet produces) must neither overflow
    nor lose the reference.  Q, K, V are column slices of one [n, T, 3C] buffer (the layout of the fused QKV projection)."""
    C = heads * 64
    T = max(Nq, Nk)
    g = torch.Generator().manual_seed(77)
    buf = torch.randn((n, T, 3 * C), generator=g)
    if mode == "wide":
        buf[..., :2 * C] *= 20.0                                    # logits ~ N(0, 400^2): one-hot rows, references of both signs
    else:
        sign = -1.0 if mode == "negative" else 1.0
        buf[..., :C] = buf[..., :C].abs() * 6.0 + 12.0              # q > 0
        buf[..., C:2 * C] = sign * (buf[..., C:2 * C].abs() * 6.0 + 12.0)   # every logit ~ -+ 2500 (natural units)
    buf = buf.half()
    q, k, v = (buf[:, :Nq, :C].float(), buf[:, :Nk, C:2 * C].float(), buf[:, :Nk, 2 * C:].float())
    qh, kh, vh = (t.reshape(n, -1, heads, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax((qh.double() @ kh.double().transpose(-1, -2)) * 0.125, -1) @ vh.double()).transpose(1, 2).reshape(n, Nq, C).float()
    B = buf.to(U.dev())
    O = torch.full((n, Nq, C), 7.0, dtype=torch.float16, device=U.dev())
    e = B.element_size()
    rc = lib.ladi_op_attention(B.data_ptr(), B.data_ptr() + C * e, B.data_ptr() + 2 * C * e, ptr(O), 3 * C, 3 * C, 3 * C, C, T * 3 * C, T * 3 * C, T * 3 * C,
                               Nq * C, n, heads, Nq, Nk, 0.125, stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    out = O.float().cpu()
    assert torch.isfinite(out).all()
    # "wide": Q is pre-multiplied by scale * log2(e) and re-rounded to fp16 inside the kernel, a 2^-11 relative error per element, i.e.
    # ~0.05 on logits of several hundred: near-ties between the top keys move by a few per cent (measured 4.1e-3 overall).  At the
    # UNet's logit sizes (< 30) the same rounding is invisible (test_flash_attention: 3e-3 holds with margin).
    tol = 1e-2 if mode == "wide" else 3e-3
    assert U.rel_l2(out, ref) < tol, U.rel_l2(out, ref)


# --------------------------------------------------------------------------------------------------------------- refinement UNet helpers
def test_maxpool2_and_bilinear_upsample(lib):
    """nn.MaxPool2d(2) and nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) on NHWC fp16 (unet_parts.py:33-36,48)"""
    n, C, H, W = 2, 72, 10, 6
    x = _rand((n, C, H, W), 260)
    xs = x.permute(0, 2, 3, 1).half().contiguous().to(U.dev())      # dense NHWC, C = 72 (8-channel vectors, not a multiple of 64)
    ref_p = F.max_pool2d(x.half().float(), 2)
    out_p = torch.empty((n, H // 2, W // 2, C), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_maxpool2(ptr(xs), n, H, W, C, ptr(out_p), stream_ptr()) == 0
    ref_u = F.interpolate(x.half().float(), scale_factor=2, mode="bilinear", align_corners=True)
    out_u = torch.empty((n, 2 * H, 2 * W, C), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_upsample2x_bilinear(ptr(xs), n, H, W, C, ptr(out_u), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(U.to_nchw(out_p), ref_p)
    assert U.rel_l2(U.to_nchw(out_u), ref_u) < 1e-3


def test_conv4x4_stride2(lib):
    """4x4 stride-2 pad-1 convolution on the igemm's generic tap loop (FeatureExtraction / FeatureRegression, ConvNet_TPS.py:31,95)"""
    N, cin, cout, h, w = 2, 64, 128, 16, 12
    x, wt, b = _rand((N, cin, h, w), 270), _rand((cout, cin, 4, 4), 271, 1 / math.sqrt(16 * cin)), _rand((cout,), 272, 0.1)
    ref = F.relu(F.conv2d(x, wt, b, stride=2, padding=1))
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cout, ksize=4, stride=2, pad=1, bias=b, act="relu")
    assert y.shape[1:3] == (h // 2, w // 2)
    assert U.rel_l2(U.to_nchw(y), ref) < TOL


def test_conv3x3_relu_two_source(lib):
    """ReLU epilogue + two-source concat K loop (the Up block's cat([skip, up]) -> conv -> folded BN -> ReLU)"""
    N, c0, c1, cout, h, w = 1, 64, 64, 128, 12, 8
    x0, x1 = _rand((N, c0, h, w), 261), _rand((N, c1, h, w), 262)
    wt, b = _rand((cout, c0 + c1, 3, 3), 263, 1 / math.sqrt(9 * (c0 + c1))), _rand((cout,), 264, 0.3)
    ref = F.relu(F.conv2d(torch.cat([x0, x1], 1), wt, b, padding=1))
    y = U.igemm(U.nhwc16(x0), U.pack_conv_weight(wt), cout, x2=U.nhwc16(x1), bias=b, act="relu")
    assert U.rel_l2(U.to_nchw(y), ref) < TOL
    assert float(U.to_nchw(y).min()) == 0.0


# --------------------------------------------------------------------------------------------------------------- misc
def test_small_linear(lib):
    M, N, K = 19, 100, 320
    x, w, b, r = _rand((M, K), 46), _rand((N, K), 47, 0.05), _rand((N,), 48, 0.1), _rand((M, N), 49)
    ref = F.gelu(F.linear(F.silu(x), w, b)) + r
    X, W, B, R = x.to(U.dev()), w.half().to(U.dev()), b.half().to(U.dev()), r.half().to(U.dev())
    out = torch.empty((M, N), dtype=torch.float32, device=U.dev())
    assert lib.ladi_op_small_linear(ptr(X), 1, K, ptr(W), ptr(B), ptr(R), N, M, N, K, 2, 1, ptr(out), 1, N, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert U.rel_l2(out.cpu(), ref) < TOL


def test_layout_roundtrip(lib):
    n, C, H, W = 2, 31, 20, 12
    x = _rand((n, C, H, W), 50)
    X = x.to(U.dev())
    nh = torch.empty((n, H, W, 64), dtype=torch.float16, device=U.dev())
    back = torch.empty((n, C, H, W), dtype=torch.float32, device=U.dev())
    assert lib.ladi_op_nchw_to_nhwc(ptr(X), 0, n, C, H, W, ptr(nh), 64, stream_ptr()) == 0
    assert lib.ladi_op_nhwc_to_nchw(ptr(nh), 64, n, C, H, W, ptr(back), 0, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), x)                       # exact: values are fp16-representable
    assert float(nh[..., C:].abs().max()) == 0.0
    assert torch.equal(nh[..., :C].float().cpu(), x.permute(0, 2, 3, 1))


@pytest.mark.parametrize("kind,steps", [(0, 50), (1, 50), (0, 20), (1, 7), (2, 50), (2, 7)])
def test_scheduler_device_vs_oracle(lib, kind, steps):
    """fused CFG + DDIM / PLMS / LMS update on the device vs the oracle's scheduler; tolerance: fp32 rounding (rel <= 1e-5; LMS 1e-4: its
    weights come from a closed-form integral here and from scipy quad at epsrel 1e-4 in the oracle, as in diffusers)"""
    from oracle import pipeline as P
    sch = P.make_scheduler(kind)
    sch.set_timesteps(steps)
    evals = len(sch.timesteps)
    B, hw, gs = 2, 96, 7.5
    g = torch.Generator().manual_seed(51)
    eps = torch.randn((evals, 2 * B, hw, 4), generator=g).half()
    lat0 = torch.randn((B, hw, 4), generator=g) * sch.init_noise_sigma
    x = lat0.clone()
    for i, t in enumerate(sch.timesteps):
        e = eps[i].float()
        eu, ec = e[:B], e[B:]
        x = sch.step(eu + gs * (ec - eu), t, x)
    E, L = eps.to(U.dev()), lat0.clone().to(U.dev())
    ac = P.alphas_cumprod().contiguous()
    rc = lib.ladi_op_sched_run(kind, steps, ctypes.c_void_p(ac.data_ptr()), ptr(E), evals, B, hw, 1, gs, ptr(L), stream_ptr())
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert U.rel_l2(L.cpu(), x) < (1e-4 if kind == 2 else 1e-5)


# --------------------------------------------------------------------------------------------------------------- pipeline pre-processing (§8 a10)
@pytest.mark.parametrize("f16", [False, True])
def test_prepare_mask_and_masked_image(lib, f16):
    """diffusers prepare_mask_and_masked_image tensor branch (tryon_pipe.py:630): binarise at 0.5, masked = image * (mask < 0.5); exact"""
    B, H, W = 2, 40, 24
    g = torch.Generator().manual_seed(60)
    img = (torch.rand((B, 3, H, W), generator=g) * 2 - 1).half().float()
    mask = torch.rand((B, 1, H, W), generator=g).half().float()
    mask[0, 0, 0, :4] = torch.tensor([0.5, 0.4999, 0.0, 1.0]).half().float()      # threshold edge: 0.5 -> 1
    dt = torch.float16 if f16 else torch.float32
    I, Mk = img.to(U.dev(), dt), mask.to(U.dev(), dt)
    masked = torch.full((B, H, W, 64), 7.0, dtype=torch.float16, device=U.dev())
    mbin = torch.empty((B, H, W), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_prepare_mask(ptr(I), ptr(Mk), 1 if f16 else 0, B, H, W, ptr(masked), 64, ptr(mbin), stream_ptr()) == 0
    torch.cuda.synchronize()
    ref_bin = (mask >= 0.5).float()
    assert torch.equal(mbin.float().cpu(), ref_bin[:, 0])
    assert torch.equal(masked[..., :3].float().cpu(), (img * (ref_bin < 0.5)).permute(0, 2, 3, 1))
    assert float(masked[..., 3:].abs().max()) == 0.0                              # padding channels zeroed


@pytest.mark.parametrize("s", [2, 4, 8])
def test_mask_down_is_nearest_interpolate(lib, s):
    """F.interpolate(mask, size=(H/s, W/s)) (nearest: source pixel floor(i * s)) — prepare_mask_latents :424-427 and mask_features; exact"""
    B, H, W = 2, 64, 48
    m = (torch.rand((B, 1, H, W), generator=torch.Generator().manual_seed(61)) > 0.5).float()
    out = torch.empty((B, H // s, W // s), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_mask_down(ptr(m[:, 0].half().contiguous().to(U.dev())), B, H, W, s, ptr(out), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.float().cpu(), F.interpolate(m, size=(H // s, W // s))[:, 0])


@pytest.mark.parametrize("f16", [False, True])
def test_pose_down8_is_bilinear_interpolate(lib, f16):
    """F.interpolate(pose_map, size=(H/8, W/8), mode='bilinear') (tryon_pipe.py:632-634): fp32 arithmetic, fp16 result"""
    B, C, H, W = 2, 18, 64, 48
    p = torch.rand((B, C, H, W), generator=torch.Generator().manual_seed(62)).half().float()
    Pd = p.to(U.dev(), torch.float16 if f16 else torch.float32)
    out = torch.empty((B, (H // 8) * (W // 8), C), dtype=torch.float16, device=U.dev())
    assert lib.ladi_op_pose_down8(ptr(Pd), 1 if f16 else 0, B, C, H, W, ptr(out), stream_ptr()) == 0
    torch.cuda.synchronize()
    ref = F.interpolate(p, size=(H // 8, W // 8), mode="bilinear").permute(0, 2, 3, 1).reshape(B, -1, C)
    assert (out.float().cpu() - ref).abs().max() <= 1e-3 and U.rel_l2(out.float().cpu(), ref) <= 5e-4


def test_posterior_sample(lib):
    """sf * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) (vae.py:329-348), incl. the clamp on both sides; rel <= 1e-5 (fp32)"""
    from oracle import models as M
    B, h, w = 2, 8, 6
    g = torch.Generator().manual_seed(63)
    mom = torch.randn((B, 8, h, w), generator=g)
    mom[0, 4, 0, 0], mom[0, 5, 0, 0], mom[1, 6, 1, 1] = 40.0, -50.0, 19.5      # clamp at 20 / -30, and an un-clamped large value
    mom = mom.half().float()
    noise = torch.randn((B, 4, h, w), generator=g)
    mm = torch.zeros((B, h * w, 64), dtype=torch.float16)
    mm[..., :8] = mom.permute(0, 2, 3, 1).reshape(B, h * w, 8).half()
    lat = torch.empty((B, h * w, 4), dtype=torch.float32, device=U.dev())
    assert lib.ladi_op_posterior_sample(ptr(mm.to(U.dev())), 64, ptr(noise.to(U.dev())), B, h * w, 0.18215, ptr(lat), stream_ptr()) == 0
    torch.cuda.synchronize()
    ref = (0.18215 * M.posterior_sample(mom, noise)).permute(0, 2, 3, 1).reshape(B, h * w, 4)
    assert U.rel_l2(lat.cpu(), ref) <= 1e-5


@pytest.mark.parametrize("cfg,cloth", [(1, True), (0, True), (1, False)])
def test_assemble_unet_input(lib, cfg, cloth):
    """31-channel order [latents | mask | masked latents | pose | cloth] and CFG batch order [uncond; cond] with zero pose / cloth in the
    uncond half (tryon_pipe.py:702-729); exact up to the fp16 cast of the fp32 latents"""
    B, hw, P_ = 2, 24, 18
    g = torch.Generator().manual_seed(64)
    lat, mlat, clat = [torch.randn((B, hw, 4), generator=g) for _ in range(3)]
    mask = (torch.rand((B, hw), generator=g) > 0.5).half()
    pose = torch.rand((B, hw, P_), generator=g).half()
    n = 2 * B if cfg else B
    out = torch.full((n, hw, 64), 3.0, dtype=torch.float16, device=U.dev())
    d = U.dev()
    keep = [lat.to(d), mask.to(d), mlat.to(d), pose.to(d), clat.to(d)]
    assert lib.ladi_op_assemble_input(ptr(out), 64, B, hw, cfg, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), P_,
                                      ptr(keep[4]) if cloth else None, stream_ptr()) == 0
    torch.cuda.synchronize()
    parts = [lat.half(), mask[..., None], mlat.half(), pose] + ([clat.half()] if cloth else [])
    cond = torch.cat(parts, dim=-1)
    C = cond.shape[-1]
    assert C == (31 if cloth else 27)
    got = out.cpu()
    if cfg:
        unc = cond.clone(); unc[..., 9:] = 0
        assert torch.equal(got[:B, :, :C], unc) and torch.equal(got[B:, :, :C], cond)
    else:
        assert torch.equal(got[..., :C], cond)
    assert float(got[..., C:].abs().max()) == 0.0


@pytest.mark.parametrize("hd,Nq,Nk", [(512, 96, 96), (512, 200, 332), (128, 64, 36), (256, 33, 100)])
def test_flash_attention_wide_head(lib, hd, Nq, Nk):
    """single wide head (the VAE AttentionBlock, d = C = 512): head dim split over the 4 waves of a workgroup, partial scores summed
    through LDS; ragged query / key counts; V consumed transposed.  vs torch softmax(QK^T / sqrt(d)) V in fp32; tolerance 3e-3"""
    n = 2
    q, k, v = _rand((n, Nq, hd), 95), _rand((n, Nk, hd), 96), _rand((n, Nk, hd), 97)
    k[0, 5] *= 3.0                                              # one dominant key: the running maximum jumps mid-sequence
    ref = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), dim=-1) @ v
    Q, K = q.half().to(U.dev()), k.half().to(U.dev())
    VT = v.half().transpose(1, 2).contiguous().to(U.dev())       # [n][hd][Nk]
    O = torch.zeros((n, Nq, hd), dtype=torch.float16, device=U.dev())
    rc = lib.ladi_op_attention_wide(ptr(Q), ptr(K), ptr(VT), ptr(O), hd, hd, Nk, hd, Nq * hd, Nk * hd, hd * Nk, Nq * hd, n, hd, Nq, Nk,
                                    1.0 / math.sqrt(hd), stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert U.rel_l2(O.float().cpu(), ref) < 3e-3, U.rel_l2(O.float().cpu(), ref)


# --------------------------------------------------------------------------------------------------------------- warping glue (§8 f-3)
@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("hw,size", [((512, 384), (256, 192)), ((512, 384), (224, 224)), ((256, 192), (512, 384)), ((37, 53), (16, 20))])
def test_resize_antialias_matches_aten(lib, f16, hw, size):
    """torchvision resize(BILINEAR, antialias=True) == F.interpolate(mode='bilinear', antialias=True) (src/inference.py:242-258,267): the three
    reductions / the grid up-sampling the reference performs, plus a non-integer ragged case; fp32 tolerance 2e-5 abs, fp16 2e-3"""
    import ladi_vton_amd as L
    x = (torch.rand((2, 3) + hw, generator=torch.Generator().manual_seed(70)) * 2 - 1)
    x = x.half().float() if f16 else x
    ref = F.interpolate(x, size=size, mode="bilinear", antialias=True, align_corners=False)
    got = L.resize_antialias(x.half().to(U.dev()) if f16 else x.to(U.dev()), size)
    torch.cuda.synchronize()
    assert got.dtype == (torch.float16 if f16 else torch.float32) and tuple(got.shape) == tuple(ref.shape)
    assert (got.float().cpu() - ref).abs().max() <= (2e-3 if f16 else 2e-5), (got.float().cpu() - ref).abs().max()


def test_grid_sample_border_matches_torch(lib):
    """F.grid_sample(..., padding_mode='border') (src/inference.py:260) incl. samples outside [-1, 1] (clamped to the border) and exactly on
    the last pixel; fp32, abs tolerance 2e-5"""
    import ladi_vton_amd as L
    g = torch.Generator().manual_seed(71)
    x = torch.rand((2, 3, 40, 28), generator=g) * 2 - 1
    grid = torch.rand((2, 64, 48, 2), generator=g) * 2.6 - 1.3
    grid[0, 0, 0] = torch.tensor([1.0, 1.0]); grid[0, 0, 1] = torch.tensor([-1.0, -1.0]); grid[0, 0, 2] = torch.tensor([27.0 / 28.0, 39.0 / 40.0])
    ref = F.grid_sample(x, grid, padding_mode="border", align_corners=False)
    got = L.grid_sample_border(x.to(U.dev()), grid.to(U.dev()))
    torch.cuda.synchronize()
    assert (got.cpu() - ref).abs().max() <= 2e-5, (got.cpu() - ref).abs().max()
    got16 = L.grid_sample_border(x.half().to(U.dev()), grid.to(U.dev()))
    assert got16.dtype == torch.float16 and (got16.float().cpu() - F.grid_sample(x.half().float(), grid, padding_mode="border", align_corners=False)).abs().max() <= 2e-3


@pytest.mark.parametrize("C,Q,hw,act", [(320, 960, (32, 16), "none"), (640, 640, (16, 16), "none"), (320, 2560, (16, 24), "geglu"),
                                        (640, 5120, (8, 16), "geglu")])
def test_linear_with_fused_layernorm(C, Q, hw, act):
    """LayerNorm fused into the X-stationary linear kernel's prologue (BasicTransformerBlock norm1 -> to_q/k/v, norm2 -> attn2.to_q,
    norm3 -> GEGLU; K = 320 / 640): same arithmetic and fp16 rounding point as the stand-alone LayerNorm + linear"""
    x = _rand((1, C) + hw, 110) * 2.0 + 0.3
    gamma, beta = 1.0 + 0.1 * _rand((C,), 111), 0.1 * _rand((C,), 112)
    w, b = _rand((Q, C), 113, 1 / math.sqrt(C)), _rand((Q,), 114, 0.1)
    t = x[0].reshape(C, -1).t()
    xn = F.layer_norm(t, (C,), gamma, beta, 1e-5).half().float()
    if act == "geglu":
        u, g = F.linear(xn, w, b).chunk(2, -1)
        ref = u * F.gelu(g)
        half = Q // 2
        wi, bi = torch.zeros_like(w), torch.zeros_like(b)
        for j in range(half):
            blk, i = divmod(j, 32)
            wi[blk * 64 + i], wi[blk * 64 + 32 + i] = w[j], w[half + j]
            bi[blk * 64 + i], bi[blk * 64 + 32 + i] = b[j], b[half + j]
        y = U.igemm(U.nhwc16(x), wi.half().contiguous().to(U.dev()), Q, ksize=1, bias=bi, act="geglu", ln=(gamma, beta, 1e-5))
        got = y.float().cpu().reshape(-1, half)
        if C == 320:      # the three-workgroup form (cfg 95) with the fused LayerNorm, bit-equal over repeats (counted waits of the two-slot ring)
            y3 = U.igemm(U.nhwc16(x), wi.half().contiguous().to(U.dev()), Q, ksize=1, bias=bi, act="geglu", ln=(gamma, beta, 1e-5), cfg=95)
            assert U.rel_l2(y3.float().cpu().reshape(-1, half), ref) < TOL
            for _ in range(5):
                assert torch.equal(U.igemm(U.nhwc16(x), wi.half().contiguous().to(U.dev()), Q, ksize=1, bias=bi, act="geglu", ln=(gamma, beta, 1e-5), cfg=95), y3)
    else:
        ref = F.linear(xn, w, b)
        y = U.igemm(U.nhwc16(x), w.half().contiguous().to(U.dev()), Q, ksize=1, bias=b, ln=(gamma, beta, 1e-5))
        got = y.float().cpu().reshape(-1, Q)
    assert U.rel_l2(got, ref) < TOL, U.rel_l2(got, ref)


@pytest.mark.parametrize("C,Q,cfg", [(128, 128, 0), (320, 960, 7), (320, 960, 0), (1280, 1280, 0)])
def test_layernorm_through_the_scratch(C, Q, cfg):
    """shapes / configurations the X-stationary kernel does not take: the launcher runs the stand-alone LayerNorm kernel into the
    caller's scratch and feeds the GEMM from there (cfg 0: the tuner times the fused and the two-kernel form and keeps the faster)"""
    x, w, b = _rand((1, C, 16, 16), 115) + 0.2, _rand((Q, C), 116, 1 / math.sqrt(C)), _rand((Q,), 117, 0.1)
    gamma, beta = 1.0 + 0.1 * _rand((C,), 118), 0.1 * _rand((C,), 119)
    ref = F.linear(F.layer_norm(x[0].reshape(C, -1).t(), (C,), gamma, beta, 1e-5).half().float(), w, b)
    y = U.igemm(U.nhwc16(x), w.half().contiguous().to(U.dev()), Q, ksize=1, bias=b, cfg=cfg, ln=(gamma, beta, 1e-5, True))
    assert U.rel_l2(y.float().cpu().reshape(-1, Q), ref) < TOL


def test_layernorm_without_scratch_needs_a_fusable_shape():
    """no scratch: only the fused form is admissible, anything else is an error (never a silently skipped LayerNorm)"""
    x, w = _rand((1, 128, 16, 16), 115), _rand((128, 128), 116, 0.1)
    with pytest.raises(AssertionError):
        U.igemm(U.nhwc16(x), w.half().contiguous().to(U.dev()), 128, ksize=1, ln=(torch.ones(128), torch.zeros(128), 1e-5))


@pytest.mark.parametrize("cin,hw,cfg", [(320, (64, 48), 0), (320, (64, 48), 23), (320, (64, 48), 25), (640, (32, 24), 0), (640, (32, 24), 26), (320, (8, 16), 27)])
def test_linear_with_fused_group_norm_affine(cin, hw, cfg):
    """GroupNorm (no activation) in front of proj_in folded into the X-stationary kernel's register panel: out = W (x * scale[n][c] +
    shift[n][c]) + b with the affine rounded to fp16 like gn_apply_kernel; every configuration that is not the X-stationary kernel must
    refuse the launch (no silent un-normalised product)"""
    N = 2
    h, w = hw
    x = _rand((N, cin, h, w), 300)
    wt, b = _rand((cin, cin, 1, 1), 301, 1 / math.sqrt(cin)), _rand((cin,), 302, 0.1)
    g = torch.Generator().manual_seed(303)
    scale = 0.5 + torch.rand((N, cin), generator=g)
    shift = torch.randn((N, cin), generator=g) * 0.3
    ss = torch.stack([scale, shift], dim=-1).contiguous().to(U.dev())
    xn = (x.half().float() * scale[:, :, None, None] + shift[:, :, None, None]).half().float()
    ref = F.conv2d(xn, wt.half().float(), b.half().float())
    y = U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cin, ksize=1, bias=b, cfg=cfg, gn=(ss, h * w))
    assert U.rel_l2(U.to_nchw(y), ref) < TOL, (cin, hw, cfg)
    with pytest.raises(AssertionError):
        U.igemm(U.nhwc16(x), U.pack_conv_weight(wt), cin, ksize=1, bias=b, cfg=7, gn=(ss, h * w))
