// GroupNorm (statistics + apply), LayerNorm and row-softmax kernels. All HBM-bound: 16-byte vector
// accesses, fp32 statistics.
#include "common.h"
#include "kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
// GroupNorm = per-channel partial statistics  ->  finalize (scale/shift per (n, c))  ->  apply.
// Partial statistics are rows of [C][2] floats (sum, sum of squares) over a fixed number of pixels; they are normally written
// by the PRODUCING igemm's epilogue (igemm.hip) and only by gn_partial_kernel when the producer could not (unaligned tiles).
// No atomics anywhere: bitwise reproducible.
// ------------------------------------------------------------------------------------------------
constexpr int GN_MAX_C = 2560;

// fallback producer: block (b, n) reduces pixels [b*ppb, (b+1)*ppb) of sample n for all channels -> part[(n*nb + b)][C][2]
__global__ __launch_bounds__(256) void gn_partial_kernel(const h16* __restrict__ src, int C, int ld, int HW, int ppb,
                                                         float* __restrict__ part) {
    __shared__ float red[2 * (GN_MAX_C > 2048 ? GN_MAX_C : 2048)];
    const int tid = threadIdx.x;
    const int n = blockIdx.y, nb = gridDim.x;
    const int octs = C >> 3;
    const int to = octs < 256 ? octs : 256;   // octets handled concurrently
    const int pl = 256 / to;                  // pixel lanes
    const int my_o = tid % to, my_p = tid / to;
    const int pix0 = blockIdx.x * ppb;
    const int npix = min(ppb, HW - pix0);
    const size_t row0 = (size_t)n * HW + pix0;
    float* rsum = red;
    float* rsq = red + (GN_MAX_C > 2048 ? GN_MAX_C : 2048);
    for (int oc0 = 0; oc0 < octs; oc0 += to) {
        const int oc = oc0 + my_o;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        if (my_p < pl && oc < octs) {
            const h16* base = src + (oc << 3);
            int p = my_p;
            for (; p + pl < npix; p += 2 * pl) {   // two independent loads in flight
                const h16x8 v0 = *reinterpret_cast<const h16x8*>(base + (row0 + p) * ld);
                const h16x8 v1 = *reinterpret_cast<const h16x8*>(base + (row0 + p + pl) * ld);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float x = (float)v0[e], y = (float)v1[e]; s[e] += x + y; q[e] += x * x + y * y; }
            }
            for (; p < npix; p += pl) {
                const h16x8 v0 = *reinterpret_cast<const h16x8*>(base + (row0 + p) * ld);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float x = (float)v0[e]; s[e] += x; q[e] += x * x; }
            }
        }
        // combine the pixel lanes through LDS: slot [my_p][channel within this octet chunk]
        __syncthreads();
        if (my_p < pl && oc < octs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { rsum[(my_p * to + my_o) * 8 + e] = s[e]; rsq[(my_p * to + my_o) * 8 + e] = q[e]; }
        }
        __syncthreads();
        const int nch = min(to, octs - oc0) * 8;
        for (int cc = tid; cc < nch; cc += 256) {
            float ts = 0.f, tq = 0.f;
            for (int l = 0; l < pl; ++l) { ts += rsum[l * to * 8 + cc]; tq += rsq[l * to * 8 + cc]; }
            float* o = part + (((size_t)n * nb + blockIdx.x) * C + (oc0 << 3) + cc) * 2;
            o[0] = ts; o[1] = tq;
        }
    }
}

// block (n, group chunk): scale_shift[n][c] = (gamma*rstd, beta - mean*gamma*rstd) for GroupNorm over the virtual concat
// (C0 | C1).  Each block owns GPB consecutive groups (nch = GPB*gs channels); threads = (row lane, channel) so that the
// partial rows are summed in parallel with coalesced 8-byte loads.
constexpr int GNF_MAX_CH = 512;   // channels per block (GPB * gs)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part0, int C0, int rps0,
                                                          const float* __restrict__ part1, int C1, int rps1, int HW, int groups,
                                                          int gpb, const h16* __restrict__ gamma, const h16* __restrict__ beta,
                                                          float eps, float* __restrict__ scale_shift, int* __restrict__ bad) {
    __shared__ float rsum[256], rsq[256];
    __shared__ float csum[GNF_MAX_CH], csq[GNF_MAX_CH];
    __shared__ float gmean[16], grstd[16];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int Ct = C0 + C1;
    const int gs = Ct / groups;
    const int g0 = blockIdx.y * gpb;
    const int ng = min(gpb, groups - g0);
    const int cbeg = g0 * gs, nch = ng * gs;
    const int nchp = nch < 256 ? nch : 256;     // channels handled concurrently
    const int RL = 256 / nchp;                  // row lanes
    const int cl = tid % nchp, rl = tid / nchp;
    for (int cb = 0; cb < nch; cb += nchp) {
        const int cc = cb + cl;                 // channel within this block's range
        float s = 0.f, q = 0.f;
        if (rl < RL && cc < nch) {
            const int c = cbeg + cc;
            const float* p; int C, rps, clc;
            if (c < C0) { p = part0; C = C0; rps = rps0; clc = c; } else { p = part1; C = C1; rps = rps1; clc = c - C0; }
            const float2* row = reinterpret_cast<const float2*>(p) + (size_t)n * rps * C + clc;
            for (int r = rl; r < rps; r += RL) { const float2 v = row[(size_t)r * C]; s += v.x; q += v.y; }
        }
        __syncthreads();
        if (rl < RL) { rsum[rl * nchp + cl] = s; rsq[rl * nchp + cl] = q; }
        __syncthreads();
        if (rl == 0 && cc < nch) {
            float ts = 0.f, tq = 0.f;
            for (int l = 0; l < RL; ++l) { ts += rsum[l * nchp + cl]; tq += rsq[l * nchp + cl]; }
            csum[cc] = ts; csq[cc] = tq;
        }
    }
    __syncthreads();
    if (tid < ng) {
        float s = 0.f, q = 0.f;
        for (int c = tid * gs; c < (tid + 1) * gs; ++c) { s += csum[c]; q += csq[c]; }
        const float inv = 1.f / ((float)gs * (float)HW);
        const float mean = s * inv;
        const float var = fmaxf(q * inv - mean * mean, 0.f);
        gmean[tid] = mean; grstd[tid] = rsqrtf(var + eps);
        // fp16-range guard: an activation that overflowed to inf (or a NaN behind it) shows up in the statistics of the next GroupNorm
        if (bad && !(fabsf(s) <= 3.0e38f && q <= 3.0e38f)) *bad = 1;
    }
    __syncthreads();
    for (int cc = tid; cc < nch; cc += 256) {
        const int c = cbeg + cc, g = cc / gs;
        const float ga = (float)gamma[c] * grstd[g];
        reinterpret_cast<float2*>(scale_shift)[(size_t)n * Ct + c] = make_float2(ga, (float)beta[c] - gmean[g] * ga);
    }
}

// y = act(x * scale + shift) (+ add): pure streaming, no LDS, no block-level setup
__global__ __launch_bounds__(256) void gn_apply_kernel(const h16* __restrict__ src0, int C0, int ld0,
                                                       const h16* __restrict__ src1, int C1, int ld1, int HW,
                                                       const float* __restrict__ scale_shift, int silu,
                                                       const h16* __restrict__ add, h16* __restrict__ out, int pix_per_block) {
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int Ct = C0 + C1;
    const int octs = Ct >> 3;
    const int to = octs < 256 ? octs : 256;
    const int pl = 256 / to;
    const int my_o = tid % to, my_p = tid / to;
    if (my_p >= pl) return;
    const int pix0 = blockIdx.x * pix_per_block;
    const int npix = min(pix_per_block, HW - pix0);
    const size_t row0 = (size_t)n * HW + pix0;
    for (int oc = my_o; oc < octs; oc += to) {
        const int c = oc << 3;
        const h16* base; int ld;
        if (c < C0) { base = src0 + c; ld = ld0; } else { base = src1 + (c - C0); ld = ld1; }
        float sc[8], sh[8];
        {
            const float4* ss = reinterpret_cast<const float4*>(scale_shift + ((size_t)n * Ct + c) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float4 t = ss[e]; sc[2 * e] = t.x; sh[2 * e] = t.y; sc[2 * e + 1] = t.z; sh[2 * e + 1] = t.w; }
        }
        auto xform = [&](const h16x8& v, const size_t row) {
            h16x8 o;
            h16x8 ad;
            if (add) ad = *reinterpret_cast<const h16x8*>(add + row * Ct + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = (float)v[e] * sc[e] + sh[e];
                if (silu) y = silu_fast(y);
                if (add) y += (float)ad[e];
                o[e] = (h16)y;
            }
            *reinterpret_cast<h16x8*>(out + row * Ct + c) = o;
        };
        int p = my_p;
        for (; p + 3 * pl < npix; p += 4 * pl) {   // four independent 16-byte loads in flight
            const h16x8 v0 = *reinterpret_cast<const h16x8*>(base + (row0 + p) * ld);
            const h16x8 v1 = *reinterpret_cast<const h16x8*>(base + (row0 + p + pl) * ld);
            const h16x8 v2 = *reinterpret_cast<const h16x8*>(base + (row0 + p + 2 * pl) * ld);
            const h16x8 v3 = *reinterpret_cast<const h16x8*>(base + (row0 + p + 3 * pl) * ld);
            xform(v0, row0 + p); xform(v1, row0 + p + pl); xform(v2, row0 + p + 2 * pl); xform(v3, row0 + p + 3 * pl);
        }
        for (; p < npix; p += pl) {
            const h16x8 v0 = *reinterpret_cast<const h16x8*>(base + (row0 + p) * ld);
            xform(v0, row0 + p);
        }
    }
}

// Round 6: partial rows of the VAE-sized tensors.  At 512x384 a producer's epilogue writes one row per 32 or 64 pixels -- 3 072-6 144 rows of
// [C][2] floats per sample, 50 MB for C = 128 at batch 8 -- and gn_finalize walked them with 8 + (n, 4 groups) blocks of strided 8-byte loads:
// 52.6 us x 62 launches per step, pure tail (profiles/r05_vae_kernel_stats.txt).  This kernel folds them, fully coalesced (a row is C * 8
// contiguous bytes), into R2 rows per sample -- block (b, n) sums rows [b rps / R2, (b + 1) rps / R2) in order: fixed association, bitwise
// reproducible -- after which the ONE-pass kernel below takes the tensor like any UNet-sized one (R2 <= GNX_MAX_RPS).
__global__ __launch_bounds__(256) void gn_reduce_rows_kernel(const float* __restrict__ part, int C, int rps, int r2, float* __restrict__ out) {
    __shared__ float4 red[256];
    const int tid = threadIdx.x, b = blockIdx.x, n = blockIdx.y;
    const int q = C >> 1;                               // float4 pieces per row (two channels each); launcher: q <= 256 or a multiple of 256
    const int lanes = q < 256 ? 256 / q : 1;            // row lanes
    const int r_lo = (int)((long long)b * rps / r2), r_hi = (int)((long long)(b + 1) * rps / r2);
    for (int q0 = 0; q0 < q; q0 += 256) {
        const int qi = q0 + (q < 256 ? tid % q : tid), rl = q < 256 ? tid / q : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < lanes && qi < q) {
            const float4* row = reinterpret_cast<const float4*>(part) + ((size_t)n * rps) * q + qi;
            int r = r_lo + rl;
            for (; r + 3 * lanes < r_hi; r += 4 * lanes) {          // four independent 16-byte loads in flight
                const float4 a0 = row[(size_t)r * q], a1 = row[(size_t)(r + lanes) * q], a2 = row[(size_t)(r + 2 * lanes) * q], a3 = row[(size_t)(r + 3 * lanes) * q];
                acc.x += (a0.x + a1.x) + (a2.x + a3.x); acc.y += (a0.y + a1.y) + (a2.y + a3.y);
                acc.z += (a0.z + a1.z) + (a2.z + a3.z); acc.w += (a0.w + a1.w) + (a2.w + a3.w);
            }
            for (; r < r_hi; r += lanes) { const float4 a0 = row[(size_t)r * q]; acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w; }
        }
        if (lanes > 1) {
            __syncthreads();
            red[tid] = acc;
            __syncthreads();
            if (tid < q) {
                float4 t = red[tid];
                for (int l = 1; l < lanes; ++l) { const float4 u = red[l * q + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                reinterpret_cast<float4*>(out)[((size_t)n * r2 + b) * q + tid] = t;
            }
        } else if (rl < lanes && qi < q) {              // one row lane: threads beyond the row's q pieces (tid >= q) hold nothing
            reinterpret_cast<float4*>(out)[((size_t)n * r2 + b) * q + qi] = acc;
        }
    }
}

// GroupNorm in ONE pass over the data (round 5): finalize folded into apply.  Block = (pixel block, sample, 64-channel chunk): the block sums the
// partial rows of the groups that overlap its chunk itself (at most 64 + 2 gs - 2 channels x rps rows of 8 bytes, L2-resident: the producer's
// epilogue wrote them a moment ago), derives scale / shift for its 64 channels and streams its 128-byte row pieces.  Removes the gn_finalize
// launch (6.8 us of launch latency and dependent loads for 1.8 MB of traffic, 61 per UNet forward) wherever the rows per sample are few; the
// summation order is fixed (row lanes, then channels of a group), so the result is bitwise reproducible.  Same rounding point as
// gn_finalize + gn_apply: statistics fp32, y = fp16(act(x * scale + shift) (+ add)).
constexpr int GNX_CH = 64;          // channels per block
constexpr int GNX_MAX_GS = 96;      // largest group size handled (channels per group); stat channels <= 64 + 2 * 96 - 2 <= 256
constexpr int GNX_DIRECT_HW = 64;   // samples of at most this many pixels: a source WITHOUT partial rows (rps = 0) has its statistics taken from the data
constexpr int GNX_MAX_RPS = 96;     // more partial rows per sample than this: the three-stage form (VAE-sized tensors).  96 = the 64x48 level behind
                                    // the 12-wave 320x192 halo tile, whose epilogue writes one row per 32 pixels (64 KB of L2-resident rows per block)
__global__ __launch_bounds__(256) void gn_norm_kernel(const h16* __restrict__ src0, int C0, int ld0, const float* __restrict__ part0, int rps0,
                                                      const h16* __restrict__ src1, int C1, int ld1, const float* __restrict__ part1, int rps1,
                                                      int HW, int gs, const h16* __restrict__ gamma, const h16* __restrict__ beta, float eps,
                                                      int silu, const h16* __restrict__ add, h16* __restrict__ out, int ppb, int* __restrict__ bad, int early) {
    __shared__ float rsum[256], rsq[256];
    __shared__ float csum[256], csq[256];
    __shared__ float gmean[GNX_CH + 2], grstd[GNX_CH + 2];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int Ct = C0 + C1;
    const int c0 = blockIdx.z * GNX_CH;
    const int nch = min(GNX_CH, Ct - c0);
    const int g_lo = c0 / gs, g_hi = (c0 + nch - 1) / gs, ng = g_hi - g_lo + 1;
    const int cb = g_lo * gs, ncs = ng * gs;            // statistics channels [cb, cb + ncs): whole groups, <= 256
    // streaming side: octet of the chunk, pixel lane (32 of them).  The FIRST round of data loads (up to four 16-byte pieces per thread) is
    // issued before the statistics phase: it needs nothing from it, and its round trip then overlaps the one of the partial rows instead
    // of following it (the statistics phase is ~2 us of every block's life, exposed once per launch because all blocks start together)
    const int my_o = tid & 7, my_p = tid >> 3;
    const int c = c0 + my_o * 8;
    const bool live = c < Ct;
    const h16* base = src0; int ld = ld0;
    if (live) { if (c < C0) { base = src0 + c; ld = ld0; } else { base = src1 + (c - C0); ld = ld1; } }
    const int pix0 = blockIdx.x * ppb;
    const int npix = min(ppb, HW - pix0);
    const size_t row0 = (size_t)n * HW + pix0;
    const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    h16x8 ev[4] = {zero, zero, zero, zero}, ea[4] = {zero, zero, zero, zero};
    if (live && early) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = my_p + 32 * k;
            if (p < npix) {
                ev[k] = *reinterpret_cast<const h16x8*>(base + (row0 + p) * ld);
                if (add) ea[k] = *reinterpret_cast<const h16x8*>(add + (row0 + p) * Ct + c);
            }
        }
    }
    {
        const int RL = 256 / ncs;                       // row lanes
        const int cl = tid % ncs, rl = tid / ncs;
        float s = 0.f, q = 0.f;
        if (rl < RL) {
            const int c = cb + cl;
            const float* p; int C, rps, clc;
            if (c < C0) { p = part0; C = C0; rps = rps0; clc = c; } else { p = part1; C = C1; rps = rps1; clc = c - C0; }
            if (p) {
                const float2* row = reinterpret_cast<const float2*>(p) + (size_t)n * rps * C + clc;
                for (int r = rl; r < rps; r += RL) { const float2 v = row[(size_t)r * C]; s += v.x; q += v.y; }
            } else {
                // no partial rows for this source (the producer could not write them: 48-pixel samples are not whole 32-pixel row blocks) and
                // the sample is tiny: the statistics come straight from the data -- no gn_partial launch (the 8x6 level of the UNet)
                const h16* sp = (c < C0) ? src0 + clc : src1 + clc;
                const int ldc = (c < C0) ? ld0 : ld1;
                for (int r = rl; r < HW; r += RL) { const float x = (float)sp[((size_t)n * HW + r) * ldc]; s += x; q += x * x; }
            }
            rsum[rl * ncs + cl] = s; rsq[rl * ncs + cl] = q;
        }
        __syncthreads();
        if (tid < ncs) {
            float ts = 0.f, tq = 0.f;
            for (int l = 0; l < RL; ++l) { ts += rsum[l * ncs + tid]; tq += rsq[l * ncs + tid]; }
            csum[tid] = ts; csq[tid] = tq;
        }
        __syncthreads();
        if (tid < ng) {
            float ts = 0.f, tq = 0.f;
            for (int c = tid * gs; c < (tid + 1) * gs; ++c) { ts += csum[c]; tq += csq[c]; }
            const float inv = 1.f / ((float)gs * (float)HW);
            const float mean = ts * inv;
            const float var = fmaxf(tq * inv - mean * mean, 0.f);
            gmean[tid] = mean; grstd[tid] = rsqrtf(var + eps);
            if (bad && !(fabsf(ts) <= 3.0e38f && tq <= 3.0e38f)) *bad = 1;     // fp16-range guard (see gn_finalize_kernel)
        }
        __syncthreads();
    }
    if (!live) return;
    float sc[8], sh[8];
    {
        const h16x8 ga = *reinterpret_cast<const h16x8*>(gamma + c), be = *reinterpret_cast<const h16x8*>(beta + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c + e) / gs - g_lo;
            sc[e] = (float)ga[e] * grstd[g];
            sh[e] = (float)be[e] - gmean[g] * sc[e];
        }
    }
    auto xform = [&](const h16x8& v, const h16x8& ad, const size_t row) {
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[e] * sc[e] + sh[e];
            if (silu) y = silu_fast(y);
            if (add) y += (float)ad[e];
            o[e] = (h16)y;
        }
        *reinterpret_cast<h16x8*>(out + row * Ct + c) = o;
    };
    int p = my_p;
    if (early) {                                // the round loaded ahead of the statistics (early = 0: A/B switch LADI_GN_EARLY=0)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (my_p + 32 * k < npix) xform(ev[k], ea[k], row0 + my_p + 32 * k);
        p = my_p + 128;
    }
    for (; p + 96 < npix; p += 128) {           // four independent 16-byte loads in flight (all loads of a round before its first store)
        const h16x8 v0 = *reinterpret_cast<const h16x8*>(base + (row0 + p) * ld);
        const h16x8 v1 = *reinterpret_cast<const h16x8*>(base + (row0 + p + 32) * ld);
        const h16x8 v2 = *reinterpret_cast<const h16x8*>(base + (row0 + p + 64) * ld);
        const h16x8 v3 = *reinterpret_cast<const h16x8*>(base + (row0 + p + 96) * ld);
        h16x8 a0 = zero, a1 = zero, a2 = zero, a3 = zero;
        if (add) {
            a0 = *reinterpret_cast<const h16x8*>(add + (row0 + p) * Ct + c);
            a1 = *reinterpret_cast<const h16x8*>(add + (row0 + p + 32) * Ct + c);
            a2 = *reinterpret_cast<const h16x8*>(add + (row0 + p + 64) * Ct + c);
            a3 = *reinterpret_cast<const h16x8*>(add + (row0 + p + 96) * Ct + c);
        }
        xform(v0, a0, row0 + p); xform(v1, a1, row0 + p + 32); xform(v2, a2, row0 + p + 64); xform(v3, a3, row0 + p + 96);
    }
    for (; p < npix; p += 32) {
        const h16x8 v0 = *reinterpret_cast<const h16x8*>(base + (row0 + p) * ld);
        h16x8 a0 = zero;
        if (add) a0 = *reinterpret_cast<const h16x8*>(add + (row0 + p) * Ct + c);
        xform(v0, a0, row0 + p);
    }
}

// LayerNorm: each wave normalises R consecutive rows with all of their 16-byte loads in flight at once (rows are only 640 B - 2.5 KB,
// so one row per wave leaves the memory system idle); statistics two-pass in registers. C % 8 == 0, C <= 64*8*MAXO.
template <int R, int MAXO>
__global__ __launch_bounds__(256) void layernorm_kernel(const h16* __restrict__ x, int ldx, const h16* __restrict__ gamma,
                                                        const h16* __restrict__ beta, float eps, int rows, int C,
                                                        h16* __restrict__ out, int ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    if (row0 >= rows) return;
    const int octs = C >> 3;
    h16x8 v[R][MAXO];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
        const bool rv = row0 + r < rows;
#pragma unroll
        for (int i = 0; i < MAXO; ++i) {
            const int oc = lane + 64 * i;
            h16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (rv && oc < octs) t = *reinterpret_cast<const h16x8*>(x + (size_t)(row0 + r) * ldx + oc * 8);
            v[r][i] = t;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i = 0; i < MAXO; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[r] += (float)v[r][i][e];
        s[r] = wave_sum(s[r]);
    }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = s[r] / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXO; ++i) {
            const int oc = lane + 64 * i;
            if (oc < octs) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)v[r][i][e] - mean[r]; ss += d * d; }
            }
        }
        ss = wave_sum(ss);
        rstd[r] = rsqrtf(ss / (float)C + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int oc = lane + 64 * i;
        if (oc < octs) {
            const h16x8 g = *reinterpret_cast<const h16x8*>(gamma + oc * 8);
            const h16x8 b = *reinterpret_cast<const h16x8*>(beta + oc * 8);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < rows) {
                    h16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)v[r][i][e] - mean[r]) * rstd[r] * (float)g[e] + (float)b[e]);
                    *reinterpret_cast<h16x8*>(out + (size_t)(row0 + r) * ldo + oc * 8) = o;
                }
            }
        }
    }
}

// one block per row: P = softmax(scale * S) ; cols % 4 == 0
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int cols, float scale,
                                                           h16* __restrict__ P) {
    __shared__ float red[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* s = S + (size_t)blockIdx.x * cols;
    h16* p = P + (size_t)blockIdx.x * cols;
    const int q4 = cols >> 2;
    float m = -3.0e38f;
    for (int i = tid; i < q4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(s)[i];
        m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float sc = scale * 1.4426950408889634f;
    const float msc = m * sc;
    float sum = 0.f;
    for (int i = tid; i < q4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(s)[i];
        sum += exp2f(v.x * sc - msc) + exp2f(v.y * sc - msc) + exp2f(v.z * sc - msc) + exp2f(v.w * sc - msc);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.f / sum;
    for (int i = tid; i < q4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(s)[i];
        h16x4 o;
        o[0] = (h16)(exp2f(v.x * sc - msc) * inv); o[1] = (h16)(exp2f(v.y * sc - msc) * inv);
        o[2] = (h16)(exp2f(v.z * sc - msc) * inv); o[3] = (h16)(exp2f(v.w * sc - msc) * inv);
        reinterpret_cast<h16x4*>(p)[i] = o;
    }
}

// pixels per block so that the grid has >= ~4 blocks per CU without making blocks tiny
inline int pick_ppb(int n, int HW, int octs) {
    int ppb = 4096 * 8 / (octs > 0 ? octs : 1);   // ~32K octets (512 KB) per block
    if (ppb < 32) ppb = 32;
    while (ppb > 48 && (long long)n * ((HW + ppb - 1) / ppb) < 768) ppb >>= 1;
    return ppb;
}

}  // namespace

int ladi_gn_partial_rows(int n, int HW, int C) { return (HW + pick_ppb(n, HW, C >> 3) - 1) / pick_ppb(n, HW, C >> 3); }

int ladi_launch_gn_partial(const h16* src, int C, int ld, int n, int HW, float* part, hipStream_t st) {
    if ((C & 7) || (ld & 7) || C > GN_MAX_C) return -1;
    const int ppb = pick_ppb(n, HW, C >> 3);
    dim3 grid((HW + ppb - 1) / ppb, n);
    hipLaunchKernelGGL(gn_partial_kernel, grid, dim3(256), 0, st, src, C, ld, HW, ppb, part);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_gn_finalize(const float* part0, int C0, int rps0, const float* part1, int C1, int rps1, int n, int HW, int groups,
                            const h16* gamma, const h16* beta, float eps, float* scale_shift, hipStream_t st, int* bad) {
    const int Ct = C0 + C1;
    if (groups > 64 || (Ct % groups) || Ct > GN_MAX_C) return -1;
    const int gs = Ct / groups;
    int gpb = 4;                                   // groups per block: 8 blocks per sample for 32 groups
    while (gpb > 1 && gpb * gs > GNF_MAX_CH) gpb >>= 1;
    if (gpb * gs > GNF_MAX_CH || gpb > 16) return -1;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(n, (groups + gpb - 1) / gpb), dim3(256), 0, st, part0, C0, rps0, part1, C1, rps1, HW,
                       groups, gpb, gamma, beta, eps, scale_shift, bad);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_gn_apply(const h16* src0, int C0, int ld0, const h16* src1, int C1, int ld1, int n, int HW,
                         const float* scale_shift, int silu, const h16* add, h16* out, hipStream_t st) {
    const int Ct = C0 + C1;
    if ((C0 & 7) || (C1 & 7) || (ld0 & 7) || (C1 && (ld1 & 7))) return -1;
    const int ppb = pick_ppb(n, HW, Ct >> 3);
    dim3 grid((HW + ppb - 1) / ppb, n);
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), 0, st, src0, C0, ld0, src1, C1, ld1, HW, scale_shift, silu, add, out, ppb);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

// rows per sample the fold leaves (ladi_launch_gn_reduce): few enough for the one-pass kernel's per-block re-summation to be noise beside its data
int ladi_gn_reduce_rows() { return 16; }
bool ladi_gn_reduce_eligible(int C, int rps) {
    const char* e = getenv("LADI_GN_REDUCE");        // =0: the three-stage form as in round 5 (A/B)
    return !(e && e[0] == '0') && !(C & 1) && rps > GNX_MAX_RPS;
}
int ladi_launch_gn_reduce(const float* part, int C, int rps, int n, float* out, hipStream_t st) {
    if (!ladi_gn_reduce_eligible(C, rps) || (reinterpret_cast<uintptr_t>(part) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return -1;
    hipLaunchKernelGGL(gn_reduce_rows_kernel, dim3(ladi_gn_reduce_rows(), n), dim3(256), 0, st, part, C, rps, ladi_gn_reduce_rows(), out);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

bool ladi_gn_norm_direct(int HW) {
    const char* e = getenv("LADI_GN_DIRECT");        // =0: always gn_partial (A/B)
    return HW <= GNX_DIRECT_HW && !(e && e[0] == '0');
}

// rps = 0 for a source means "no partial rows: take the statistics from the data" (only for samples of <= GNX_DIRECT_HW pixels)
bool ladi_gn_norm_eligible(int C0, int rps0, int C1, int rps1, int groups, int HW) {
    const char* e = getenv("LADI_GN_ONEPASS");       // read per call (host side, at graph-capture time): one process can A/B both forms
    const bool on = !(e && e[0] == '0');
    const int Ct = C0 + C1;
    if (!on || groups <= 0 || (Ct % groups)) return false;
    const int gs = Ct / groups;
    const char* me = getenv("LADI_GN_MAX_RPS");      // sweeps
    const int max_rps = me ? atoi(me) : GNX_MAX_RPS;
    const int lo = ladi_gn_norm_direct(HW) ? 0 : 1;
    return gs <= GNX_MAX_GS && rps0 >= lo && rps0 <= max_rps && (C1 == 0 || (rps1 >= lo && rps1 <= max_rps)) && groups * gs == Ct;
}

int ladi_launch_gn_norm(const h16* src0, int C0, int ld0, const float* part0, int rps0, const h16* src1, int C1, int ld1, const float* part1,
                        int rps1, int n, int HW, int groups, const h16* gamma, const h16* beta, float eps, int silu, const h16* add, h16* out,
                        hipStream_t st, int* bad) {
    const int Ct = C0 + C1;
    if ((C0 & 7) || (C1 & 7) || (ld0 & 7) || (C1 && (ld1 & 7)) || !ladi_gn_norm_eligible(C0, rps0, C1, rps1, groups, HW)) return -1;
    if ((rps0 == 0 && part0) || (rps0 > 0 && !part0) || (C1 && ((rps1 == 0 && part1) || (rps1 > 0 && !part1)))) return -1;
    const int chunks = (Ct + GNX_CH - 1) / GNX_CH;
    // pixels per block: 128-byte row pieces x ppb rows; halve until the grid has ~3 blocks per CU (LADI_GN_PPB pins it: sweeps)
    const char* pe = getenv("LADI_GN_PPB");
    const int pin = pe ? atoi(pe) : 0;
    // (round 6: VAE-sized tensors reach this kernel behind ladi_launch_gn_reduce -- larger pixel blocks there, so that a block's re-summation
    // of the partial rows, <= 16 rows x ~70 channels, stays ~1 % of the bytes it streams)
    int ppb = HW >= 32768 ? 2048 : 512;
    if (pin >= 32) ppb = pin;
    else while (ppb > 64 && (long long)n * chunks * ((HW + ppb - 1) / ppb) < 768) ppb >>= 1;
    dim3 grid((HW + ppb - 1) / ppb, n, chunks);
    const char* ee = getenv("LADI_GN_EARLY");
    hipLaunchKernelGGL(gn_norm_kernel, grid, dim3(256), 0, st, src0, C0, ld0, part0, rps0, src1, C1, ld1, part1, rps1, HW, Ct / groups, gamma, beta,
                       eps, silu, add, out, ppb, bad, (ee && ee[0] == '0') ? 0 : 1);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_layernorm(const h16* x, int ldx, const h16* gamma, const h16* beta, float eps, int rows, int C, h16* out,
                          int ldo, hipStream_t st) {
    if ((C & 7) || C > 4096 || (ldx & 7) || (ldo & 7)) return -1;
    const int octs = C >> 3;
    // few rows (the 16x12 / 8x6 levels: 3 072 / 768 tokens at batch 8): one row per wave -- four rows per wave would leave most SIMDs without a wave and
    // the launch is latency, not bandwidth (7.8 us for 7.9 MB in round 5); LADI_LN_R4=1 restores the four-row form everywhere (A/B)
    static const bool r4_always = [] { const char* e = getenv("LADI_LN_R4"); return e && e[0] == '1'; }();
    const bool few = rows <= 4096 && !r4_always;
    if (octs <= 64 && few) hipLaunchKernelGGL((layernorm_kernel<1, 1>), dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, gamma, beta, eps, rows, C, out, ldo);
    else if (octs <= 64) hipLaunchKernelGGL((layernorm_kernel<4, 1>), dim3((rows + 15) / 16), dim3(256), 0, st, x, ldx, gamma, beta, eps, rows, C, out, ldo);
    else if (octs <= 192 && few) hipLaunchKernelGGL((layernorm_kernel<1, 3>), dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, gamma, beta, eps, rows, C, out, ldo);
    else if (octs <= 192) hipLaunchKernelGGL((layernorm_kernel<4, 3>), dim3((rows + 15) / 16), dim3(256), 0, st, x, ldx, gamma, beta, eps, rows, C, out, ldo);
    else hipLaunchKernelGGL((layernorm_kernel<1, 8>), dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, gamma, beta, eps, rows, C, out, ldo);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_softmax_rows(const float* S, int rows, int cols, float scale, h16* P, hipStream_t st) {
    if (cols & 3) return -1;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, S, cols, scale, P);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}
