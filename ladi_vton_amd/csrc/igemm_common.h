// Device-side pieces shared by the implicit-GEMM kernels (igemm.hip, igemm8.hip): LDS swizzle and the fused epilogue.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BK>
__device__ __forceinline__ int swz(int row, int chunk) {
    if constexpr (BK == 64) return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
    else return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3);
}

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the kernels of this file.  On entry every wave holds acc[TQ][TP] (32x32 MFMA accumulators) of its
// (TQ*32 channels) x (TP*32 pixels) sub-tile at channel offset q0 + wq*TQ*32, pixel offset p0 + wp*TP*32; the staging ring is dead
// (the caller has drained every LDS-DMA).
// ------------------------------------------------------------------------------------------------
template <int WQ, int WP, int TQ, int TP>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& a, f32x16 (&acc)[TQ][TP], h16* smem, const int q0, const int p0,
                                               const int pt, const int z, const int wave, const int lane) {
    const int wq = wave / WP, wp = wave % WP;
    const int l31 = lane & 31, hh = lane >> 5;
    // ------------------------------------------------------------------------------------------
    // Epilogue: lane owns pixel (col) l31 of each pixel sub-tile and 4-channel groups of each q sub-tile.
    // ------------------------------------------------------------------------------------------
    const bool geglu = (a.act == LADI_ACT_GEGLU);
    const int Qout = geglu ? a.Q / 2 : a.Q;
    const float* rowadd = a.rowadd;
    if (rowadd && a.rowadd_idx) rowadd += (size_t)(*a.rowadd_idx) * a.rowadd_stride;
    const size_t zo = (size_t)z * a.bs_out;
    const size_t zr = (size_t)z * a.bs_res;
    const bool vec_ok = ((a.ldo & 3) == 0);

    // per-pixel quantities of this lane's TP pixel columns (compile-time indices: no scratch)
    int pj[TP]; bool prow[TP]; float pbj[TP];
    static_for<0, TP>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        pj[j] = p0 + (wp * TP + j) * 32 + l31;
        prow[j] = pj[j] < a.P;
        pbj[j] = (prow[j] && a.bias && a.bias_per_pixel) ? (float)a.bias[pj[j]] : 0.f;
    });

    if (a.out_f32) {
        // ---- fp32 output (attention scores): direct per-lane stores, no residual / mask / statistics
        static_for<0, TP>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            static_for<0, TQ>([&](auto Ic) {
                constexpr int iu = decltype(Ic)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = q0 + (wq * TQ + iu) * 32 + 8 * g + 4 * hh;
                    if (prow[j] && co < Qout) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[iu][j][4 * g + e];
                            if (a.bias_per_pixel) x += pbj[j];
                            else if (a.bias && (co + e) < a.Q) x += (float)a.bias[co + e];
                            v[e] = x * a.out_scale;
                        }
                        float* op = reinterpret_cast<float*>(a.out) + zo + (size_t)pj[j] * a.ldo + co;
                        if ((co + 3 < Qout) && vec_ok) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (co + e < Qout) op[e] = v[e]; }
                    }
                }
            });
        });
        return;
    }

    // ---- fp16 output: the accumulator tile is transposed through LDS so that every global access of the epilogue (output,
    //      residuals) is a 16-byte piece of a contiguous >=64-byte run of ONE pixel row (the K loop leaves lanes owning pixel
    //      COLUMNS, whose 8-byte stores would each touch a different row).  Each wave owns a private [32 pixels][CW] fp16 patch.
    constexpr int CWF = TQ * 32;                    // channels of this wave (non-GEGLU)
    constexpr int RSF = CWF + 4;                    // padded row stride (halves): 8-byte aligned, conflict-free b64 writes
    const int CW = geglu ? CWF / 2 : CWF;
    const int LPR = CW / 8;                         // lanes per pixel row in the read-back (8 channels = 16 B each)
    const int RPW = 64 / LPR;                       // pixel rows per read-back pass
    __syncthreads();                                // every wave is done with the staging ring
    h16* patch = smem + wave * (32 * RSF);
    const int rb_row = lane / LPR, rb_chunk = lane - rb_row * LPR;
    const bool rb_lane = rb_row < RPW;
    const int cw0 = geglu ? (q0 + wq * TQ * 32) / 2 : (q0 + wq * TQ * 32);   // first output channel of this wave
    const int co8 = cw0 + rb_chunk * 8;
    const bool ovec = ((a.ldo & 7) == 0) && (co8 + 7 < Qout);
    const bool r0vec = a.res0 && ((a.ldr0 & 7) == 0) && (co8 + 7 < Qout);
    const bool r1vec = a.res1 && ((a.ldr1 & 7) == 0) && (co8 + 7 < Qout);
    const bool want_stats = (a.stats != nullptr);
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }

    static_for<0, TP>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        // (1) lane = pixel column: bias / time embedding / activation, round to fp16, 8-byte LDS writes
        static_for<0, TQ>([&](auto Ic) {
            constexpr int iu = decltype(Ic)::value;
            constexpr int ig = (iu + 1 < TQ) ? iu + 1 : iu;
            if (!(geglu && (iu & 1))) {   // GEGLU: 32-row blocks alternate u | g; g is consumed with its u block
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int qw = q0 + (wq * TQ + iu) * 32 + 8 * g + 4 * hh;       // W-row index of reg 4g
                    const int lc = (geglu ? (iu / 2) * 32 : iu * 32) + 8 * g + 4 * hh;  // channel inside the wave patch
                    const int co = cw0 + lc;
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[iu][j][4 * g + e];
                        if (a.bias_per_pixel) x += pbj[j];
                        else if (a.bias && (qw + e) < a.Q) x += (float)a.bias[qw + e];
                        if (geglu) {
                            float gg = acc[ig][j][4 * g + e];
                            if (a.bias && (qw + 32 + e) < a.Q) gg += (float)a.bias[qw + 32 + e];
                            x = x * gelu_f(gg);
                        } else {
                            if (rowadd && (co + e) < Qout) x += rowadd[co + e];
                            if (a.act == LADI_ACT_SILU) x = silu_f(x);
                            else if (a.act == LADI_ACT_GELU) x = gelu_f(x);
                            else if (a.act == LADI_ACT_RELU) x = fmaxf(x, 0.f);
                        }
                        o[e] = (h16)(x * a.out_scale);
                    }
                    *reinterpret_cast<h16x4*>(patch + l31 * RSF + lc) = o;
                }
            }
        });
        __builtin_amdgcn_wave_barrier();
        // (2) lane = (pixel row, 8-channel chunk): residuals, mask, statistics, 16-byte coalesced stores
        const int pbase = p0 + (wp * TP + j) * 32;
        for (int r = 0; r < 32; r += RPW) {
            const int prw = r + rb_row;
            const int p = pbase + prw;
            if (rb_lane && prw < 32 && p < a.P && co8 < Qout) {
                const h16* src = patch + prw * RSF + rb_chunk * 8;
                const h16x4 lo = *reinterpret_cast<const h16x4*>(src);
                const h16x4 hi = *reinterpret_cast<const h16x4*>(src + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (float)lo[e]; v[4 + e] = (float)hi[e]; }
                if (a.res0) {
                    const h16* rp = a.res0 + zr + (size_t)p * a.ldr0 + co8;
                    if (r0vec) { const h16x8 rr = *reinterpret_cast<const h16x8*>(rp);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rr[e]; }
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (co8 + e < Qout) v[e] += (float)rp[e]; }
                }
                if (a.res1) {
                    const h16* rp = a.res1 + zr + (size_t)p * a.ldr1 + co8;
                    if (r1vec) { const h16x8 rr = *reinterpret_cast<const h16x8*>(rp);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rr[e]; }
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (co8 + e < Qout) v[e] += (float)rp[e]; }
                }
                if (a.mask) {
                    const float mk = 1.f - (float)a.mask[p];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= mk;
                }
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (h16)v[e];
                h16* op = reinterpret_cast<h16*>(a.out) + zo + (size_t)p * a.ldo + co8;
                if (ovec) *reinterpret_cast<h16x8*>(op) = o;
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (co8 + e < Qout) op[e] = o[e]; }
                if (want_stats) {   // statistics of the values as stored (fp16-rounded)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float q = (co8 + e < Qout) ? (float)o[e] : 0.f;
                        ssum[e] += q; ssq[e] += q * q;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    });

    // optional per-channel statistics of the OUTPUT (sum, sum of squares over this wave's TP*32 pixels) for the GroupNorm that
    // consumes it: plain stores of partial rows, no atomics (deterministic); the launcher guarantees sample alignment
    if (want_stats) {
        for (int k = 1; k < RPW; ++k) {
            const int srcl = lane + k * LPR;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s1 = __shfl(ssum[e], srcl), s2 = __shfl(ssq[e], srcl);
                if (lane < LPR) { ssum[e] += s1; ssq[e] += s2; }
            }
        }
        if (lane < LPR && co8 < Qout) {
            const size_t row = (size_t)pt * WP + wp;
            float* sp = a.stats + (row * Qout + co8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (co8 + e < Qout) { sp[2 * e] = ssum[e]; sp[2 * e + 1] = ssq[e]; }
        }
    }
}


}  // namespace
