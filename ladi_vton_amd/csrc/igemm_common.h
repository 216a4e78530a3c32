// Device-side pieces shared by the implicit-GEMM kernels (igemm.hip, igemm8.hip): LDS swizzle and the fused epilogue.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS bytes the fused epilogue uses: one transpose patch per wave + the per-W-row additive vector
template <int WQ, int WP, int TQ>
constexpr int igemm_epilogue_lds_bytes() { return WQ * WP * 32 * (TQ * 32 + 4) * 2 + WQ * TQ * 32 * 4; }

template <int BK>
__device__ __forceinline__ int swz(int row, int chunk) {
    if constexpr (BK == 64) return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
    else return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3);
}

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the implicit-GEMM kernels.  On entry every wave holds acc[TQ][TP] (32x32 MFMA accumulators) of its
// (TQ*32 channels) x (TP*32 pixels) sub-tile at channel offset q0 + wq*TQ*32, pixel offset p0 + wp*TP*32; the staging ring is dead
// (the caller has drained every LDS-DMA).  igemm_epilogue() dispatches to
//   * igemm_epilogue_fast     fp16 output with 16-byte aligned rows (every layer of the UNet / VAE / EMASC except conv_out),
//   * igemm_epilogue_generic  everything else (fp32 output, narrow / unaligned outputs): element-wise bounds checks.
// ------------------------------------------------------------------------------------------------
template <int WQ, int WP, int TQ, int TP>
__device__ __forceinline__ void igemm_epilogue_generic(const IGemmArgs& a, f32x16 (&acc)[TQ][TP], h16* smem, const int q0, const int p0,
                                                       const int pt, const int z, const int wave, const int lane, const int pstr = 32) {
    const int wq = wave / WP, wp = wave % WP;
    const int l31 = lane & 31, hh = lane >> 5;
    const float bmul = a.bias_mul != 0.f ? a.bias_mul : 1.f;
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
        pj[j] = p0 + (wp * TP + j) * pstr + l31;
        prow[j] = pj[j] < a.P;
        pbj[j] = (prow[j] && a.bias && a.bias_per_pixel) ? (float)a.bias[pj[j]] * bmul : 0.f;
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
                            else if (a.bias && (co + e) < a.Q) x += (float)a.bias[co + e] * bmul;
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
                        else if (a.bias && (qw + e) < a.Q) x += (float)a.bias[qw + e] * bmul;
                        if (geglu) {
                            float gg = acc[ig][j][4 * g + e];
                            if (a.bias && (qw + 32 + e) < a.Q) gg += (float)a.bias[qw + 32 + e] * bmul;
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
        const int pbase = p0 + (wp * TP + j) * pstr;
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



// Fast path.  Three things bound an epilogue on this chip and shape it:
//   * VMEM counters are in-order and count stores too: a load issued after stores cannot be waited for without draining those stores
//     (~1 us each).  So the global LOADS of a pixel sub-tile (the residual rows) are all issued before its first STORE, and the
//     per-channel vectors (bias, time-embedding row) never come from global memory inside the loops: they are summed once per
//     workgroup into an LDS vector;
//   * lanes own pixel COLUMNS after the K loop, NHWC rows want lanes along channels: each wave transposes its 32-pixel sub-tiles
//     through a private LDS patch so that every global access is a 16-byte piece of a contiguous run of one pixel row;
//   * all trip counts are compile-time (GEGLU is a template flag), so the passes of a sub-tile are unrolled and their LDS reads, loads
//     and stores overlap.
template <int WQ, int WP, int TQ, int TP, bool GEGLU>
__device__ __forceinline__ void igemm_epilogue_fast(const IGemmArgs& a, f32x16 (&acc)[TQ][TP], h16* smem, const int q0, const int p0,
                                                    const int pt, const int z, const int wave, const int lane, const int pstr = 32) {
    static_assert(!GEGLU || (TQ % 2 == 0), "GEGLU pairs 32-row blocks (u | g)");
    constexpr int NW = WQ * WP, NT = 64 * NW, BQ = WQ * TQ * 32;
    constexpr int CWF = TQ * 32;                    // W rows of this wave
    constexpr int RSF = CWF + 4;                    // padded patch row stride (halves): 8-byte aligned, conflict-free b64 writes
    constexpr int CW = GEGLU ? CWF / 2 : CWF;       // output channels of this wave
    constexpr int LPR = CW / 8;                     // lanes per pixel row in the read-back (8 channels = 16 B each)
    constexpr int RPW = 64 / LPR;                   // pixel rows per read-back pass
    constexpr int NPASS = (32 + RPW - 1) / RPW;
    const int wq = wave / WP, wp = wave % WP;
    const int l31 = lane & 31, hh = lane >> 5;
    const float bmul = a.bias_mul != 0.f ? a.bias_mul : 1.f;
    const int Qout = GEGLU ? a.Q / 2 : a.Q;
    const size_t zo = (size_t)z * a.bs_out;
    const size_t zr = (size_t)z * a.bs_res;

    __syncthreads();                                // every wave is done with the staging ring
    h16* patch = smem + wave * (32 * RSF);
    float* cadd = reinterpret_cast<float*>(smem + NW * (32 * RSF));      // [BQ] per-W-row additive vector: bias (+ time-embedding row)
    {
        const float* rowadd = a.rowadd;
        if (rowadd && a.rowadd_idx) rowadd += (size_t)(*a.rowadd_idx) * a.rowadd_stride;
        for (int c = threadIdx.x; c < BQ; c += NT) {
            const int q = q0 + c;
            float v = 0.f;
            if (q < a.Q) {
                if (a.bias && !a.bias_per_pixel) v = (float)a.bias[q] * bmul;
                if (!GEGLU && rowadd) v += rowadd[q];
            }
            cadd[c] = v;
        }
    }
    __syncthreads();

    const int rb_row = lane / LPR, rb_chunk = lane - rb_row * LPR;
    const bool rb_lane = rb_row < RPW;
    const int cw0 = GEGLU ? (q0 + wq * CWF) / 2 : (q0 + wq * CWF);       // first output channel of this wave
    const int co8 = cw0 + rb_chunk * 8;
    const bool cok = rb_lane && co8 < Qout;
    const bool want_stats = (a.stats != nullptr);
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }

    static_for<0, TP>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        const int pbase = p0 + (wp * TP + j) * pstr;
        // (1) lane = pixel column: per-channel vector / activation, round to fp16, 8-byte LDS writes
        const int pcol = pbase + l31;
        const float pb = (a.bias && a.bias_per_pixel && pcol < a.P) ? (float)a.bias[pcol] * bmul : 0.f;
        // the activation is selected ONCE per sub-tile (wave-uniform switch around the unrolled loops): a per-element `if (act == ...)`
        // chain costs ~20 scalar branches per 4 values, more than all the arithmetic of this epilogue together
        auto step1 = [&](auto ActC) {
            constexpr int ACT = decltype(ActC)::value;
            static_for<0, TQ>([&](auto Ic) {
                constexpr int iu = decltype(Ic)::value;
                if constexpr (!(GEGLU && (iu & 1))) {   // GEGLU: 32-row blocks alternate u | g; g is consumed with its u block
                    const int wr0 = (wq * TQ + iu) * 32 + 4 * hh;                        // W row (inside the workgroup tile) of reg 0
                    const int lc0 = (GEGLU ? (iu / 2) * 32 : iu * 32) + 4 * hh;          // channel inside the wave patch
                    f32x4 ca[4], cg[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        ca[g] = *reinterpret_cast<const f32x4*>(cadd + wr0 + 8 * g);
                        if constexpr (GEGLU) cg[g] = *reinterpret_cast<const f32x4*>(cadd + wr0 + 8 * g + 32);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        h16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[iu][j][4 * g + e] + ca[g][e] + pb;
                            if constexpr (GEGLU) {
                                constexpr int ig = (iu + 1 < TQ) ? iu + 1 : iu;
                                x *= gelu_f(acc[ig][j][4 * g + e] + cg[g][e]);
                            } else if constexpr (ACT == LADI_ACT_SILU) x = silu_fast(x);
                            else if constexpr (ACT == LADI_ACT_GELU) x = gelu_f(x);
                            else if constexpr (ACT == LADI_ACT_RELU) x = fmaxf(x, 0.f);
                            o[e] = (h16)(x * a.out_scale);
                        }
                        *reinterpret_cast<h16x4*>(patch + l31 * RSF + lc0 + 8 * g) = o;
                    }
                }
            });
        };
        if constexpr (GEGLU) step1(IntC<LADI_ACT_GEGLU>{});
        else if (a.act == LADI_ACT_SILU) step1(IntC<LADI_ACT_SILU>{});
        else if (a.act == LADI_ACT_GELU) step1(IntC<LADI_ACT_GELU>{});
        else if (a.act == LADI_ACT_RELU) step1(IntC<LADI_ACT_RELU>{});
        else step1(IntC<LADI_ACT_NONE>{});
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");      // keep the loads below behind the patch write: the accumulators of this sub-tile are dead now
        // the read-back passes run in groups of at most GP: (1b) every global load of a group (residual rows, in read-back layout)
        // before the group's first store, (2) lane = (pixel row, 8-channel chunk): residuals, mask, statistics, 16-byte coalesced stores
        constexpr int GP = NPASS > 6 ? (NPASS + 1) / 2 : NPASS;
        static_for<0, (NPASS + GP - 1) / GP>([&](auto Gc) {
        constexpr int i0 = decltype(Gc)::value * GP;
        constexpr int i1 = (i0 + GP < NPASS) ? i0 + GP : NPASS;
        h16x8 r0v[GP], r1v[GP];
        bool pok[GP];
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int prw = i * RPW + rb_row;
            const int p = pbase + prw;
            pok[i - i0] = cok && prw < 32 && p < a.P;
            if (a.res0 && pok[i - i0]) r0v[i - i0] = *reinterpret_cast<const h16x8*>(a.res0 + zr + (size_t)p * a.ldr0 + co8);
            if (a.res1 && pok[i - i0]) r1v[i - i0] = *reinterpret_cast<const h16x8*>(a.res1 + zr + (size_t)p * a.ldr1 + co8);
        }
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int prw = i * RPW + rb_row;
            const int p = pbase + prw;
            if (pok[i - i0]) {
                const h16* src = patch + prw * RSF + rb_chunk * 8;
                const h16x4 lo = *reinterpret_cast<const h16x4*>(src);
                const h16x4 hi = *reinterpret_cast<const h16x4*>(src + 4);
                h16x8 o;
                if (a.res0 || a.res1 || a.mask) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = (float)lo[e]; v[4 + e] = (float)hi[e]; }
                    if (a.res0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)r0v[i - i0][e]; }
                    if (a.res1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)r1v[i - i0][e]; }
                    if (a.mask) {
                        const float mk = 1.f - (float)a.mask[p];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= mk;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (h16)v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
                }
                *reinterpret_cast<h16x8*>(reinterpret_cast<h16*>(a.out) + zo + (size_t)p * a.ldo + co8) = o;
                if (want_stats) {   // statistics of the values as stored (fp16-rounded)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float q = (float)o[e]; ssum[e] += q; ssq[e] += q * q; }
                }
            }
        }
        asm volatile("" ::: "memory");
        });
        __builtin_amdgcn_wave_barrier();
    });

    // optional per-channel statistics of the OUTPUT (sum, sum of squares over this wave's TP*32 pixels) for the GroupNorm that
    // consumes it: plain stores of partial rows, no atomics (deterministic); the launcher guarantees sample alignment
    if (want_stats) {
#pragma unroll
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
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<float4*>(sp + 4 * e) = make_float4(ssum[2 * e], ssq[2 * e], ssum[2 * e + 1], ssq[2 * e + 1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// In-launch split-K combine (round 4; replaces the separate splitk_reduce_kernel pass for the kernels whose every wave reaches the
// epilogue).  Each K-slice workgroup PUBLISHES its fp32 accumulators to its slab with write-through (sc1) 16-byte stores -- one private,
// fully coalesced [wave][fragment][lane] image per (slice, tile), no tile is shared between workgroups of one slice -- drains them
// (`s_waitcnt vmcnt(0)` in every wave, then the workgroup barrier) and takes a ticket on the tile's arrival counter (relaxed, agent
// scope).  The workgroup that draws the last ticket reads the S slabs back with sc1 loads (served past the L1; the producers' data is
// already at the memory side) IN SLICE ORDER -- its own included, so the sum does not depend on who arrives last: bitwise reproducible --
// re-arms the counter and runs the ordinary fused epilogue; everyone else is done.  This is the write-through form of the hand-off in
// cdna_hip_programming.md (split-K seam: sc1 slab stores -> vmcnt(0) -> barrier -> relaxed agent fetch_add; reducer sc1 loads): correct
// for ANY placement of a tile's slices over CUs / XCDs.  Counters live in a caller-owned, zero-initialised buffer that every launch
// leaves zeroed (hipGraph replay needs no memset node).
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int WQ, int WP, int TQ, int TP>
__device__ __forceinline__ bool igemm_splitk_combine(const IGemmArgs& a, f32x16 (&acc)[TQ][TP], h16* smem, const int q0, const int pt,
                                                     const int z, const int wave, const int lane) {
    constexpr int NW = WQ * WP, NV = TQ * TP * 4, BQ = WQ * TQ * 32, BP = WP * TP * 32;
    constexpr unsigned WG_BYTES = (unsigned)NW * NV * 64 * 16;
    const int nq = (a.Q + BQ - 1) / BQ, np = (a.P + BP - 1) / BP;
    const int tile = pt * nq + q0 / BQ, ntiles = nq * np;
    const int S = a.splitk;
    char* const base = reinterpret_cast<char*>(a.sk_ws);
    const unsigned voff = (unsigned)(wave * NV * 64 + lane) * 16u;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + ((size_t)z * ntiles + tile) * WG_BYTES, 0, WG_BYTES, 0x00020000);
        static_for<0, TQ>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            static_for<0, TP>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, voff + (unsigned)(((i * TP + j) * 4 + g) * 1024), 0, /*sc1*/ 16);
                }
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // every wave's slab stores are out; the staging ring is dead
    int* const tk = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) *tk = __hip_atomic_fetch_add(a.sk_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *tk;
    if (ticket != S - 1) return false;
    static_for<0, TQ>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        static_for<0, TP>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            f32x4 sum[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) sum[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int zz = 0; zz < S; ++zz) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + ((size_t)zz * ntiles + tile) * WG_BYTES, 0, WG_BYTES, 0x00020000);
                u32x4_t r[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) r[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)(((i * TP + j) * 4 + g) * 1024), 0, /*sc1*/ 16);
#pragma unroll
                for (int g = 0; g < 4; ++g) sum[g] += __builtin_bit_cast(f32x4, r[g]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = sum[g][e];
        });
    });
    if (threadIdx.x == 0) __hip_atomic_store(a.sk_cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();   // every wave has read its ticket from smem[0] before the epilogue reuses that word for its transpose patches
    return true;
}

// pstr: pixel-index distance between consecutive 32-pixel sub-tiles of the workgroup tile -- 32 for tiles of consecutive pixels (every kernel
// but the 2-D blocked halo form, whose sub-tiles are 32-pixel segments of consecutive IMAGE ROWS: pstr = image width, p0 = first pixel of the block)
template <int WQ, int WP, int TQ, int TP>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& a, f32x16 (&acc)[TQ][TP], h16* smem, const int q0, const int p0,
                                               const int pt, const int z_in, const int wave, const int lane, const int pstr = 32) {
    int z = z_in;
    if (a.splitk > 1 && a.sk_cnt) {      // in-launch split-K: only the last-arriving slice of a tile runs the fused epilogue
        if (!igemm_splitk_combine<WQ, WP, TQ, TP>(a, acc, smem, q0, pt, z, wave, lane)) return;
        z = 0;
    }
    const bool geglu = (a.act == LADI_ACT_GEGLU);
    const int Qout = geglu ? a.Q / 2 : a.Q;
    // workgroup-uniform: fp16 rows whose 8-channel chunks are whole and 16-byte aligned (ld % 8 == 0 with 16-byte aligned bases)
    const bool fast = !a.out_f32 && !(a.ldo & 7) && !(Qout & 7) && (!a.res0 || !(a.ldr0 & 7)) && (!a.res1 || !(a.ldr1 & 7)) &&
                      !((reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.res0) | reinterpret_cast<uintptr_t>(a.res1)) & 15) &&
                      !(((size_t)z * a.bs_out | (size_t)z * a.bs_res) & 7) && (!a.stats || !(reinterpret_cast<uintptr_t>(a.stats) & 15));
    if (fast) {
        if constexpr (TQ % 2 == 0) {
            if (geglu) { igemm_epilogue_fast<WQ, WP, TQ, TP, true>(a, acc, smem, q0, p0, pt, z, wave, lane, pstr); return; }
        }
        if (!geglu) { igemm_epilogue_fast<WQ, WP, TQ, TP, false>(a, acc, smem, q0, p0, pt, z, wave, lane, pstr); return; }
    }
    igemm_epilogue_generic<WQ, WP, TQ, TP>(a, acc, smem, q0, p0, pt, z, wave, lane, pstr);
}

}  // namespace
