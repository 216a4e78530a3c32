// "X-stationary" linear layers for shallow reductions (K = 320 / 640: the token-wise projections and the GEGLU up-projection of
// the two high-resolution transformer levels of the UNet; diffusers BasicTransformerBlock to_q / to_k / to_v /
// to_out / proj_in / ff.net[0]).
//
// The tiled igemm streams BOTH operands through LDS; with K this shallow a tile lives for only 5-10 K steps, so its pipeline
// fill, its epilogue and the repeated re-staging of the pixel panel (once per Q tile) dominate: 200-430 TFLOP/s measured.
// Here a wave keeps its pixel panel X[32*PB pixels][K] in REGISTERS as ready-made MFMA B fragments for the whole kernel
// (K/16 x 4 VGPRs per 32 pixels, loaded from HBM exactly once), and only the weights stream through a 3-deep LDS-DMA ring in
// stages of [32 output channels][320 k] shared by the 4 waves of the workgroup:
//     D[q][p] (32 x 32*PB per wave) += W_stage[q][k] * X[p][k]
//   * staging traffic per MFMA drops from 1 KB (64x64 tile) to 128-256 B, LDS reads to one ds_read_b128 per PB MFMAs;
//   * every 32-channel block finishes after K/320 stages and is written out at once (bias, optional residual or GEGLU gate,
//     fp16, transpose through a per-wave LDS patch -> 64-byte row segments): no long-lived accumulator, no fill/drain per tile;
//   * the residual tile arrives by LDS-DMA straight into the patch one channel block ahead, so the loop contains no ordinary
//     global load (the compiler would drain the DMA ring for its result); the only vector-memory operations in the loop are
//     DMAs and output stores, all issued unconditionally (the launcher requires whole tiles), and every hand-over uses a counted
//     s_waitcnt vmcnt(N) where N = operations issued since the awaited one (running counter; gfx9 retires VMEM in order).
// Restrictions (ladi_linear_xs_eligible): 1x1, single source, K in {320, 640}; epilogue = bias, bias + one residual, or GEGLU;
// P a multiple of the workgroup's pixel panel; Q a multiple of 32 (64 for GEGLU) split evenly over gridDim.y.
#include "common.h"
#include "kernels.h"
#include <cstdio>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int XS_KB = 320;                       // k extent of one weight stage
constexpr int XS_STAGE = 32 * XS_KB * 2;         // bytes per stage: 32 rows x 640 B
constexpr int XS_DMA = XS_STAGE / 16 / 256;      // 16-byte DMA pieces per thread per stage (5)
constexpr int XS_PLD = 40;                       // halves per patch row (32 channels + 8 pad): 80 B, 16-byte aligned rows
constexpr int XS_PATCH = 32 * XS_PLD * 2;        // bytes per wave patch (MODE 0 / 2)
constexpr int XS_RPATCH = 32 * 64;               // bytes per residual landing / transpose buffer (MODE 1: dense swizzled rows, x2)
constexpr int XS_SMEM_MAX = 80 * 1024;           // two workgroups per CU
constexpr int XS_SMEM_MAX3 = 53 * 1024;          // three workgroups per CU (NST = 2 forms)

#define XS_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform; clamping down is conservative)
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n < 24 ? n : 24) {
        XS_CASE(1) XS_CASE(2) XS_CASE(3) XS_CASE(4) XS_CASE(5) XS_CASE(6) XS_CASE(7) XS_CASE(8) XS_CASE(9) XS_CASE(10) XS_CASE(11)
        XS_CASE(12) XS_CASE(13) XS_CASE(14) XS_CASE(15) XS_CASE(16) XS_CASE(17) XS_CASE(18) XS_CASE(19) XS_CASE(20) XS_CASE(21)
        XS_CASE(22) XS_CASE(23) XS_CASE(24)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#undef XS_CASE

// KH = K / 320, PB = 32-pixel blocks per wave, MODE 0: bias | 1: bias + residual (PB == 1) | 2: GEGLU (PB == 1; 32-row blocks of W
// alternate u | g), PRE: rewrite of the pixel panel in the prologue -- 1: LayerNorm (a compile-time variant: a run-time branch around
// the rewrite of the 160-VGPR panel makes the compiler keep both versions live and spill ~250 VGPRs), 2: GroupNorm affine
// x * scale[n][c] + shift[n][c] (round 4: the GroupNorm in front of proj_in has no activation, so its apply pass -- a full HBM round
// trip of the block input -- folds into the consumer's register panel; statistics / finalize stay where they were).
// grid = (P / (128*PB), channel slices); block = 4 waves.
// NST: depth of the weight ring.  3 (two stages in flight) at two workgroups per CU is the round-1 form; NST = 2 (round 4) trades
// the second in-flight stage for a THIRD workgroup per CU (52 KB of LDS instead of 72; the K = 320 kernels need 127-154 VGPRs, so the
// registers allow three waves per SIMD): the other workgroups' MFMAs cover the shorter prefetch distance.
template <int KH, int PB, int MODE, int PRE, int NST>
__global__ __launch_bounds__(256, (NST == 2 ? 3 : 2)) void linear_xs_kernel(const IGemmArgs a, int qb_per_slice) {
    constexpr bool LN = (PRE == 1);
    constexpr int XS_NST = NST;
    static_assert(NST == 2 || NST == 3, "weight ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* ring = smem_raw;
    char* patch_base = smem_raw + XS_NST * XS_STAGE;
    constexpr int PATCH_BYTES = (MODE == 1) ? 2 * XS_RPATCH : XS_PATCH;
    h16* bias_s = reinterpret_cast<h16*>(patch_base + 4 * PATCH_BYTES);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    constexpr int KS = KH * (XS_KB / 16);            // k16 steps over the whole reduction
    constexpr int HALVES = (MODE == 2) ? 2 : 1;      // 32-row weight blocks per output channel block
    constexpr int SUB = HALVES * KH;                 // stages per output channel block
    constexpr int STORES = 2 * PB;                   // output store instructions per finished channel block
    const int p0 = blockIdx.x * (128 * PB) + wave * (32 * PB);
    const int K = KH * XS_KB;
    const int qb0 = blockIdx.y * qb_per_slice;       // first 32-row weight block of this slice
    const int nstage = qb_per_slice * KH;
    const int nblk = qb_per_slice / HALVES;          // output channel blocks of this slice
    const int ob0 = qb0 / HALVES;                    // first output channel block

    // ---- weight DMA addressing: LDS position i = j*256 + tid (16-byte units) -> row i/40, physical chunk i%40; the XOR swizzle
    //      of the ds_read_b128 side is applied to the SOURCE chunk (the LDS image of a DMA is lane-linear)
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.W), 0, 0x7FFFFFFF, 0x00020000);
    unsigned wrel[XS_DMA];
#pragma unroll
    for (int j = 0; j < XS_DMA; ++j) {
        const int i = j * 256 + tid;
        const int row = i / (XS_KB / 8), cph = i - row * (XS_KB / 8);
        const int c = cph ^ ((row >> 1) & 7);
        wrel[j] = (unsigned)((row * K + c * 8) * 2);
    }
    int vm_issued = 0;                               // running count of this wave's VMEM operations (wave-uniform)
    auto issue_w = [&](int s) {   // stage s = (weight block s / KH, k half s % KH)
        const int qb = qb0 + s / KH, kh = s - (s / KH) * KH;
        const unsigned base = (unsigned)((qb * 32 * K + kh * XS_KB) * 2);
        char* dst = ring + (s % XS_NST) * XS_STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < XS_DMA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + j * 4096), 16, wrel[j] + base, 0, 0, 0);
        vm_issued += XS_DMA;
    };
    // ---- residual tile of output block ob (MODE 1): 32 pixels x 64 B by two DMAs into the wave's landing buffer (ob & 1), lane =
    //      (row 16r + lane/4, physical chunk lane%4), chunk swizzle (row >> 2) & 3 on the source side
    char* patch_w = patch_base + wave * PATCH_BYTES;
    const int rb_row = lane >> 2, rb_chunk = lane & 3;
    const __amdgpu_buffer_rsrc_t rsr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(MODE == 1 ? a.res0 : a.W), 0, 0x7FFFFFFF, 0x00020000);
    unsigned rrel[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = rb_row + 16 * r;
        rrel[r] = (unsigned)(((p0 + row) * a.ldr0 + ((rb_chunk ^ ((row >> 2) & 3)) << 3)) * 2);
    }
    auto issue_res = [&](int ob) {
        char* dst = patch_w + ((ob - ob0) & 1) * XS_RPATCH;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsr, (lds_ptr_t)(dst + r * 1024), 16, rrel[r] + (unsigned)(ob * 64), 0, 0, 0);
        vm_issued += 2;
    };

    // ---- the wave's pixel panel as MFMA B fragments: lane = pixel l31 (of block pb), k half hh
    h16x8 xf[PB][KS];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const h16* xp = a.src0 + (size_t)(p0 + pb * 32 + l31) * a.ld0 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[pb][ks] = *reinterpret_cast<const h16x8*>(xp + ks * 16);
    }
    // marks = value of vm_issued right after an awaited operation was issued: m_cur -> the stage the next hand-over waits for,
    // m_next -> the stage after it; mr_cur / mr_next likewise for residual tiles
    int m_cur, m_next = 0, mr_cur = 0, mr_next = 0;
    issue_w(0); m_cur = vm_issued;
    if (NST == 3 && nstage > 1) { issue_w(1); m_next = vm_issued; }
    if (MODE == 1) { issue_res(ob0); mr_cur = vm_issued; }
    // ---- optional fused LayerNorm (a.ln_gamma): the lane pair (l31, hh = 0 / 1) holds one whole pixel row, so the statistics are two
    //      register passes and one cross-half shuffle each; same arithmetic and rounding point (fp16 result) as layernorm_kernel
    if constexpr (LN) {
        // pin(): an ordered no-op on one 4-VGPR fragment.  The three passes are fully unrolled over a 160-VGPR panel; left alone the
        // scheduler hoists every fp16 -> fp32 conversion to the top (320+ live floats, 500+ spilled VGPRs).  A pin before and after
        // each fragment's arithmetic chains the fragments one after another, so only one fragment's floats are live at a time.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto pin = [](h16x8& v) {
            i32x4 t = __builtin_bit_cast(i32x4, v);
            asm volatile("" : "+v"(t));
            v = __builtin_bit_cast(h16x8, t);
        };
        const float invK = 1.f / (float)K;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                pin(xf[pb][ks]);
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) t += (float)xf[pb][ks][e];
                s += t;
                asm volatile("" : "+v"(s));
            }
            s += __shfl_xor(s, 32);
            const float mean = s * invK;
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                pin(xf[pb][ks]);
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xf[pb][ks][e] - mean; t += d * d; }
                q += t;
                asm volatile("" : "+v"(q));
            }
            q += __shfl_xor(q, 32);
            const float rstd = rsqrtf(q * invK + a.ln_eps);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                pin(xf[pb][ks]);
                const h16x8 g = *reinterpret_cast<const h16x8*>(a.ln_gamma + ks * 16 + hh * 8);
                const h16x8 b = *reinterpret_cast<const h16x8*>(a.ln_beta + ks * 16 + hh * 8);
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)xf[pb][ks][e] - mean) * rstd * (float)g[e] + (float)b[e]);
                xf[pb][ks] = o;
                pin(xf[pb][ks]);
            }
        }
    }
    // ---- optional fused GroupNorm affine (a.gn_ss: [n][K][2] floats = (scale, shift) per sample and channel, written by gn_finalize):
    //      a 32-pixel block lies inside one sample (the launcher requires gn_hw % 32 == 0); same arithmetic and rounding point (fp16
    //      result) as gn_apply_kernel without activation
    if constexpr (PRE == 2) {
        // the (scale, shift) rows of this workgroup's sample: K x 2 floats, staged ONCE through LDS by all four waves (coalesced 16-byte
        // loads) and read back as broadcasts -- per-lane global loads of the table cost twice the bytes of the panel itself
        float* ss_s = reinterpret_cast<float*>(bias_s + qb_per_slice * 32);
        {
            const int n_wg = (int)(blockIdx.x * (128 * PB)) / a.gn_hw;        // the launcher guarantees one sample per workgroup panel
            const float4* src = reinterpret_cast<const float4*>(a.gn_ss + (size_t)n_wg * K * 2);
            for (int i = tid; i < K / 2; i += 256) reinterpret_cast<float4*>(ss_s)[i] = src[i];
        }
        __syncthreads();
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto pin = [](h16x8& v) {
            i32x4 t = __builtin_bit_cast(i32x4, v);
            asm volatile("" : "+v"(t));
            v = __builtin_bit_cast(h16x8, t);
        };
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                pin(xf[pb][ks]);
                const float4* s4 = reinterpret_cast<const float4*>(ss_s + (ks * 16 + hh * 8) * 2);
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 t = s4[e];
                    o[2 * e] = (h16)((float)xf[pb][ks][2 * e] * t.x + t.y);
                    o[2 * e + 1] = (h16)((float)xf[pb][ks][2 * e + 1] * t.z + t.w);
                }
                xf[pb][ks] = o;
                pin(xf[pb][ks]);
            }
        }
    }
    // ---- bias of this slice -> LDS, last in the prologue (its wait drains everything above, which stage 0 needs anyway)
    for (int i = tid; i < qb_per_slice * 32; i += 256) bias_s[i] = a.bias ? a.bias[qb0 * 32 + i] : (h16)0.f;

    // Round 6, GEGLU only: the accumulators of an output channel block start from the block's BIAS rows (lane = pixel column, register 4 g + e
    // = row 8 g + 4 hh + e: four 8-byte broadcast reads per 32-row block, converted here) instead of zero, so the finished block needs no bias
    // pass: the additions are gone and the conversions sit in front of the block's MFMAs instead of in the VALU-bound tail behind them.
    // (fp32 rows in LDS would save the conversions too, but push the three-workgroup form of the 64x48-level GEGLU past its 53 KB.)  The plain
    // and residual forms keep "sum, then + bias": they are not VALU-bound, and the residual form is tested BIT-equal to the tiled kernels' epilogue.
    constexpr bool BIAS_INIT = (MODE == 2);
    f32x16 acc[HALVES][PB];
    auto acc_init = [&](int wblk) {      // wblk: first 32-row weight block (inside this slice) of the output block about to be accumulated
        if constexpr (!BIAS_INIT) {
#pragma unroll
            for (int h = 0; h < HALVES; ++h)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][pb][r] = 0.f;
            return;
        }
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            const h16* bp = bias_s + (wblk + h) * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const h16x4 b4 = *reinterpret_cast<const h16x4*>(bp + 8 * g);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[h][pb][4 * g + e] = (float)b4[e];
            }
        }
    };

    h16* outp = reinterpret_cast<h16*>(a.out);
    // swizzled A-fragment addresses: chunk (2*k16 + hh) ^ swz == ((2*k16) & ~7) + (((2*k16) & 7) ^ t) with t = hh ^ swz, so four
    // lane offsets (k16 & 3) plus a compile-time displacement cover every k step
    const int tsw = hh ^ ((l31 >> 1) & 7);
    int aoff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) aoff[m] = l31 * (XS_KB * 2) + ((tsw ^ (2 * m)) << 4);

    for (int s0 = 0; s0 < nstage; s0 += SUB) {
        const int ob = ob0 + s0 / SUB;
        static_for<0, SUB>([&](auto Sc) {
            constexpr int sub = decltype(Sc)::value;
            constexpr int half = sub / KH, kh = sub % KH;
            const int s = s0 + sub;
            if (s == 0) {
                // prologue: X panel, bias, stages 0 and 1, first residual.  The builtin (not asm) form lets the compiler's own
                // scoreboard see that the X-panel loads have landed; otherwise it re-waits for them down to vmcnt(0) inside stage 0
                __builtin_amdgcn_s_waitcnt(BIAS_INIT ? 0x0070 : 0x0F70);   // vmcnt(0) (+ lgkmcnt(0): the bias rows are read right behind the barrier)
            } else wait_vm_n(vm_issued - m_cur);      // stage s has landed (everything issued after it may stay in flight)
            asm volatile("s_barrier" ::: "memory");
            if (s == 0) acc_init(0);                  // the bias rows are in LDS behind the first barrier
            if (NST == 3) m_cur = m_next;
            if (MODE == 1 && sub == 0 && ob + 1 < ob0 + nblk) { issue_res(ob + 1); mr_next = vm_issued; }
            if (NST == 3) { if (s + 2 < nstage) { issue_w(s + 2); m_next = vm_issued; } }
            else if (s + 1 < nstage) { issue_w(s + 1); m_cur = vm_issued; }     // two slots: the one stage s - 1 just left is refilled
            const char* sW = ring + (s % XS_NST) * XS_STAGE;
#pragma unroll
            for (int k16 = 0; k16 < XS_KB / 16; ++k16) {
                const h16x8 af = *reinterpret_cast<const h16x8*>(sW + aoff[k16 & 3] + (((2 * k16) & ~7) << 4));
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    acc[half][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, xf[pb][kh * (XS_KB / 16) + k16], acc[half][pb], 0, 0, 0);
            }
            if (sub == SUB - 1) {
                // ---- output channel block finished: bias (+ residual | GEGLU gate), fp16, transpose through LDS, 16-byte stores
                const h16* bsl = bias_s + (s0 / KH) * 32;   // bias rows of this block's first weight block (plain / residual forms)
                if (MODE == 1) {
                    wait_vm_n(vm_issued - mr_cur);          // this block's residual tile has landed in the wave's buffer
                    mr_cur = mr_next;
                    char* buf = patch_w + ((ob - ob0) & 1) * XS_RPATCH;
                    const int rsw4 = (l31 >> 2) & 3;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int lc = 8 * g + 4 * hh;
                        const h16x4 b4 = *reinterpret_cast<const h16x4*>(bsl + lc);
                        h16x4* cell = reinterpret_cast<h16x4*>(buf + l31 * 64 + ((g ^ rsw4) << 4) + 8 * hh);
                        const h16x4 rr = *cell;
                        h16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const h16 x = (h16)(acc[0][0][4 * g + e] + (float)b4[e]);   // same rounding points as the igemm epilogue
                            o[e] = (h16)((float)x + (float)rr[e]);
                        }
                        *cell = o;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int row = rb_row + 16 * r;
                        const h16x8 v = *reinterpret_cast<const h16x8*>(buf + r * 1024 + lane * 16);
                        *reinterpret_cast<h16x8*>(outp + (size_t)(p0 + row) * a.ldo + ob * 32 + ((rb_chunk ^ ((row >> 2) & 3)) << 3)) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                    h16* patch = reinterpret_cast<h16*>(patch_w);
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int lc = 8 * g + 4 * hh;
                            h16x4 o;
                            if (MODE == 2) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[0][pb][4 * g + e] * gelu_f(acc[HALVES - 1][pb][4 * g + e]));
                            } else {
                                const h16x4 b4 = *reinterpret_cast<const h16x4*>(bsl + lc);
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[0][pb][4 * g + e] + (float)b4[e]);
                            }
                            *reinterpret_cast<h16x4*>(patch + l31 * XS_PLD + lc) = o;
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int row = rb_row + 16 * r;
                            const h16x8 v = *reinterpret_cast<const h16x8*>(patch + row * XS_PLD + rb_chunk * 8);
                            *reinterpret_cast<h16x8*>(outp + (size_t)(p0 + pb * 32 + row) * a.ldo + ob * 32 + rb_chunk * 8) = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                vm_issued += STORES;
                acc_init((s0 + SUB) / KH < qb_per_slice ? (s0 + SUB) / KH : 0);      // next block's bias rows (the last block re-reads block 0: unused)
            }
        });
    }
}

template <int KH, int PB, int MODE, int PRE, int NST = 3>
int launch_xs_ln(const IGemmArgs& a, int qs, hipStream_t st) {
    const int qb_per_slice = (a.Q / 32) / qs;
    const int smem = NST * XS_STAGE + 4 * (MODE == 1 ? 2 * XS_RPATCH : XS_PATCH) + ((qb_per_slice * 32 * 2 + 15) & ~15) + (PRE == 2 ? KH * XS_KB * 8 : 0);
    if (smem > (NST == 2 ? XS_SMEM_MAX3 : XS_SMEM_MAX)) return -10;
    auto kfn = linear_xs_kernel<KH, PB, MODE, PRE, NST>;
    static unsigned long long attr_done = 0;
    if (ladi_ensure_dyn_lds(reinterpret_cast<const void*>(kfn), NST == 2 ? XS_SMEM_MAX3 : XS_SMEM_MAX, attr_done)) return -10;
    dim3 grid((unsigned)(a.P / (128 * PB)), (unsigned)qs);
    hipLaunchKernelGGL(kfn, grid, dim3(256), smem, st, a, qb_per_slice);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

template <int KH, int PB, int MODE, int NST = 3>
int launch_xs(const IGemmArgs& a, int qs, hipStream_t st) {
    if constexpr (MODE == 0) { if (a.gn_ss) return launch_xs_ln<KH, PB, MODE, 2, NST>(a, qs, st); }
    return a.ln_gamma ? launch_xs_ln<KH, PB, MODE, 1, NST>(a, qs, st) : launch_xs_ln<KH, PB, MODE, 0, NST>(a, qs, st);
}

}  // namespace

// pb: 32-pixel blocks per wave (1 or 2), qs: output-channel slices over gridDim.y
bool ladi_linear_xs_eligible(const IGemmArgs& a, int batch, int pb, int qs, int nst) {
    if (nst == 2 && (a.K != 320 || pb != 1 || a.res0)) return false;     // three-workgroup form: K = 320, 32 pixels per wave, plain / GEGLU
    if (nst != 2 && nst != 3) return false;
    if (batch != 1 || a.ksize != 1 || a.stride != 1 || a.ups || a.C1 || a.src1 || a.K != a.C0) return false;
    const bool geglu = a.act == LADI_ACT_GEGLU;
    const bool res = a.res0 != nullptr;
    if (!(a.K == 320 || (a.K == 640 && pb == 1))) return false;
    if ((geglu || res) && pb != 1) return false;
    if ((a.ldw && a.ldw != a.K) || (a.ld0 % 8) || (a.ldo % 8)) return false;
    if (a.out_f32 || (a.act != LADI_ACT_NONE && !geglu) || a.rowadd || a.bias_per_pixel || a.mask || a.res1 || a.stats) return false;
    if (geglu && res) return false;
    if (res && ((a.ldr0 % 8) || (size_t)a.P * a.ldr0 * 2 >= 0x7FFFFFFFull)) return false;
    if (a.out_scale != 1.f || a.splitk > 1 || (a.bias_mul != 0.f && a.bias_mul != 1.f)) return false;
    // GroupNorm affine: plain projections only; a workgroup's pixel panel (128 * pb pixels) must lie inside one sample
    if (a.gn_ss && (geglu || res || a.ln_gamma || a.gn_hw <= 0 || (a.gn_hw % (128 * pb)))) return false;
    const int unit = geglu ? 64 : 32;
    if ((a.Q % unit) || ((a.Q / unit) % qs) || (a.P % (128 * pb))) return false;
    if ((size_t)a.Q * a.K * 2 >= 0x7FFFFFFFull) return false;
    if (nst * XS_STAGE + 4 * (res ? 2 * XS_RPATCH : XS_PATCH) + (a.Q / qs) * 2 + 16 + (a.gn_ss ? a.K * 8 : 0) > (nst == 2 ? XS_SMEM_MAX3 : XS_SMEM_MAX)) return false;
    return true;
}

// the kernel symbol (as rocprofv3 prints it) ladi_launch_linear_xs launches for these arguments: <KH, PB, MODE, LN>
void ladi_linear_xs_symbol(const IGemmArgs& a, int pb, int nst, char* out, int n) {
    const int kh = a.K == 320 ? 1 : 2, mode = a.act == LADI_ACT_GEGLU ? 2 : (a.res0 ? 1 : 0);
    const int pbb = (mode == 0 && a.K == 320 && pb == 2) ? 2 : 1;
    snprintf(out, (size_t)n, "linear_xs_kernel<%d, %d, %d, %d, %d>", kh, pbb, mode, (mode == 0 && a.gn_ss) ? 2 : (a.ln_gamma ? 1 : 0), nst);
}

int ladi_launch_linear_xs(const IGemmArgs& a, int pb, int qs, hipStream_t st, int nst) {
    if (!ladi_linear_xs_eligible(a, 1, pb, qs, nst)) return -1;
    if (nst == 2) return a.act == LADI_ACT_GEGLU ? launch_xs<1, 1, 2, 2>(a, qs, st) : launch_xs<1, 1, 0, 2>(a, qs, st);
    if (a.act == LADI_ACT_GEGLU) return a.K == 320 ? launch_xs<1, 1, 2>(a, qs, st) : launch_xs<2, 1, 2>(a, qs, st);
    if (a.res0) return a.K == 320 ? launch_xs<1, 1, 1>(a, qs, st) : launch_xs<2, 1, 1>(a, qs, st);
    if (a.K == 320) return pb == 2 ? launch_xs<1, 2, 0>(a, qs, st) : launch_xs<1, 1, 0>(a, qs, st);
    return launch_xs<2, 1, 0>(a, qs, st);
}
