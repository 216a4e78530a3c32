// Loader / consumer implicit GEMM (round 3), reached through ladi_launch_igemm (cfg 62..).
//
// Why.  The global -> LDS path of a CU delivers 32 B/clk with one K tile in flight and ~45-48 B/clk with 64+ KiB continuously in flight
// (tools/lds_dma_rate.hip, tools/dma_conv_pattern.hip) -- and an LDS-DMA instruction that the texture addresser cannot accept yet stalls
// the ISSUING wave.  Instruction issue is in order, so in the ring kernels (igemm_kernel.h) every cycle a wave spends blocked on a DMA
// issue is a cycle it issues no MFMA: staging time and matrix time ADD instead of overlapping (3x3 conv 320 -> 320 @ 64x48: staging stream
// alone 46 us, MFMA stream alone ~55 us, kernel 95-105 us), and tiles below 256x256 -- which need more than the path delivers anyway --
// sit at 17-30 % MFMA busy.
//
// What.  Wave specialisation: NL loader waves do nothing but address arithmetic and LDS-DMA issue (they may stall on the addresser as long
// as they like) and keep NST-1 whole K tiles in flight; WQ x WP consumer waves never touch vector memory inside the K loop: per K tile one
// s_barrier, ds_read_b128 fragments (double-buffered in registers) and MFMAs.  One barrier per K tile does both hand-offs:
//     loader   : wait (counted vmcnt) for MY pieces of tile kt | barrier kt | issue tile kt+NST-1 into the slot tile kt-1 occupied
//     consumer :                                                 barrier kt | read + multiply tile kt
//   RAW: a loader passes barrier kt only after its pieces of tile kt have landed.  WAR: a consumer reaches barrier kt only after every
//   fragment read of tile kt-1 has been waited for (its MFMAs are issued), and the slot of tile kt-1 is refilled after that barrier.
// Same operand orientation, LDS image (lane-linear DMA blocks of 8 rows x 128 B, XOR swizzle on the source chunk and on the read), K order
// (channel chunk outer, tap inner), uniform-soffset addressing and fused epilogue as igemm_kernel.h.  Not supported here (the launcher
// rejects them, the tuner never offers them): the folded nearest-2x upsample and batched launches.
#include "common.h"
#include "kernels.h"
#include "igemm_common.h"

namespace {

template <int WQ, int WP, int TQ, int TP, int NL, int NST>
__global__ __launch_bounds__(64 * (WQ * WP + NL), ((WQ * WP + NL) + 3) / 4) void igemm_lc_kernel(const IGemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only (see igemm_kernel.h)
    constexpr int NC = WQ * WP;                       // consumer waves
    constexpr int BQ = WQ * TQ * 32, BP = WP * TP * 32, BK = 64;
    constexpr int IQ = BQ / 8, IP = BP / 8;          // 1-KiB DMA instructions (8 rows x 128 B) per K tile: weight rows / pixel rows
    static_assert(IQ % NL == 0 && IP % NL == 0, "DMA instructions must split evenly over the loader waves");
    constexpr int LQ = IQ / NL, LP = IP / NL, LI = LQ + LP;   // per loader wave
    static_assert(LI * (NST - 2) <= 63, "vmcnt is a 6-bit counter");
    constexpr int STAGE = (BQ + BP) * BK;            // halves per stage
    constexpr int NKK = BK / 16;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h16* smem = reinterpret_cast<h16*>(smem_raw);

    const int tid = threadIdx.x;
    const int nq = (a.Q + BQ - 1) / BQ, np = (a.P + BP - 1) / BP;
    int qt, pt;
    {
        const int b = blockIdx.x;
        if (a.tile_map == 1) {          // pixel tiles split across the 8 XCDs, q fastest inside an XCD
            const int npx = (np + 7) >> 3, xcd = b & 7, loc = b >> 3;
            pt = xcd * npx + loc / nq; qt = loc % nq;
            if (pt >= np) return;
        } else if (a.tile_map == 2) {   // channel tiles split across the XCDs, p fastest inside an XCD
            const int nqx = (nq + 7) >> 3, xcd = b & 7, loc = b >> 3;
            qt = xcd * nqx + loc / np; pt = loc % np;
            if (qt >= nq) return;
        } else { qt = b % nq; pt = b / nq; }
    }
    const int q0 = qt * BQ, p0 = pt * BP;
    const int z = blockIdx.z;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    int nk = a.K / BK;
    const int ntap = a.ksize * a.ksize;
    int tap0 = 0, cb0 = 0;
    if (a.splitk > 1) {
        const int sps = (nk + a.splitk - 1) / a.splitk;
        const int start = z * sps;
        nk = max(0, min(sps, nk - start));
        cb0 = (start / ntap) * BK; tap0 = start - (start / ntap) * ntap;
    }

    if (wave >= NC) {
        // ================================================================ loader wave
        const int lw = wave - NC;
        const int r8 = lane >> 3, c8 = lane & 7;
        const int HoWo = a.Ho * a.Wo, HsWs = a.Hs * a.Ws;
        const int n_first = p0 / HoWo;
        const int Ct = a.C0 + a.C1;
        const int ldw = a.ldw ? a.ldw : a.K;
        const int back_px = a.pad * a.Ws + a.pad;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<h16*>(a.src0 + (size_t)n_first * HsWs * a.ld0) - (ptrdiff_t)back_px * a.ld0, 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<h16*>(a.src1 ? a.src1 + (size_t)n_first * HsWs * a.ld1 - (ptrdiff_t)back_px * a.ld1 : a.src0), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.W), 0, 0x7FFFFFFF, 0x00020000);
        // my DMA instructions: weight blocks t = lw + NL*i (rows 8t .. 8t+7 of the tile), pixel blocks likewise.  The swizzled source chunk of
        // row 8t + r8 is c8 ^ ((row >> 1) & 7) = c8 ^ (r8 >> 1) ^ 4*(t & 1): folded into the per-row offsets below.
        unsigned wbase[LQ], vox0[LP], vox1[LP], vmask[LP];
#pragma unroll
        for (int i = 0; i < LQ; ++i) {
            const int t = lw + NL * i, row = 8 * t + r8, q = q0 + row;
            const int clog = c8 ^ ((row >> 1) & 7);
            wbase[i] = (q < a.Q) ? (unsigned)(((size_t)q * ldw + clog * 8) * 2) : OOB;
        }
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            const int t = lw + NL * i, row = 8 * t + r8, p = p0 + row;
            const int clog = c8 ^ ((row >> 1) & 7);
            const bool ok = p < a.P;
            const int pp = ok ? p : 0;
            const int n = pp / HoWo, rem = pp - n * HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
            const int y0 = oy * a.stride - a.pad, x0 = ox * a.stride - a.pad;
            const int pix = (n - n_first) * HsWs + y0 * a.Ws + x0 + back_px;
            vox0[i] = (unsigned)((pix * a.ld0 + clog * 8) * 2);
            vox1[i] = (unsigned)((pix * a.ld1 + clog * 8) * 2);
            unsigned m = 0;
            if (ok) {
                int tt = 0;
                for (int dy = 0; dy < a.ksize; ++dy)
                    for (int dx = 0; dx < a.ksize; ++dx, ++tt)
                        if ((unsigned)(y0 + dy) < (unsigned)a.Hs && (unsigned)(x0 + dx) < (unsigned)a.Ws) m |= 1u << tt;
            }
            vmask[i] = m;
        }
        int tap = tap0, cb = cb0;
        int tdy = tap / a.ksize, tdx = tap - (tap / a.ksize) * a.ksize;
        auto issue = [&](int slot) {
            const bool s0 = cb < a.C0;
            const __amdgpu_buffer_rsrc_t rs = s0 ? rs0 : rs1;
            const int ld = s0 ? a.ld0 : a.ld1;
            const int csub = s0 ? cb : cb - a.C0;
            char* sbase = smem_raw + (size_t)slot * (STAGE * 2) + lw * 1024;
            const unsigned so_w = (unsigned)((tap * Ct + cb) * 2);
            const unsigned so_x = (unsigned)(((tdy * a.Ws + tdx) * ld + csub) * 2);
            const unsigned bit = 1u << tap;
            // pixel blocks first (the activation stream misses in L2 more often than the weight panel shared by every workgroup)
#pragma unroll
            for (int i = 0; i < LP; ++i) {
                const unsigned vo = (vmask[i] & bit) ? (s0 ? vox0[i] : vox1[i]) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sbase + (BQ * BK * 2) + i * (NL * 1024)), 16, vo, so_x, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < LQ; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(sbase + i * (NL * 1024)), 16, wbase[i], so_w, 0, 0);
            if (++tdx == a.ksize) { tdx = 0; ++tdy; }
            if (++tap == ntap) { tap = 0; tdy = 0; tdx = 0; cb += BK; }
        };
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nk) issue(s);
        for (int kt = 0; kt < nk; ++kt) {
            const int later = min(NST - 2, nk - 1 - kt);   // K tiles issued after tile kt that may stay in flight
            if (NST >= 6 && later >= 4) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * LI <= 63 ? 4 * LI : 63) : "memory");
            else if (NST >= 5 && later >= 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * LI <= 63 ? 3 * LI : 63) : "memory");
            else if (NST >= 4 && later >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LI <= 63 ? 2 * LI : 63) : "memory");
            else if (NST >= 3 && later >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + NST - 1 < nk) issue((kt + NST - 1) % NST);
        }
        return;   // nothing outstanding: every stage has been waited for; the barrier hardware drops ended waves from the count
    }

    // ==================================================================== consumer wave
    const int wq = wave / WP, wp = wave % WP;
    const int l31 = lane & 31, hh = lane >> 5;
    f32x16 acc[TQ][TP];
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_barrier" ::: "memory");
        const h16* sW = smem + (kt % NST) * STAGE;
        const h16* sX = sW + BQ * BK;
        h16x8 af[2][TQ], bf[2][TP];
        auto load_frags = [&](auto Kc) {
            constexpr int kk = decltype(Kc)::value;
            const int chunk = kk * 2 + hh;
#pragma unroll
            for (int i = 0; i < TQ; ++i) af[kk & 1][i] = *reinterpret_cast<const h16x8*>(sW + swz<BK>((wq * TQ + i) * 32 + l31, chunk));
#pragma unroll
            for (int j = 0; j < TP; ++j) bf[kk & 1][j] = *reinterpret_cast<const h16x8*>(sX + swz<BK>((wp * TP + j) * 32 + l31, chunk));
        };
        load_frags(IntC<0>{});
        static_for<0, NKK>([&](auto Kc) {
            constexpr int kk = decltype(Kc)::value;
            if constexpr (kk + 1 < NKK) load_frags(IntC<kk + 1>{});
#pragma unroll
            for (int i = 0; i < TQ; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
        });
    }
    igemm_epilogue<WQ, WP, TQ, TP>(a, acc, smem, q0, p0, pt, z, wave, lane);
#endif
}

template <int WQ, int WP, int TQ, int TP, int NL, int NST>
int launch_lc(IGemmArgs a, int batch, hipStream_t st) {
    constexpr int BQ = WQ * TQ * 32, BP = WP * TP * 32;
    constexpr int RING = NST * (BQ + BP) * 64 * (int)sizeof(h16), EPI = igemm_epilogue_lds_bytes<WQ, WP, TQ>();
    constexpr int SMEM = RING > EPI ? RING : EPI;
    static_assert(SMEM <= 160 * 1024, "LDS budget of one CU");
    if (a.ups || (batch != 1 && a.splitk <= 1)) return -16;
    static unsigned long long attr_done = 0;
    auto kfn = igemm_lc_kernel<WQ, WP, TQ, TP, NL, NST>;
    if (ladi_ensure_dyn_lds(reinterpret_cast<const void*>(kfn), SMEM, attr_done)) return -10;
    const int nq = (a.Q + BQ - 1) / BQ, np = (a.P + BP - 1) / BP;
    int blocks = nq * np;
    a.tile_map = 0;
    if (np >= 16) { a.tile_map = 1; blocks = 8 * ((np + 7) / 8) * nq; }
    else if (nq >= 16) { a.tile_map = 2; blocks = 8 * ((nq + 7) / 8) * np; }
    dim3 grid((unsigned)blocks, 1, (unsigned)batch);
    hipLaunchKernelGGL(kfn, grid, dim3(64 * (WQ * WP + NL)), SMEM, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

}  // namespace

// (tq, tp) = consumer wave tile in 32-blocks on a 2 x 2 consumer grid; nst = ring depth
int ladi_launch_igemm_lc(const IGemmArgs& a, int tq, int tp, int nst, int batch, hipStream_t st) {
    if (tq == 2 && tp == 2 && nst == 4) return launch_lc<2, 2, 2, 2, 2, 4>(a, batch, st);   // 128x128, 128 KB ring
    if (tq == 2 && tp == 2 && nst == 5) return launch_lc<2, 2, 2, 2, 2, 5>(a, batch, st);   // 128x128, 160 KB ring
    if (tq == 4 && tp == 2 && nst == 3) return launch_lc<2, 2, 4, 2, 2, 3>(a, batch, st);   // 256x128, 144 KB ring
    if (tq == 2 && tp == 4 && nst == 3) return launch_lc<2, 2, 2, 4, 2, 3>(a, batch, st);   // 128x256, 144 KB ring
    if (tq == 2 && tp == 1 && nst == 6) return launch_lc<2, 2, 2, 1, 2, 6>(a, batch, st);   // 128x64, 144 KB ring
    if (tq == 5 && tp == 2 && nst == 2) return launch_lc<2, 2, 5, 2, 2, 2>(a, batch, st);   // 320x128, 112 KB ring
    if (tq == 3 && tp == 3 && nst == 3) return launch_lc<2, 2, 3, 3, 2, 3>(a, batch, st);   // 192x192, 144 KB ring
    return -7;
}
