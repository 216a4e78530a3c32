// Halo-resident 3x3 convolution (round 3), reached through ladi_launch_igemm (cfg 74..): kernel template and launcher template.  Included by the
// instantiation units igemm_halo_inst_*.hip (two forms each, so the build parallelises: one form takes ~30 s of hipcc) and by the harnesses under tools/.
//
// Why.  What bounds every implicit-GEMM tile below 256x256 on this chip is the global -> LDS staging rate (25-32 B/clk/CU in these
// kernels; tools/dma_conv_pattern.hip), and the ring kernels stage the PIXEL operand once per filter tap: nine
// shifted copies of the same activation rows per channel chunk.  In row-major pixel order a tap is a LINEAR shift: output pixel p reads
// input pixel p + (dy-1)*W + (dx-1).  So for a tile of BP consecutive output pixels the nine taps live in ONE contiguous range of
// BP + 2W + 2 input pixels.  This kernel stages that range once per 64-channel chunk (the "halo tile"), keeps it in LDS for all nine taps
// and streams only the WEIGHT tile per tap; the tap is applied when the B fragments are READ (row = pixel + tap shift), and the lanes
// whose tap falls outside the image (left / right edge wrap-around, top / bottom rows, the neighbouring sample) read a row of zeros
// instead.  Staged bytes per channel chunk drop from 9 (BQ + BP) x 128 B to (9 BQ + BP + 2W + 2) x 128 B -- 0.42x for a 128x256 tile at
// W = 24 -- which moves the 32x24 / 64x48-level convolutions from staging-bound to MFMA-bound.
//
// Scope: 3x3, stride 1, pad 1, no folded upsample, W <= 48 (every level of the UNet at 512x384; wider images would need a 2-D blocked halo
// tile), channel counts multiples of 64, two-source concat supported (the chunk selects the source).  8 waves (2 x 4) per workgroup, one
// workgroup per CU, weights double-buffered; the halo tile double-buffered (NXB = 2: the next chunk's tile arrives spread over the taps)
// or single (NXB = 1, for the 320-row weight tile: one exposed tile load per nine taps).  Same fragment layout, swizzle, epilogue and
// split-K convention as igemm_kernel.h.
#pragma once
#include "common.h"
#include "kernels.h"
#include "igemm_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int HALO_WMAX = 48;
// Ablation switches for tools/r05/halo_ablate.hip (compiled with -DLADI_HALO_ABL=<mask>; the library never defines it, so every
// `if constexpr` below folds to the full kernel): 1 = no weight DMA in the loop, 2 = no halo-tile DMA in the loop, 4 = no fragment ds_reads
// (MFMAs on loop-invariant registers), 8 = no MFMAs (fragments kept alive), 16 = no per-step wait + barrier
#ifndef LADI_HALO_ABL
#define LADI_HALO_ABL 0
#endif
constexpr int ABL = LADI_HALO_ABL;
// Round 6: the fragment double buffer below is only real if the order is pinned.  Left alone, hipcc sinks the reads of k-group kk + 1 behind the
// MFMAs of kk and keeps ONE fragment register set -- [4 ds_read_b128, s_waitcnt lgkmcnt(0), 4 MFMA] x 4 per step, an exposed LDS round trip
// per k-group (ISA excerpt: tools/r06/README.md; this is also why round 5's "prefetch all sixteen reads" A/B compiled to the same 182 registers
// and measured nothing).  A sched_barrier on both sides of each k-group's MFMAs keeps the source order: 197 registers, conv 640 -> 640 @ 32x24
// 97.0 -> 91.9 us on one box (profiles/r06_halo_sched.txt).  Only the two-workgroups-per-CU forms with a 64 x 64 wave tile gain (the other forms:
// +-1 %; the twelve-wave 320 x 192 form would spill at its 168-register budget), so only they are pinned.  -DLADI_HALO_PIN=0 / 1 overrides (A/B).
#ifndef LADI_HALO_PIN
#define LADI_HALO_PIN -1
#endif

// halo-tile passes issued at tap `tt` of a chunk that has a successor (the LX passes of the next chunk's tile are spread over taps 0..7)
template <int LX>
constexpr int nx_at(int tt) { return tt < 8 ? (LX * (tt + 1)) / 8 - (LX * tt) / 8 : 0; }
// pieces of halo tile issued in the D steps before tap t (taps < 0 belong to the previous chunk, which always has a successor; the
// current chunk's own taps count only if it has one: `pf`)
template <int LX, int NXB>
constexpr int nx_sum(int t, int D, bool pf) {
    if (NXB != 2) return 0;
    int n = 0;
    for (int k = 1; k <= D; ++k) {
        const int tt = t - k;
        if (tt >= 0) n += pf ? nx_at<LX>(tt) : 0;
        else n += nx_at<LX>(9 + tt);
    }
    return n;
}

// WMAX: widest image row the halo buffer is sized for (48: every level of the UNet at 512x384; 24: the 32x24 level and below, whose smaller
// buffer leaves room for a third weight slot at two workgroups per CU).  WPN = 6: twelve waves (2 x 6), three per SIMD -- the 320x192 tile
// that covers the 64x48 level (49 152 pixels x 320 channels at batch 8) with exactly 256 workgroups.
// ONE = 1 (round 5): ONE workgroup of four waves per CU, one wave per SIMD with the whole 512-register file -- 320x192 as 2 x 2 waves of
// 160 x 96 (240 accumulator registers, fragments double-buffered on top): per MFMA the wave reads 8 / 15 KB of fragments where the twelve-wave
// form of the same tile reads 6 / 5 KB, and issues 20 DMA pieces per 60 MFMAs.
// G2D = 1 (round 5): 2-D BLOCKED halo tile for images whose rows are wider than WMAX pixels (every VAE / EMASC level above 64x48, the 128x96
// latent grid of 1024x768).  The pixel tile is TH = BP / 32 image rows x 32 columns; the staged block is (TH + 2) x 34 pixels stored with a row
// pitch of 34, so a tap is again a LINEAR shift of the row index -- (dy - 1) * 34 + (dx - 1) -- and, because the left / right neighbours are
// real halo columns (zero-filled by the descriptor's bounds check outside the image), the consumer needs no validity masks at all.  Staged
// rows per chunk: (TH + 2) * 34 for TH * 32 pixels (1.33x at TH = 8) where the ring kernels stage 9x.  Requires W % 32 == 0, H % TH == 0
// (whole blocks), a single halo buffer (NXB = 1); the epilogue sees sub-tiles one image row apart (igemm_epilogue pstr = W).
// UPS = 1 (round 6): the nearest-2x upsample of Upsample2D FOLDED into the 3x3 convolution that follows it (diffusers Upsample2D: interpolate x 2 ->
// conv; runtime ConvOpt::ups).  The source is the LOW-resolution image (Hs x Ws), the output 2 Hs x 2 Ws; tap (dy, dx) of output pixel (y, x) reads
// source pixel ((y + dy - 1) >> 1, (x + dx - 1) >> 1).  The staged "halo tile" is the run of low-resolution image rows the tile's output rows touch
// (a QUARTER of the pixels a plain halo tile stages), and the tap is applied at read time through two per-lane 3-entry tables (source row, source
// column) instead of the linear shift.  The ring kernels did this layer with per-lane addressing of every tap's gather (12 VALU per DMA piece):
// 948-955 TFLOP/s where the plain convolutions of the same size reach 1 180 (profiles/r05_unet_forward_kernel_stats.txt: the 382 / 379 us launches).
// Requires whole tiles inside a sample (Ho Wo % BP == 0), a single source, an output row of at most WMAX pixels, one halo buffer.
template <int TQ, int TP, int NXB, int NSTW, int WPN, int WMAX = 48, int ONE = 0, int G2D = 0, int UPS = 0>
__global__ __launch_bounds__(128 * WPN, (ONE ? 1 : (WPN == 6 ? 3 : 2))) void igemm_halo_kernel(const IGemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only (see igemm_kernel.h)
    constexpr int WQ = 2, WP = WPN, BK = 64, NT = 128 * WPN;    // 8 waves (2 x 4), or 4 waves (2 x 2) with two workgroups per CU
    constexpr int BQ = WQ * TQ * 32, BP = WP * TP * 32;
    constexpr int RPP = NT / 8;                                  // 64 tile rows per DMA pass of the workgroup
    constexpr int RQ = (BQ + RPP - 1) / RPP;                     // weight passes per tap
    constexpr int TH = BP / 32, HC = 34;                         // G2D: image rows per block, row pitch of the staged block (32 + 2 halo columns)
    static_assert(!G2D || NXB == 1, "the 2-D blocked form keeps one halo buffer");
    static_assert(!UPS || (NXB == 1 && !G2D && !ONE), "the folded-upsample form: one halo buffer, linear tile");
    constexpr int XROWS = ((G2D ? (TH + 2) * HC : BP + 2 * WMAX + 2) + RPP - 1) / RPP * RPP;   // rows of one halo-tile buffer (whole passes)
    constexpr int LX = XROWS / RPP;                              // halo passes per channel chunk
    constexpr int WSLOT = RQ * RPP * BK;                         // halves per weight slot (padded to whole passes)
    constexpr int XBUF = XROWS * BK;                             // halves per halo buffer
    constexpr int D = NSTW - 1;                                  // weight tiles issued ahead of the one being multiplied
    static_assert(NSTW >= 2 && NSTW <= 4, "weight ring depth");
    // double halo buffer: the counted wait at taps 0..D-2 of a chunk lets the LAST halo passes of that chunk's own tile (issued at taps
    // 9-D..7 of the previous chunk) stay in flight; that is safe only while those passes cover rows the first taps do not read -- the
    // first D-1 taps read rows < BP + 2, the late passes start at row floor(LX (9 - D) / 8) * RPP (made explicit)
    static_assert(NXB != 2 || ((LX * (9 - D)) / 8) * RPP >= BP + 2, "late halo passes would overlap the rows the first taps read");
    constexpr int ZOFF = NSTW * WSLOT + NXB * XBUF;              // the row of zeros (halves)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h16* smem = reinterpret_cast<h16*>(smem_raw);

    const int tid = threadIdx.x;
    const int nq = (a.Q + BQ - 1) / BQ, np = (a.P + BP - 1) / BP;
    int qt, pt, z = blockIdx.z;
    {
        const int b = blockIdx.x;
        if ((a.tile_map & 15) == 3) {
            // Round 6 -- WEIGHT-SLICE-major map for the deep, few-pixel levels (8x6 / 16x12: the weights are 4-15x the pixel operand).  A unit =
            // (channel tile, K slice, group of G pixel tiles); unit u runs on XCD u % 8 and its G workgroups are consecutive there, so they
            // start together and the XCD's L2 serves G - 1 of the G reads of every weight tile.  With the plain maps the pixel tiles of one
            // weight slice land on 4-8 different XCDs and every L2 fetches the slice again: 197 MB of fabric traffic per launch for 34 MB of
            // operands on the 8x6 convolutions (profiles/r05_pmc_fetch_size.txt), at 4.3 TB/s the limiter of that kernel.  grid.z = 1.
            const int G = a.tile_map >> 4, S = a.splitk > 1 ? a.splitk : 1, npg = (np + G - 1) / G;
            const int xcd = b & 7, loc = b >> 3, ui = loc / G, pi = loc - ui * G;
            const int u = ui * 8 + xcd;                      // unit index = (qt * S + z) * npg + pg
            if (u >= nq * S * npg) return;
            const int pg = u % npg, qz = u / npg;
            pt = pg * G + pi; qt = qz / S; z = qz - qt * S;
            if (pt >= np) return;
        } else if (a.tile_map == 1) {
            const int npx = (np + 7) >> 3, xcd = b & 7, loc = b >> 3;
            pt = xcd * npx + loc / nq; qt = loc % nq;
            if (pt >= np) return;
        } else if (a.tile_map == 2) {
            const int nqx = (nq + 7) >> 3, xcd = b & 7, loc = b >> 3;
            qt = xcd * nqx + loc / np; pt = loc % np;
            if (qt >= nq) return;
        } else { qt = b % nq; pt = b / nq; }
    }
    const int q0 = qt * BQ;
    // G2D: block pt = (sample, block row, block column); p0 = pixel index of the block's first pixel, sub-tile r is image row y0 + r
    int g_n = 0, g_y0 = 0, g_x0 = 0;
    if constexpr (G2D) {
        const int txn = a.Ws >> 5, tps = (a.Hs / TH) * txn;
        g_n = pt / tps;
        const int t = pt - g_n * tps, ty = t / txn;
        g_y0 = ty * TH; g_x0 = (t - ty * txn) * 32;
    }
    const int p0 = G2D ? (g_n * a.Hs + g_y0) * a.Ws + g_x0 : pt * BP;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wq = wave / WP, wp = wave % WP;
    const int l31 = lane & 31, hh = lane >> 5;
    const int r0 = tid >> 3;                                     // row inside a DMA pass (RPP rows)
    const int c8 = tid & 7;

    const int Ws = a.Ws, Hs = a.Hs, HW = Hs * Ws;
    const int Ct = a.C0 + a.C1;
    const int ldw = a.ldw ? a.ldw : a.K;
    // descriptors with the exact extent of each operand: rows before the first / after the last pixel of the tensor are zero-filled
    // (the launcher guarantees P * ld * 2 < 2^31)
    const int Psrc = UPS ? a.P / 4 : a.P;                        // pixels of the source tensor
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.src0), 0, (unsigned)((size_t)Psrc * a.ld0 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.src1 ? a.src1 : a.src0), 0,
                                                                         (unsigned)((size_t)Psrc * (a.src1 ? a.ld1 : a.ld0) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.W), 0, 0x7FFFFFFF, 0x00020000);

    // ---- DMA-side per-lane offsets (constant over the K loop)
    unsigned wbase[RQ];
#pragma unroll
    for (int i = 0; i < RQ; ++i) {
        const int row = r0 + RPP * i, q = q0 + row;
        const int clog = c8 ^ ((row >> 1) & 7);
        wbase[i] = (row < BQ && q < a.Q) ? (unsigned)(((size_t)q * ldw + clog * 8) * 2) : OOB;
    }
    // UPS: the tile lies inside sample u_n (launcher: Ho Wo % BP == 0); its output rows u_y0 .. u_y1 touch the source rows (u_y0 - 1) >> 1 ..
    // (u_y1 + 1) >> 1; halo-tile row r = source pixel u_l0 + r of that sample (u_l0 < 0 on the first tile: rows in front of the image are zero-filled)
    int u_n = 0, u_q0 = 0, u_r0 = 0, u_l0 = 0, u_rows = 0;
    if constexpr (UPS) {
        const int Wo = 2 * Ws, HWo = 4 * HW;
        u_n = p0 / HWo; u_q0 = p0 - u_n * HWo;
        const int y0 = u_q0 / Wo, y1 = (u_q0 + BP - 1) / Wo;
        u_r0 = (y0 - 1) >> 1;
        u_l0 = u_r0 * Ws;
        u_rows = (((y1 + 1) >> 1) - u_r0 + 1) * Ws;               // staged rows actually read (the passes beyond them are skipped)
    }
    unsigned xo0[LX], xo1[LX];
#pragma unroll
    for (int i = 0; i < LX; ++i) {
        const int row = r0 + RPP * i;                            // halo-tile row = input pixel p0 - Ws - 1 + row
        long long pin = (long long)p0 - Ws - 1 + row;
        const int clog = c8 ^ ((row >> 1) & 7);
        bool ok = pin >= 0 && pin < a.P;
        if constexpr (UPS) {
            const int lin = u_l0 + row;
            ok = lin >= 0 && lin < HW && row < u_rows;
            pin = (long long)u_n * HW + lin;
        }
        if constexpr (G2D) {                                     // row = (block row rr, block column cc) of the (TH + 2) x 34 block
            const int rr = row / HC, cc = row - rr * HC;
            const int y = g_y0 - 1 + rr, x = g_x0 - 1 + cc;
            ok = rr < TH + 2 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
            pin = ((long long)g_n * Hs + y) * Ws + x;
        }
        xo0[i] = ok ? (unsigned)((pin * a.ld0 + clog * 8) * 2) : OOB;
        xo1[i] = ok ? (unsigned)((pin * a.ld1 + clog * 8) * 2) : OOB;
    }
    // ---- consumer-side: this lane's pixel in each of its TP blocks: halo-tile row of the centre tap and the 9-bit validity mask
    int rb[TP]; unsigned vm[TP];
    int uyr[UPS ? TP : 1][3], uxr[UPS ? TP : 1][3];             // UPS: halo-tile row offset of source row (y + d - 1) >> 1, source column (x + d - 1) >> 1
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int pl = (wp * TP + j) * 32 + l31, p = p0 + pl;
        rb[j] = G2D ? (wp * TP + j + 1) * HC + l31 + 1 : pl + Ws + 1;
        unsigned m = G2D ? 0x1ffu : 0u;                          // G2D: whole blocks inside the image, out-of-image taps are zero-filled halo pixels
        if constexpr (UPS) {
            const int Wo = 2 * Ws, Ho = 2 * Hs;
            const int q = u_q0 + pl, oy = q / Wo, ox = q - oy * Wo;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                uyr[j][d] = (((oy + d - 1) >> 1) - u_r0) * Ws;
                uxr[j][d] = (ox + d - 1) >> 1;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if ((unsigned)(oy + t / 3 - 1) < (unsigned)Ho && (unsigned)(ox + t % 3 - 1) < (unsigned)Wo) m |= 1u << t;
        } else
        if (!G2D && p < a.P) {
            const int rem = p % HW, oy = rem / Ws, ox = rem - oy * Ws;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if ((unsigned)(oy + t / 3 - 1) < (unsigned)Hs && (unsigned)(ox + t % 3 - 1) < (unsigned)Ws) m |= 1u << t;
        }
        vm[j] = m;
    }

    // split-K over whole (chunk, tap) steps
    int nk = a.K / BK, s_begin = 0;
    if (a.splitk > 1) {
        const int sps = (nk + a.splitk - 1) / a.splitk;
        s_begin = z * sps;
        nk = max(0, min(sps, nk - s_begin));
    }
    const int s_end = s_begin + nk;

    auto issue_w = [&](int s) {        // weight tile of step s = (chunk s / 9, tap s % 9) into ring slot s % NSTW
        const int cb = (s / 9) * BK, tap = s - (s / 9) * 9;
        const unsigned so = (unsigned)((tap * Ct + cb) * 2);
        char* base = smem_raw + (size_t)(s % NSTW) * (WSLOT * 2) + wave * 1024;
#pragma unroll
        for (int i = 0; i < RQ; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(base + i * (RPP * BK * 2)), 16, wbase[i], so, 0, 0);
    };
    auto issue_x = [&](int chunk, int i0, int i1) {   // passes [i0, i1) of the halo tile of `chunk` into buffer chunk % NXB
        const int cb = chunk * BK;
        const bool s0 = cb < a.C0;
        const __amdgpu_buffer_rsrc_t rs = s0 ? rs0 : rs1;
        const unsigned so = (unsigned)((s0 ? cb : cb - a.C0) * 2);
        char* base = smem_raw + (size_t)(NSTW * WSLOT + (NXB == 2 ? (chunk & 1) : 0) * XBUF) * 2 + wave * 1024;
#pragma unroll
        for (int i = 0; i < LX; ++i)
            if (i >= i0 && i < i1 && (!UPS || i * RPP < u_rows))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + i * (RPP * BK * 2)), 16, s0 ? xo0[i] : xo1[i], so, 0, 0);
    };

    f32x16 acc[TQ][TP];
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (tid < 8) *reinterpret_cast<h16x8*>(smem + ZOFF + tid * 8) = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
    const int c_first = s_begin / 9, c_last = (s_end - 1) / 9;
    if (nk > 0) {
        issue_x(c_first, 0, LX);
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (s_begin + d < s_end) issue_w(s_begin + d);
    }

    for (int c = c_first; c <= c_last && nk > 0; ++c) {
        const h16* sX = smem + NSTW * WSLOT + (NXB == 2 ? (c & 1) : 0) * XBUF;
        const bool prefetch_x = (NXB == 2) && (c < c_last);
        static_for<0, 9>([&](auto Tc) {
            constexpr int t = decltype(Tc)::value;
            const int s = c * 9 + t;
            if (s >= s_begin && s < s_end) {
                // Counted wait for the weight tile of THIS step.  DMAs complete in issue order per wave; behind W(s) (issued D steps ago) the
                // wave has issued, per step since then, the pieces of the next chunk's halo tile (nx(tap) of them while the chunk has a
                // successor) and one weight tile (RQ pieces).  Steady state only -- near the ends of the slice the count is smaller than
                // the formula, and a count that is too LARGE would not wait long enough: there the wait is vmcnt(0).
                constexpr int NPF = RQ * (D - 1) + nx_sum<LX, NXB>(t, D, true), NNOPF = RQ * (D - 1) + nx_sum<LX, NXB>(t, D, false);
                static_assert(NPF <= 63, "vmcnt is a 6-bit counter");
                // (single halo buffer: the tile of this chunk was issued BEHIND the weight tiles at the end of the previous chunk, so the first
                // tap drains everything)
                const bool steady = (s - D >= s_begin) && (s + D - 1 < s_end) && !(NXB == 1 && t == 0);
                if constexpr (!(ABL & 16)) {
                    if (steady && prefetch_x) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPF) : "memory");
                    else if (steady) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NNOPF) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
                if constexpr (!(ABL & 1)) { if (s + D < s_end) issue_w(s + D); }
                if constexpr (NXB == 2 && t < 8) {
                    // a split-K slice may enter the chunk at tap t > 0: its first step also issues the passes of the taps it skipped
                    // (those steps are not "steady": they wait with vmcnt(0))
                    if constexpr (!(ABL & 2)) { if (prefetch_x) issue_x(c + 1, s == s_begin ? 0 : (LX * t) / 8, (LX * (t + 1)) / 8); }
                }
                const h16* sW = smem + (s % NSTW) * WSLOT;
                const int tshift = (t / 3 - 1) * (G2D ? HC : Ws) + (t % 3 - 1);
                int xoff[TP];          // half offset of the lane's row in the halo tile (or the zero row), swizzle term separate
                int xsw[TP];
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const bool valid = (vm[j] >> t) & 1u;
                    int row = rb[j] + tshift;
                    if constexpr (UPS) row = uyr[j][t / 3] + uxr[j][t % 3];
                    xoff[j] = valid ? (int)(sX - smem) + row * 64 : ZOFF;
                    xsw[j] = valid ? ((row >> 1) & 7) : 0;
                }
                // fragments double-buffered in registers when the accumulators leave room (the 320x256 tile holds 160 accumulator registers:
                // single buffer there, its partner wave on the SIMD covers the LDS latency)
                constexpr int DB = (TQ * TP * 16 + 2 * (TQ + TP) * 4 <= (ONE ? 400 : 200)) ? 1 : 0;
                h16x8 af[1 + DB][TQ], bf[1 + DB][TP];
                auto load_frags = [&](auto Kc) {
                    constexpr int kk = decltype(Kc)::value;
                    const int chunk = kk * 2 + hh;
                    if constexpr (ABL & 4) {                   // ablation: no LDS reads, the MFMAs run on opaque register contents
#pragma unroll
                        for (int i = 0; i < TQ; ++i) asm volatile("" : "=v"(af[kk & DB][i]));
#pragma unroll
                        for (int j = 0; j < TP; ++j) asm volatile("" : "=v"(bf[kk & DB][j]));
                        return;
                    }
#pragma unroll
                    for (int i = 0; i < TQ; ++i) af[kk & DB][i] = *reinterpret_cast<const h16x8*>(sW + swz<BK>((wq * TQ + i) * 32 + l31, chunk));
#pragma unroll
                    for (int j = 0; j < TP; ++j) bf[kk & DB][j] = *reinterpret_cast<const h16x8*>(smem + xoff[j] + ((chunk ^ xsw[j]) << 3));
                };
                constexpr bool PIN = LADI_HALO_PIN >= 0 ? (LADI_HALO_PIN != 0 && DB) : (DB && WPN == 2 && TQ * TP <= 4 && !ONE);
                if constexpr (DB) load_frags(IntC<0>{});
                static_for<0, 4>([&](auto Kc) {
                    constexpr int kk = decltype(Kc)::value;
                    if constexpr (DB) { if constexpr (kk + 1 < 4) load_frags(IntC<kk + 1>{}); }
                    else load_frags(IntC<kk>{});
                    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TQ; ++i)
#pragma unroll
                        for (int j = 0; j < TP; ++j)
                        {
                            const h16x8 fa = af[kk & DB][i], fb = bf[kk & DB][j];
                            if constexpr (ABL & 8) asm volatile("" ::"v"(fa), "v"(fb));
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[i][j], 0, 0, 0);
                        }
                    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
        if constexpr (NXB == 1) {
            if (c < c_last) {        // single halo buffer: every wave must be done with it before the next chunk's tile overwrites it
                if constexpr (!(ABL & 16)) asm volatile("s_barrier" ::: "memory");
                if constexpr (!(ABL & 2)) issue_x(c + 1, 0, LX);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    igemm_epilogue<WQ, WP, TQ, TP>(a, acc, smem, q0, p0, pt, z, wave, lane, G2D ? Ws : 32);
#endif
}

template <int TQ, int TP, int NXB, int NSTW, int WPN, int WMAX = 48, int ONE = 0, int G2D = 0, int UPS = 0>
int launch_halo(IGemmArgs a, int batch, hipStream_t st) {
    constexpr int BQ = 64 * TQ, BP = 32 * WPN * TP, RPP = 16 * WPN;
    constexpr int RQ = (BQ + RPP - 1) / RPP, XROWS = ((G2D ? (BP / 32 + 2) * 34 : BP + 2 * WMAX + 2) + RPP - 1) / RPP * RPP;
    constexpr int SMEM = (NSTW * RQ * RPP * 64 + NXB * XROWS * 64) * (int)sizeof(h16) + 128;
    static_assert(SMEM <= 160 * 1024, "LDS budget of one CU");
    static_assert(SMEM >= igemm_epilogue_lds_bytes<2, WPN, TQ>(), "epilogue patches must fit");
    if (UPS) {
        if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || !a.ups || 2 * a.Ws > WMAX || a.Ho != 2 * a.Hs || a.Wo != 2 * a.Ws || a.C1 || a.src1) return -16;
        if (((a.Ho * a.Wo) % BP) || (a.P % (a.Ho * a.Wo))) return -16;                       // whole tiles inside a sample
        if ((((BP + 2 * a.Ws - 1) / (2 * a.Ws) + 2) / 2 + 2) * a.Ws > XROWS) return -16;     // source rows a tile can touch (generous bound)
    } else
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.ups || (!G2D && a.Ws > WMAX) || a.Ho != a.Hs || a.Wo != a.Ws) return -16;
    if ((a.C0 % 64) || (a.C1 % 64) || (batch != 1 && a.splitk <= 1)) return -16;
    if (G2D && ((a.Ws % 32) || (a.Hs % (BP / 32)) || batch != 1 || a.splitk > 1)) return -16;   // whole (BP / 32) x 32 blocks, no split-K
    if (NXB == 2 && a.splitk > 1) {
        // a K slice that ENTERS a chunk at its last tap would never prefetch the next chunk's halo tile (the passes ride on taps 0..7):
        // such a split is refused here rather than mis-computed (unreachable with the shipped split factors 2 and 4)
        const int nk = a.K / 64, sps = (nk + a.splitk - 1) / a.splitk;
        for (int z = 1; z < a.splitk; ++z)
            if ((z * sps) % 9 == 8 && z * sps < nk) return -16;
    }
    if ((size_t)a.P * (size_t)std::max(a.ld0, a.ld1) * 2 >= 0x7FFFFFFFull) return -16;   // 32-bit byte offsets from the tensor base
    static unsigned long long attr_done = 0;
    auto kfn = igemm_halo_kernel<TQ, TP, NXB, NSTW, WPN, WMAX, ONE, G2D, UPS>;
    if (ladi_ensure_dyn_lds(reinterpret_cast<const void*>(kfn), SMEM, attr_done)) return -10;
    const int nq = (a.Q + BQ - 1) / BQ, np = (a.P + BP - 1) / BP;
    int blocks = nq * np;
    a.tile_map = 0;
    if (np >= 16) { a.tile_map = 1; blocks = 8 * ((np + 7) / 8) * nq; }
    else if (nq >= 16) { a.tile_map = 2; blocks = 8 * ((nq + 7) / 8) * np; }
    // weight-slice-major map (see the kernel): when the weights outweigh the pixel operand at least 3:1.  G = half of the pixel tiles of a
    // slice (the 16x12 level: 24 tiles -> 40 units of 12, 60 workgroups per XCD).  Measured (profiles/r06_halo_map3.txt): fabric traffic of the
    // 16x12 convolution 1280 -> 1280 283 + 38 MB -> 112 + 38 MB per launch, time unchanged (2560 -> 1280: -4 %); on the 8x6 level (G = all 6
    // tiles) 148 + 17 MB -> 50 + 17 MB and the launch gets 5 % SLOWER (six workgroups of an XCD asking for the same weight lines at the same
    // moment) -- that kernel was never bandwidth-bound (VERDICT r05 read its 4.3 TB/s as the limiter), so the map is offered from 13 pixel
    // tiles up only; LADI_HALO_MAP3=0 keeps the round-5 maps everywhere, =2 forces it for every eligible launch (A/B)
    int gz = batch;
    {
        static const int mode = [] { const char* e = getenv("LADI_HALO_MAP3"); return e ? atoi(e) : 1; }();
        const bool off = mode == 0 || (mode != 2 && np <= 12);
        const size_t wbytes = (size_t)a.Q * a.K * 2, xbytes = (size_t)(UPS ? a.P / 4 : a.P) * (a.C0 + a.C1) * 2;
        const int S = a.splitk > 1 ? a.splitk : 1;
        if (!off && !G2D && (batch == 1 || a.splitk > 1) && wbytes >= 3 * xbytes && np <= 32 && np >= 2) {
            const int G = np <= 12 ? np : (np + 1) / 2;
            const int units = nq * S * ((np + G - 1) / G);
            if (G <= 255 && units >= 8) { a.tile_map = 3 | (G << 4); blocks = 8 * ((units + 7) / 8) * G; gz = 1; }
        }
    }
    dim3 grid((unsigned)blocks, 1, (unsigned)gz);
    hipLaunchKernelGGL(kfn, grid, dim3(128 * WPN), SMEM, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

}  // namespace

// one line per form in an instantiation unit: the external entry point igemm_halo.hip dispatches to
#define LADI_HALO_INSTANTIATE(NAME, ...) \
    int ladi_halo_launch_##NAME(IGemmArgs a, int batch, hipStream_t st) { return launch_halo<__VA_ARGS__>(a, batch, st); }
