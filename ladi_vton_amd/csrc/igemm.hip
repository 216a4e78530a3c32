// Implicit-GEMM convolution / linear / batched GEMM on gfx950 MFMA (v_mfma_f32_32x32x16_f16).
//
// One kernel family covers every dense contraction of the LaDI-VTON hot path:
//   conv3x3 s1/s2 (with optional folded nearest-2x upsample and two-source channel concat),
//   conv1x1, nn.Linear, and the batched products of the VAE mid-block attention.
//
// Orientation: D[q][p] = sum_k W[q][k] * X[p][k]   (q = output channel, p = output pixel / token)
//   A operand (MFMA rows)  = weight tile  [BQ][BK]  in LDS
//   B operand (MFMA cols)  = gathered activation tile [BP][BK] in LDS (im2col is done by the per-lane gather address)
//   -> every lane ends up owning ONE pixel (col = lane&31) and groups of 4 CONSECUTIVE output channels
//      (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so NHWC epilogue loads/stores are 8-byte vectors and
//      per-pixel quantities (mask) are lane-local.
//
// Data movement (the part that bounds this kernel: LDS write bandwidth is ~1/3 of LDS read bandwidth on CDNA4):
//   * global -> LDS goes through the LDS-DMA path (`buffer_load_dwordx4 ... offen lds`): no VGPR round trip and no
//     ds_write instructions.  Zero padding of the convolution halo, ragged tile rows and rows >= Q come for free from
//     the buffer descriptor's bounds check (an out-of-range voffset returns 0 into LDS).
//   * the LDS image of a DMA is lane-linear (M0 base + lane*16 B), so the bank-conflict XOR swizzle is applied to the
//     per-lane SOURCE address and again on the ds_read_b128 side (cdna_hip_programming.md §5.4 rule 21).
//   * NST-stage LDS ring, one raw s_barrier per K step, counted vmcnt so that up to NST-1 stages stay in flight.
//   * K-loop order: channel chunk outer, tap inner, so the taps of a 3x3 window re-use input rows from L2.
//   * workgroup -> tile mapping is XCD-aware (block b runs on XCD b%8): each XCD walks a contiguous range of pixel
//     (or channel) tiles so co-resident workgroups share operand panels in that XCD's private L2.
#include "common.h"
#include "kernels.h"
#include "igemm_common.h"
#include <vector>
#include <unordered_map>
#include <map>
#include <mutex>
#include <array>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <utility>
#include <dlfcn.h>

int ladi_igemm_num_cfgs();
#include "igemm_tiles.h"
// one launcher per ring-kernel tile shape, defined in igemm_inst_*.hip (igemm_kernel.h)
#define X(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV) int ladi_igemm_launch_base_##base(IGemmArgs a, int batch, hipStream_t st);
LADI_IGEMM_TILES_ALL(X)
#undef X

namespace {

// ------------------------------------------------------------------------------------------------
// split-K second pass: out = epilogue(sum_z part[z]) for problems with few output tiles and a deep K loop (the 8x6 level of the
// UNet at batch 8 has only ~120 tiles for 256 CUs).  One block per 32 output pixels, threads own channels, pixels walked
// sequentially -> coalesced rows and the same deterministic per-channel partial statistics rows as the fused epilogue.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, const IGemmArgs a) {
    __shared__ float red[2][4][64];
    const int p_base = blockIdx.x * 32;
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;   // 64 channels x 4 pixel lanes per block
    const int c = blockIdx.y * 64 + cl;
    const size_t slice = (size_t)a.P * a.Q;
    const float* rowadd = a.rowadd;
    if (rowadd && a.rowadd_idx) rowadd += (size_t)(*a.rowadd_idx) * a.rowadd_stride;
    float ssum = 0.f, ssq = 0.f;
    if (c < a.Q) {
        const float b = (a.bias ? (float)a.bias[c] * (a.bias_mul != 0.f ? a.bias_mul : 1.f) : 0.f) + (rowadd ? rowadd[c] : 0.f);
        for (int r = pl; r < 32; r += 4) {
            const int p = p_base + r;
            if (p >= a.P) break;
            float x = b;
            for (int z = 0; z < S; ++z) x += part[(size_t)z * slice + (size_t)p * a.Q + c];
            if (a.act == LADI_ACT_SILU) x = silu_f(x);
            else if (a.act == LADI_ACT_GELU) x = gelu_f(x);
            else if (a.act == LADI_ACT_RELU) x = fmaxf(x, 0.f);
            x = (float)(h16)(x * a.out_scale);   // same rounding point as the fused epilogue (fp16 before the residual add)
            if (a.res0) x += (float)a.res0[(size_t)p * a.ldr0 + c];
            if (a.res1) x += (float)a.res1[(size_t)p * a.ldr1 + c];
            if (a.mask) x *= 1.f - (float)a.mask[p];
            const h16 o = (h16)x;
            reinterpret_cast<h16*>(a.out)[(size_t)p * a.ldo + c] = o;
            const float q = (float)o;
            ssum += q; ssq += q * q;
        }
    }
    if (a.stats) {
        red[0][pl][cl] = ssum; red[1][pl][cl] = ssq;
        __syncthreads();
        if (pl == 0 && c < a.Q) {
            float* sp = a.stats + ((size_t)blockIdx.x * a.Q + c) * 2;
            sp[0] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
            sp[1] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
        }
    }
}

// Fallback split-K workspace for launches that bring none (op-level entry points, one-off GEMMs outside a planned arena).  GROW-ONLY and
// never freed while the process lives: a captured hipGraph bakes the pointer its launches were recorded with, so a buffer that was ever
// handed out must stay valid (retired buffers are kept; growth is geometric, so the waste is bounded by the largest request).  The module
// graphs do not use it: their slabs come from the handle's planned arena (launch_conv_into), whose base address is part of the graph key.
float* g_ws = nullptr;
size_t g_ws_bytes = 0;
std::vector<float*> g_ws_retired;
bool ensure_ws(size_t bytes, hipStream_t st) {
    if (bytes <= g_ws_bytes) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;  // cannot allocate in a capture
    const size_t want = std::max(bytes, 2 * g_ws_bytes);
    float* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), want) != hipSuccess) return false;
    if (g_ws) g_ws_retired.push_back(g_ws);
    g_ws = p; g_ws_bytes = want;
    return true;
}

struct CfgInfo { int bq, bp, blocks_per_cu; bool geglu_ok; float eff; int tp; int base; int split; int bk; bool tune; };
// tile configurations (0 = choose: measured per shape when autotuning is on, else the cost model below).
// {bq, bp, workgroups per CU (cost model), GEGLU-capable, cost-model efficiency (0 = measured selection only), tp, base kernel, split-K,
//  K step, offered to the tuner}
constexpr int NCFG = 109;
const CfgInfo kCfg[NCFG + 1] = {
    {0, 0, 0, false, 0.f, 0, 0, 1, 0, false},
    {128, 256, 2, true, 0.80f, 4, 1, 1, 32, true},   // 1: <2,2,2,4> BK32 NST3
    {320, 128, 2, false, 1.00f, 2, 2, 1, 32, true},  // 2: <2,2,5,2> BK32 NST2 (Cout = 320 layers, no padding waste)
    {128, 128, 3, true, 1.00f, 2, 3, 1, 32, true},   // 3: <2,2,2,2> BK32 NST3
    {128, 64, 4, true, 0.80f, 1, 4, 1, 32, true},    // 4: <2,2,2,1> BK32 NST3
    {64, 64, 6, false, 0.60f, 1, 5, 1, 32, true},    // 5: <2,2,1,1> BK32 NST3
    {256, 128, 2, true, 0.85f, 2, 6, 1, 32, true},   // 6: <2,2,4,2> BK32 NST3
    {128, 128, 2, true, 0.00f, 2, 7, 1, 64, true},   // 7: <2,2,2,2> BK64 NST2   (eff 0: picked by measurement only, not by the fallback cost model)
    {128, 256, 1, true, 0.00f, 4, 8, 1, 64, false},  // 8: <2,2,2,4> BK64 NST2   (never won a shape: not offered to the tuner)
    {128, 64, 2, true, 0.00f, 1, 9, 1, 64, true},    // 9: <2,2,2,1> BK64 NST3
    {320, 128, 1, false, 0.00f, 2, 10, 1, 64, false},  // 10: <2,2,5,2> BK64 NST2
    {128, 64, 2, false, 0.00f, 1, 9, 2, 64, true},    // 11: cfg 9 + split-K 2
    {128, 64, 2, false, 0.00f, 1, 9, 4, 64, true},    // 12: cfg 9 + split-K 4
    {128, 64, 2, false, 0.00f, 1, 9, 8, 64, true},    // 13: cfg 9 + split-K 8
    {128, 128, 2, false, 0.00f, 2, 7, 2, 64, true},   // 14: cfg 7 + split-K 2
    {128, 128, 2, false, 0.00f, 2, 7, 4, 64, true},   // 15: cfg 7 + split-K 4
    {64, 64, 5, false, 0.00f, 1, 16, 1, 32, true},    // 16: <2,2,1,1> BK32 NST4 (deeper prefetch for shallow-K, latency-bound GEMMs)
    {128, 64, 3, true, 0.00f, 1, 17, 1, 32, true},    // 17: <2,2,2,1> BK32 NST4
    {128, 128, 2, true, 0.00f, 2, 18, 1, 32, true},   // 18: <2,2,2,2> BK32 NST4
    {128, 256, 2, true, 0.00f, 2, 19, 1, 32, true},   // 19: <2,4,2,2> BK32 NST3   8 waves (512 threads)
    {256, 128, 2, true, 0.00f, 2, 20, 1, 32, true},   // 20: <4,2,2,2> BK32 NST3   8 waves
    {256, 256, 1, true, 0.00f, 2, 21, 1, 32, true},   // 21: <2,4,4,2> BK32 NST3   8 waves, 96 KB LDS
    {320, 256, 1, false, 0.00f, 2, 22, 1, 64, true},  // 22: <2,4,5,2> BK64 NST2   8 waves, 144 KB LDS
    // 23..27: X-stationary linear kernel (linear_xs.hip; K = 320 / 640 token-wise projections).  bq field = channel slices
    // over gridDim.y, tp field = 32-pixel blocks per wave
    {1, 256, 2, false, 0.00f, 2, 23, 1, 0, true},    // 23: 64 pixels / wave, 1 channel slice
    {2, 256, 2, false, 0.00f, 2, 23, 1, 0, true},    // 24: 64 pixels / wave, 2 channel slices
    {1, 128, 2, true, 0.00f, 1, 23, 1, 0, true},     // 25: 32 pixels / wave, 1 channel slice   (25..27 also: residual, GEGLU)
    {2, 128, 2, true, 0.00f, 1, 23, 1, 0, true},     // 26: 32 pixels / wave, 2 channel slices
    {5, 128, 2, true, 0.00f, 1, 23, 1, 0, true},     // 27: 32 pixels / wave, 5 channel slices
    // 28..31: 8-wave tiles + split-K for the deep-K, few-pixel levels (half the staging bytes per MAC of the 128x64 tile)
    {256, 128, 2, false, 0.00f, 2, 20, 4, 32, true},  // 28: cfg 20 + split-K 4
    {256, 128, 2, false, 0.00f, 2, 20, 8, 32, true},  // 29: cfg 20 + split-K 8
    {128, 256, 2, false, 0.00f, 2, 19, 4, 32, true},  // 30: cfg 19 + split-K 4
    {128, 256, 2, false, 0.00f, 2, 19, 8, 32, true},  // 31: cfg 19 + split-K 8
    // 32..38: igemm8_kernel, the phase-staggered 8-wave large-tile pipeline (BK = 64, one workgroup per CU, 2 K-tile buffers)
    {320, 256, 1, false, 0.00f, 2, 32, 1, 64, true},  // 32: igemm8<5,2>  (Cout = 320 / 640 / 960 / 1280 layers without padding waste)
    {256, 256, 1, true, 0.00f, 2, 33, 1, 64, true},   // 33: igemm8<4,2>
    {320, 256, 1, false, 0.00f, 2, 32, 2, 64, true},  // 34: cfg 32 + split-K 2
    {320, 256, 1, false, 0.00f, 2, 32, 4, 64, true},  // 35: cfg 32 + split-K 4
    {256, 256, 1, false, 0.00f, 2, 33, 2, 64, true},  // 36: cfg 33 + split-K 2
    {256, 256, 1, false, 0.00f, 2, 33, 4, 64, true},  // 37: cfg 33 + split-K 4
    {256, 256, 1, false, 0.00f, 2, 33, 8, 64, true},  // 38: cfg 33 + split-K 8
    // 39..53 (round 3): 4-wave tiles with ONE workgroup per CU (one wave per SIMD, up to 240 accumulator registers): grids that fill the
    // 256 CUs at batch 8 and rings that keep 64-128 KB in flight per CU; DMA issue interleaved with the MFMA groups
    {320, 192, 1, false, 0.00f, 3, 39, 1, 64, true},  // 39: <2,2,5,3> BK64 NST2 interleaved: 256 tiles on the 49 152-pixel level
    {320, 192, 1, false, 0.00f, 3, 40, 1, 32, true},  // 40: <2,2,5,3> BK32 NST4 interleaved (every stage issued 3 K steps ahead)
    {256, 256, 1, true, 0.00f, 4, 41, 1, 64, true},   // 41: <2,2,4,4> BK64 NST2
    {128, 128, 1, true, 0.00f, 2, 42, 1, 64, true},   // 42: <2,2,2,2> BK64 NST4 (128 KB ring)
    {128, 256, 1, true, 0.00f, 4, 43, 1, 64, true},   // 43: <2,2,2,4> BK64 NST3 (144 KB ring)
    {128, 64, 1, true, 0.00f, 1, 44, 1, 64, true},    // 44: <2,2,2,1> BK64 NST5 (120 KB ring)
    {256, 192, 1, true, 0.00f, 3, 45, 1, 64, true},   // 45: <2,2,4,3> BK64 NST2
    {320, 192, 1, false, 0.00f, 3, 39, 2, 64, true},  // 46: cfg 39 + split-K 2
    {128, 128, 2, true, 0.00f, 2, 47, 1, 64, true},   // 47: cfg 7 with interleaved DMA issue
    {128, 64, 2, true, 0.00f, 1, 48, 1, 64, true},    // 48: cfg 9 with interleaved DMA issue
    {256, 256, 1, false, 0.00f, 4, 41, 4, 64, true},  // 49: cfg 41 + split-K 4
    {256, 256, 1, false, 0.00f, 4, 41, 2, 64, true},  // 50: cfg 41 + split-K 2
    {256, 192, 1, false, 0.00f, 3, 45, 3, 64, true},  // 51: cfg 45 + split-K 3
    {128, 128, 2, false, 0.00f, 2, 47, 2, 64, true},  // 52: cfg 47 + split-K 2
    {128, 64, 2, false, 0.00f, 1, 48, 4, 64, true},   // 53: cfg 48 + split-K 4
    // 54..61 (round 3): more shapes of the phase-staggered 8-wave pipeline, igemm8<TQ,TP> = (64 TQ) x (128 TP)
    {128, 256, 1, true, 0.00f, 2, 54, 1, 64, true},   // 54: igemm8<2,2>
    {256, 128, 1, true, 0.00f, 1, 55, 1, 64, true},   // 55: igemm8<4,1>
    {128, 128, 2, true, 0.00f, 1, 56, 1, 64, true},   // 56: igemm8<2,1> (64 KB of LDS: two workgroups = 16 waves per CU)
    {320, 128, 1, false, 0.00f, 1, 57, 1, 64, true},  // 57: igemm8<5,1>
    {192, 256, 1, false, 0.00f, 2, 58, 1, 64, true},  // 58: igemm8<3,2>
    {128, 256, 1, false, 0.00f, 2, 54, 2, 64, true},  // 59: cfg 54 + split-K 2
    {256, 128, 1, false, 0.00f, 1, 55, 2, 64, true},  // 60: cfg 55 + split-K 2
    {320, 128, 1, false, 0.00f, 1, 57, 2, 64, true},  // 61: cfg 57 + split-K 2
    // 62..73 (round 3): loader / consumer kernel (igemm_lc.hip): 2 x 2 consumer waves that only read LDS and issue MFMAs + 2 loader waves
    // that keep NST-1 whole K tiles of LDS-DMA in flight; one workgroup (6 waves) per CU
    {128, 128, 1, true, 0.00f, 2, 62, 1, 64, true},   // 62: 128x128, 4-deep ring (128 KB)
    {128, 128, 1, true, 0.00f, 2, 63, 1, 64, true},   // 63: 128x128, 5-deep ring (160 KB)
    {256, 128, 1, true, 0.00f, 2, 64, 1, 64, true},   // 64: 256x128, 3-deep ring (144 KB)
    {128, 256, 1, true, 0.00f, 4, 65, 1, 64, true},   // 65: 128x256, 3-deep ring (144 KB)
    {128, 64, 1, true, 0.00f, 1, 66, 1, 64, true},    // 66: 128x64, 6-deep ring (144 KB)
    {320, 128, 1, false, 0.00f, 2, 67, 1, 64, true},  // 67: 320x128, 2-deep ring (112 KB)
    {192, 192, 1, false, 0.00f, 3, 68, 1, 64, true},  // 68: 192x192, 3-deep ring (144 KB)
    {128, 128, 1, false, 0.00f, 2, 62, 2, 64, true},  // 69: cfg 62 + split-K 2
    {256, 128, 1, false, 0.00f, 2, 64, 2, 64, true},  // 70: cfg 64 + split-K 2
    {256, 128, 1, false, 0.00f, 2, 64, 4, 64, true},  // 71: cfg 64 + split-K 4
    {128, 64, 1, false, 0.00f, 1, 66, 4, 64, true},   // 72: cfg 66 + split-K 4
    {128, 128, 1, false, 0.00f, 2, 62, 4, 64, true},  // 73: cfg 62 + split-K 4
    // 74..83 (round 3): halo-resident 3x3 convolution (igemm_halo.hip): the pixel range of all nine taps staged ONCE per channel chunk,
    // only the weight tile streamed per tap; 8 waves, one workgroup per CU
    {128, 256, 1, false, 0.00f, 2, 74, 1, 64, true},  // 74: 128x256, double halo buffer
    {256, 256, 1, false, 0.00f, 2, 75, 1, 64, true},  // 75: 256x256, single halo buffer
    {320, 256, 1, false, 0.00f, 2, 76, 1, 64, true},  // 76: 320x256, single halo buffer
    {128, 128, 1, false, 0.00f, 1, 77, 1, 64, true},  // 77: 128x128
    {256, 128, 1, false, 0.00f, 1, 78, 1, 64, true},  // 78: 256x128
    {128, 256, 1, false, 0.00f, 2, 74, 2, 64, true},  // 79: cfg 74 + split-K 2
    {128, 128, 1, false, 0.00f, 1, 77, 2, 64, true},  // 80: cfg 77 + split-K 2
    {256, 128, 1, false, 0.00f, 1, 78, 2, 64, true},  // 81: cfg 78 + split-K 2
    {256, 256, 1, false, 0.00f, 2, 75, 2, 64, true},  // 82: cfg 75 + split-K 2
    {128, 128, 1, false, 0.00f, 1, 77, 4, 64, true},  // 83: cfg 77 + split-K 4
    {128, 128, 2, false, 0.00f, 2, 84, 1, 64, true},  // 84: halo 128x128, 4 waves, two workgroups per CU
    {128, 192, 2, false, 0.00f, 3, 85, 1, 64, true},  // 85: halo 128x192, 4 waves, two workgroups per CU
    {128, 128, 2, false, 0.00f, 2, 84, 2, 64, true},  // 86: cfg 84 + split-K 2
    {128, 128, 2, false, 0.00f, 2, 84, 4, 64, true},  // 87: cfg 84 + split-K 4
    // 88..92 (round 4): halo kernel with the buffer sized for rows <= 24 pixels (a third weight slot at two workgroups per CU), and the
    // 12-wave 320x192 form: 256 workgroups on the 64x48 level at batch 8 (the 320x256 tiles leave a quarter of the chip idle there)
    {128, 128, 2, false, 0.00f, 2, 88, 1, 64, true},  // 88: halo 128x128, 4 waves, W <= 24, 3 weight slots
    {128, 192, 2, false, 0.00f, 3, 89, 1, 64, true},  // 89: halo 128x192, 4 waves, W <= 24, 3 weight slots
    {128, 128, 2, false, 0.00f, 2, 88, 2, 64, true},  // 90: cfg 88 + split-K 2
    {128, 128, 2, false, 0.00f, 2, 88, 4, 64, true},  // 91: cfg 88 + split-K 4
    {320, 192, 1, false, 0.00f, 1, 92, 1, 64, true},  // 92: halo 320x192, 12 waves
    // 93..95 (round 4): X-stationary kernel with a two-slot weight ring = THREE workgroups per CU (K = 320: the 64x48 level)
    {1, 128, 3, true, 0.00f, 1, 93, 1, 0, true},     // 93: 32 pixels / wave, 1 channel slice
    {2, 128, 3, true, 0.00f, 1, 93, 1, 0, true},     // 94: 32 pixels / wave, 2 channel slices
    {5, 128, 3, true, 0.00f, 1, 93, 1, 0, true},     // 95: 32 pixels / wave, 5 channel slices
    // 96 (round 5): halo 320x192 on FOUR waves, one per SIMD (160 x 96 per wave): the twelve-wave tile's fill of the 64x48 level with the
    // LDS reads per MFMA of the one-wave-per-SIMD ring tiles
    // measured (profiles/r05_halo_one_wave.txt): 107 us on the 320 -> 320 convolution where the twelve-wave form takes 92 -- at one wave per SIMD
    // nothing covers the exposed halo-tile load of each channel chunk and the wave's own ds_read latency; kept selectable, not offered to the tuner
    {320, 192, 1, false, 0.00f, 3, 96, 1, 64, false},  // 96: halo 320x192, 4 waves, one per SIMD
    // 97..99 (round 5): halo 128x256 on four waves of 64 x 128, two workgroups per CU, rows <= 24 pixels -- what the ablation of cfg 88 asks for
    // (profiles/r05_halo_ablate.txt: its fragment reads cost twice what its DMA costs): 0.75 KB of LDS reads per MFMA instead of 1, half the
    // weight DMA per MFMA; split-K keeps two workgroups per CU on the 32x24 / 16x12 levels
    {128, 256, 2, false, 0.00f, 4, 97, 1, 64, true},  // 97: halo 128x256, 4 waves, W <= 24
    {128, 256, 2, false, 0.00f, 4, 97, 2, 64, true},  // 98: cfg 97 + split-K 2
    {128, 256, 2, false, 0.00f, 4, 97, 4, 64, true},  // 99: cfg 97 + split-K 4
    // 100..103 (round 5): 2-D BLOCKED halo tiles (igemm_halo.hip G2D): stride-1 3x3 convolutions on images wider than 48 pixels -- the VAE /
    // EMASC levels above 64x48 and the 128x96 latent grid of 1024x768 -- stage (TH + 2) x 34 pixels per TH x 32 block instead of nine shifted copies
    {128, 256, 1, false, 0.00f, 2, 100, 1, 64, true},  // 100: halo2d 128x256 (8 rows x 32), 8 waves
    {256, 256, 1, false, 0.00f, 2, 101, 1, 64, true},  // 101: halo2d 256x256, 8 waves
    {320, 256, 1, false, 0.00f, 2, 102, 1, 64, true},  // 102: halo2d 320x256, 8 waves
    {128, 128, 2, false, 0.00f, 2, 103, 1, 64, true},  // 103: halo2d 128x128 (4 rows x 32), 4 waves, two workgroups per CU
    // 104..108 (round 6): halo forms of the FOLDED-UPSAMPLE convolution (igemm_halo_kernel.h UPS: nearest 2x + 3x3 of Upsample2D): the low-resolution
    // rows a tile touches staged once per channel chunk, taps applied at read time -- the ring kernels gather every tap with per-lane addressing
    // (64x48 640 -> 640: 382 us, 32x24 1280 -> 1280: 379 us in round 5; here 307 / 304 us, profiles/r06_halo_ups.txt)
    {128, 192, 2, false, 0.00f, 3, 104, 1, 64, true},  // 104: halo-ups 128x192, 4 waves, two workgroups per CU
    {320, 192, 1, false, 0.00f, 1, 105, 1, 64, true},  // 105: halo-ups 320x192, 12 waves
    {128, 128, 2, false, 0.00f, 2, 106, 1, 64, true},  // 106: halo-ups 128x128, 4 waves, two workgroups per CU
    {128, 192, 2, false, 0.00f, 3, 104, 2, 64, true},  // 107: cfg 104 + split-K 2
    {128, 128, 2, false, 0.00f, 2, 106, 2, 64, true},  // 108: cfg 106 + split-K 2
    // 109 (round 6): the 8x6 level's convolutions are chains of 45 dependent K steps per slice at split 4 (latency, not bandwidth: profiles/r06_halo_map3.txt);
    // split 8 on the two-workgroups-per-CU form halves the chain (480 workgroups = one round) at the price of twice the slab traffic -- offered, the tuner decides
    {128, 128, 2, false, 0.00f, 2, 88, 8, 64, true},   // 109: cfg 88 + split-K 8
};
inline bool is_xs(int base) { return base == 23 || base == 93; }
inline int xs_nst(int base) { return base == 93 ? 2 : 3; }
inline bool is_lc(int base) { return base >= 62 && base <= 68; }
// kernels whose every wave reaches the shared epilogue can combine their K slices in the launch (igemm_common.h igemm_splitk_combine); the
// loader / consumer kernel (its loader waves hold no accumulators) and the X-stationary kernel keep the two-pass form
inline bool sk_inline_ok(int base) { return base != 23 && base != 93 && !is_lc(base); }
// arrival counters for launches that bring none (op-level entry points and the module handles that set no Ctx::sk_cnt): one buffer per
// (device, stream), zeroed once, every launch leaves it zeroed -- two handles working on different streams or devices at the same time
// never share a tile's ticket.  Never freed (a captured graph may hold the pointer).
constexpr int SK_MAX_TILES = 1024;
std::mutex g_sk_mu;
std::map<std::pair<int, hipStream_t>, int*> g_sk_cnts;
int* ensure_sk_cnt(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_sk_mu);
    auto it = g_sk_cnts.find({dev, st});
    if (it != g_sk_cnts.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;   // no allocation inside a capture
    int* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), SK_MAX_TILES * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, SK_MAX_TILES * sizeof(int)) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_sk_cnts[{dev, st}] = p;
    return p;
}
int g_sk_two_pass = -1;       // -1: environment LADI_SPLITK_TWO_PASS decides (read once)
bool sk_two_pass_forced() {
    if (g_sk_two_pass < 0) g_sk_two_pass = getenv("LADI_SPLITK_TWO_PASS") != nullptr ? 1 : 0;
    return g_sk_two_pass == 1;
}
inline bool is_halo(int base) { return (base >= 74 && base <= 78) || base == 84 || base == 85 || base == 88 || base == 89 || base == 92 || base == 96 || base == 97; }
inline bool is_halo2d(int base) { return base >= 100 && base <= 103; }
inline int halo2d_th(int base) { return base == 103 ? 4 : 8; }
inline bool is_halo_ups(int base) { return base >= 104 && base <= 106; }

// rocprofv3's name of the kernel a configuration launches (bench.py groups its per-launch timings by symbol)
std::string cfg_symbol(int c) {
    const int b = kCfg[c].base;
    if (b == 23 || b == 93) return "linear_xs_kernel";
    switch (b) {
        case 32: return "igemm8_kernel<5, 2, 0>";
        case 33: return "igemm8_kernel<4, 2, 0>";
        case 54: return "igemm8_kernel<2, 2, 0>";
        case 55: return "igemm8_kernel<4, 1, 0>";
        case 56: return "igemm8_kernel<2, 1, 0>";
        case 57: return "igemm8_kernel<5, 1, 0>";
        case 58: return "igemm8_kernel<3, 2, 0>";
        case 74: return "igemm_halo_kernel<2, 2, 2, 3, 4, 48, 0, 0, 0>";
        case 75: return "igemm_halo_kernel<4, 2, 1, 3, 4, 48, 0, 0, 0>";
        case 76: return "igemm_halo_kernel<5, 2, 1, 2, 4, 48, 0, 0, 0>";
        case 77: return "igemm_halo_kernel<2, 1, 2, 4, 4, 48, 0, 0, 0>";
        case 78: return "igemm_halo_kernel<4, 1, 1, 3, 4, 48, 0, 0, 0>";
        case 84: return "igemm_halo_kernel<2, 2, 1, 2, 2, 48, 0, 0, 0>";
        case 85: return "igemm_halo_kernel<2, 3, 1, 2, 2, 48, 0, 0, 0>";
        case 88: return "igemm_halo_kernel<2, 2, 1, 3, 2, 24, 0, 0, 0>";
        case 104: return "igemm_halo_kernel<2, 3, 1, 2, 2, 48, 0, 0, 1>";
        case 105: return "igemm_halo_kernel<5, 1, 1, 2, 6, 48, 0, 0, 1>";
        case 106: return "igemm_halo_kernel<2, 2, 1, 2, 2, 48, 0, 0, 1>";
        case 89: return "igemm_halo_kernel<2, 3, 1, 3, 2, 24, 0, 0, 0>";
        case 92: return "igemm_halo_kernel<5, 1, 1, 2, 6, 48, 0, 0, 0>";
        case 96: return "igemm_halo_kernel<5, 3, 1, 2, 2, 48, 1, 0, 0>";
        case 97: return "igemm_halo_kernel<2, 4, 1, 2, 2, 24, 0, 0, 0>";
        case 100: return "igemm_halo_kernel<2, 2, 1, 3, 4, 48, 0, 1, 0>";
        case 101: return "igemm_halo_kernel<4, 2, 1, 3, 4, 48, 0, 1, 0>";
        case 102: return "igemm_halo_kernel<5, 2, 1, 2, 4, 48, 0, 1, 0>";
        case 103: return "igemm_halo_kernel<2, 2, 1, 2, 2, 48, 0, 1, 0>";
        case 62: return "igemm_lc_kernel<2, 2, 2, 2, 2, 4>";
        case 63: return "igemm_lc_kernel<2, 2, 2, 2, 2, 5>";
        case 64: return "igemm_lc_kernel<2, 2, 4, 2, 2, 3>";
        case 65: return "igemm_lc_kernel<2, 2, 2, 4, 2, 3>";
        case 66: return "igemm_lc_kernel<2, 2, 2, 1, 2, 6>";
        case 67: return "igemm_lc_kernel<2, 2, 5, 2, 2, 2>";
        case 68: return "igemm_lc_kernel<2, 2, 3, 3, 2, 3>";
#define X(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV) \
        case base: return "igemm_kernel<" #WQ ", " #WP ", " #TQ ", " #TP ", " #BK ", " #NST ", " #OCC ", " #ILV ">";
        LADI_IGEMM_TILES_ALL(X)
#undef X
        default: return "?";
    }
}

int launch_base(int cfg, const IGemmArgs& a, int batch, hipStream_t st) {
    switch (kCfg[cfg].base) {
#define X(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV) case base: return ladi_igemm_launch_base_##base(a, batch, st);
        LADI_IGEMM_TILES_ALL(X)
#undef X
        case 23: return ladi_launch_linear_xs(a, kCfg[cfg].tp, kCfg[cfg].bq, st, 3);
        case 93: return ladi_launch_linear_xs(a, kCfg[cfg].tp, kCfg[cfg].bq, st, 2);
        case 32: return ladi_launch_igemm8(a, 5, 2, batch, st);
        case 33: return ladi_launch_igemm8(a, 4, 2, batch, st);
        case 54: return ladi_launch_igemm8(a, 2, 2, batch, st);
        case 55: return ladi_launch_igemm8(a, 4, 1, batch, st);
        case 56: return ladi_launch_igemm8(a, 2, 1, batch, st);
        case 57: return ladi_launch_igemm8(a, 5, 1, batch, st);
        case 58: return ladi_launch_igemm8(a, 3, 2, batch, st);
        case 74: return ladi_launch_igemm_halo(a, 2, 2, 2, batch, st);
        case 75: return ladi_launch_igemm_halo(a, 4, 2, 1, batch, st);
        case 76: return ladi_launch_igemm_halo(a, 5, 2, 1, batch, st);
        case 77: return ladi_launch_igemm_halo(a, 2, 1, 2, batch, st);
        case 78: return ladi_launch_igemm_halo(a, 4, 1, 1, batch, st);
        case 84: return ladi_launch_igemm_halo(a, 2, 2, 10, batch, st);
        case 85: return ladi_launch_igemm_halo(a, 2, 3, 10, batch, st);
        case 88: return ladi_launch_igemm_halo(a, 2, 2, 11, batch, st);
        case 89: return ladi_launch_igemm_halo(a, 2, 3, 11, batch, st);
        case 92: return ladi_launch_igemm_halo(a, 5, 1, 12, batch, st);
        case 96: return ladi_launch_igemm_halo(a, 5, 3, 13, batch, st);
        case 97: return ladi_launch_igemm_halo(a, 2, 4, 11, batch, st);
        case 100: return ladi_launch_igemm_halo(a, 2, 2, 20, batch, st);
        case 101: return ladi_launch_igemm_halo(a, 4, 2, 20, batch, st);
        case 102: return ladi_launch_igemm_halo(a, 5, 2, 20, batch, st);
        case 103: return ladi_launch_igemm_halo(a, 2, 2, 21, batch, st);
        case 104: return ladi_launch_igemm_halo(a, 2, 3, 30, batch, st);
        case 105: return ladi_launch_igemm_halo(a, 5, 1, 31, batch, st);
        case 106: return ladi_launch_igemm_halo(a, 2, 2, 30, batch, st);
        case 62: return ladi_launch_igemm_lc(a, 2, 2, 4, batch, st);
        case 63: return ladi_launch_igemm_lc(a, 2, 2, 5, batch, st);
        case 64: return ladi_launch_igemm_lc(a, 4, 2, 3, batch, st);
        case 65: return ladi_launch_igemm_lc(a, 2, 4, 3, batch, st);
        case 66: return ladi_launch_igemm_lc(a, 2, 1, 6, batch, st);
        case 67: return ladi_launch_igemm_lc(a, 5, 2, 2, batch, st);
        case 68: return ladi_launch_igemm_lc(a, 3, 3, 3, batch, st);
        default: return -7;
    }
}

struct ProfRec { hipEvent_t e0, e1; int cfg; double flops; int P, Q, K, ks; char sym[56]; };
bool g_prof = false;
std::vector<ProfRec> g_recs;

struct TuneKey {
    int P, Q, K, C0, C1, Wo, flags, batch;
    bool operator==(const TuneKey& o) const {
        return P == o.P && Q == o.Q && K == o.K && C0 == o.C0 && C1 == o.C1 && Wo == o.Wo && flags == o.flags && batch == o.batch;
    }
};
struct TuneHash {
    size_t operator()(const TuneKey& k) const {
        size_t h = 1469598103934665603ULL;
        const int v[8] = {k.P, k.Q, k.K, k.C0, k.C1, k.Wo, k.flags, k.batch};
        for (int x : v) { h ^= (size_t)(unsigned)x; h *= 1099511628211ULL; }
        return h;
    }
};
bool g_autotune = true;
std::unordered_map<TuneKey, int, TuneHash> g_tuned;
// optional persistence of the measured choices across processes (LADI_TUNE_CACHE=<file>): one line per problem shape
bool g_cache_loaded = false;
void tune_cache_read(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return;
    TuneKey k; int cfg;
    char line[256];
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#') continue;
        if (sscanf(line, "%d %d %d %d %d %d %d %d %d", &k.P, &k.Q, &k.K, &k.C0, &k.C1, &k.Wo, &k.flags, &k.batch, &cfg) == 9 && cfg >= 1 && cfg <= NCFG)
            g_tuned[k] = cfg;
    }
    fclose(f);
}
// Selections are loaded from (1) the table shipped next to the library (tune_gfx950.txt: the choices measured on MI355X for the layer
// shapes of the released model at the BASELINE batch sizes, so that those runs pick the same tiles -- hence the same summation order and
// the same last bits -- in every process) and (2) LADI_TUNE_CACHE=<file> (read, and appended to when a new shape is measured).
void tune_cache_load() {
    if (g_cache_loaded) return;
    g_cache_loaded = true;
    if (!getenv("LADI_TUNE_NO_SHIPPED")) {
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&ladi_igemm_num_cfgs), &info) && info.dli_fname) {
            std::string p(info.dli_fname);
            const size_t s = p.find_last_of('/');
            tune_cache_read(((s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/tune_gfx950.txt").c_str());
        }
    }
    if (const char* path = getenv("LADI_TUNE_CACHE")) tune_cache_read(path);
}
void tune_cache_append(const TuneKey& k, int cfg) {
    const char* path = getenv("LADI_TUNE_CACHE");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%d %d %d %d %d %d %d %d %d\n", k.P, k.Q, k.K, k.C0, k.C1, k.Wo, k.flags, k.batch, cfg);
    fclose(f);
}

}  // namespace

int ladi_igemm_num_cfgs() { return NCFG; }
// kernel symbol (as rocprofv3 prints it, without the "void " / argument list) a tile configuration launches; "" for an unknown id
const char* ladi_igemm_cfg_symbol(int cfg) {
    static std::string names[NCFG + 1];
    if (cfg < 1 || cfg > NCFG) return "";
    if (names[cfg].empty()) names[cfg] = cfg_symbol(cfg);
    return names[cfg].c_str();
}

// the split-K admission rule (shared by the tuner and the workspace planner): few output tiles, deep K
static bool splitk_admissible(const IGemmArgs& a, int c) {
    const long long tiles = (long long)((a.Q + kCfg[c].bq - 1) / kCfg[c].bq) * ((a.P + kCfg[c].bp - 1) / kCfg[c].bp);
    return !(tiles * kCfg[c].split > 1024 || tiles > 256 || (a.K / 64) / kCfg[c].split < 8);
}

// Everything that makes configuration c illegal (or, with `strict`, pointless) for this launch.  The tuner, a selection read from the
// tune table and an explicitly requested configuration all pass through it: the tune key folds several epilogue features into one bit,
// so a cached choice is re-validated against the launch it is applied to (a split-K selection must never reach a launch with a
// per-pixel bias or an fp32 output, nor one whose planned slab was sized without it).
static bool cfg_admissible(const IGemmArgs& a, int batch, int c, bool strict) {
    const CfgInfo& ci = kCfg[c];
    const bool geglu = a.act == LADI_ACT_GEGLU;
    if (is_xs(ci.base)) {
        IGemmArgs t = a; t.stats = nullptr;
        return ladi_linear_xs_eligible(t, batch, ci.tp, ci.bq, xs_nst(ci.base));
    }
    if (a.gn_ss) return false;                                        // GroupNorm affine of the operand: X-stationary kernel only
    if (a.ln_gamma && !a.ln_scratch) return false;                    // no scratch: only the fused (X-stationary) form
    if (is_lc(ci.base) && (a.ups || batch != 1)) return false;        // loader / consumer kernel: no folded upsample, no batched launches
    if (is_halo_ups(ci.base) && !ladi_igemm_halo_ups_eligible(a, batch, ci.bp)) return false;   // folded-upsample halo: single source, whole tiles inside a sample
    if (is_halo2d(ci.base) && !ladi_igemm_halo2d_eligible(a, batch, halo2d_th(ci.base))) return false;   // 2-D blocked halo: W % 32 == 0, whole blocks
    if (is_halo(ci.base) && (!ladi_igemm_halo_eligible(a, batch) || ((ci.base == 88 || ci.base == 89 || ci.base == 97) && a.Ws > 24))) return false;   // halo-resident kernel: 3x3 stride-1 convolutions on narrow images
    if (geglu && !ci.geglu_ok) return false;
    if (ci.bk == 64 && ((a.C0 % 64) || (a.C1 % 64))) return false;
    if (ci.split > 1 && (batch != 1 || geglu || a.out_f32 || a.bias_per_pixel)) return false;
    if (strict) {
        if (ci.bq > 2 * a.Q && ci.bq > 64) return false;              // grossly oversized in Q
        if (ci.split > 1 && !splitk_admissible(a, c)) return false;   // split-K: few tiles, deep K only
    }
    return true;
}

// slab bytes configuration c needs: the two-pass form writes [split][P][Q] floats, the in-launch form whole (padded) tile images
static size_t splitk_cfg_bytes(const IGemmArgs& a, int c) {
    const CfgInfo& ci = kCfg[c];
    const size_t tiles = (size_t)((a.Q + ci.bq - 1) / ci.bq) * (size_t)((a.P + ci.bp - 1) / ci.bp);
    return (size_t)ci.split * std::max((size_t)a.P * (size_t)a.Q, tiles * (size_t)ci.bq * (size_t)ci.bp) * sizeof(float);
}
size_t ladi_igemm_splitk_ws_bytes(const IGemmArgs& a, int batch) {
    if (batch != 1 || a.act == LADI_ACT_GEGLU || a.out_f32 || a.bias_per_pixel || a.P <= 0 || a.Q <= 0) return 0;
    size_t need = 0;
    for (int c = 1; c <= NCFG; ++c)
        if (kCfg[c].split > 1 && !is_xs(kCfg[c].base) && splitk_admissible(a, c)) need = std::max(need, splitk_cfg_bytes(a, c));
    return need;
}

int ladi_launch_igemm(const IGemmArgs& a_in, int batch, int cfg, hipStream_t st, int* stats_row_px, float* ws, size_t ws_bytes, int* sk_cnt) {
    IGemmArgs a = a_in;
    const int cfg_in = cfg;
    if (stats_row_px) *stats_row_px = 0;
    if (a.ksize != 1 && a.ksize != 3 && a.ksize != 4) return -1;   // 4: the stride-2 convs of the TPS matching network
    if ((a.C0 % 32) || (a.C1 % 32)) return -2;
    if (a.K != a.ksize * a.ksize * (a.C0 + a.C1)) return -3;
    if ((a.ld0 % 8) || (a.C1 && (a.ld1 % 8)) || (a.K % 8) || (a.ldw % 8)) return -4;
    if (a.P <= 0 || a.Q <= 0) return -5;
    const bool geglu = a.act == LADI_ACT_GEGLU;
    if (geglu && (a.Q % 64)) return -6;
    // ---- measured tile-shape selection ("measure, don't guess"): the first time a problem shape is seen outside a stream
    //      capture, every admissible configuration is timed with HIP events on the launch stream and the fastest is cached.
    //      Re-running a launch is idempotent (outputs never alias inputs in this library).
    // flags also carry the epilogue features that decide which kernels are admissible (residuals, fused statistics, other)
    // (bits 12 / 13 sit ABOVE the ksize field at bits 8..10: at 256 / 512 they were absorbed by ksize = 1 / 3 and a fused-GroupNorm
    // projection shared its entry with the plain 1x1 of the same shape -- ADVICE r04)
    const int epi = ((a.bias_mul != 0.f && a.bias_mul != 1.f) ? (1 << 13) : 0) | (a.gn_ss ? (1 << 12) : 0) | (a.ln_gamma ? 128 : 0) | ((a.res0 || a.res1) ? 2 : 0) | (a.stats ? 8 : 0) |
                    ((a.rowadd || a.mask || a.bias_per_pixel || a.out_f32 || a.out_scale != 1.f || (a.act != LADI_ACT_NONE && !geglu)) ? 64 : 0);
    TuneKey key{a.P, a.Q, a.K, a.C0, a.C1, a.Wo, (a.ksize << 8) | (a.stride << 4) | (a.ups << 2) | (geglu ? 1 : 0) | epi, batch};
    if (cfg == 0 && g_autotune) {
        tune_cache_load();
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) cfg = it->second;
        else {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone && !g_prof) {
                const bool was = g_autotune;
                g_autotune = false;
                hipEvent_t e0, e1;
                if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                    float best_ms = 1e30f; int best_cfg = 0;
                    std::vector<std::pair<float, int>> cand;     // (ms per launch incl. penalty, cfg)
                                        for (int c = 1; c <= NCFG; ++c) {
                        if (!kCfg[c].tune || !cfg_admissible(a, batch, c, true)) continue;
                        // X-stationary kernel: no fused statistics there, charge the separate statistics pass the consumer then needs (~3 TB/s read)
                        const float penalty_ms = (is_xs(kCfg[c].base) && a.stats) ? 3.f * (float)((double)a.P * a.Q * 2.0 / 3.0e9) : 0.f;
                        if (ladi_launch_igemm(a, batch, c, st, nullptr, ws, ws_bytes, sk_cnt) != 0) continue;   // warm-up (also sets function attributes)
                        (void)hipEventRecord(e0, st);
                        for (int r = 0; r < 3; ++r) (void)ladi_launch_igemm(a, batch, c, st, nullptr, ws, ws_bytes, sk_cnt);
                        (void)hipEventRecord(e1, st);
                        if (hipEventSynchronize(e1) != hipSuccess) continue;
                        float ms = 0.f;
                        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
                        cand.push_back({(ms + penalty_ms) / 3.f, c});
                        if (ms + penalty_ms < best_ms) { best_ms = ms + penalty_ms; best_cfg = c; }
                    }
                    // second look at the front-runners: 3 launches are enough to rank the field but not to separate candidates a few
                    // percent apart (a mis-pick costs that layer on every forward of the run), so the best three within 15 % are
                    // re-timed over 10 launches each
                    if (cand.size() > 1) {
                        std::sort(cand.begin(), cand.end());
                        const float lim = cand[0].first * 1.15f;
                        float best2 = 1e30f; int best2_cfg = 0;
                        for (size_t i = 0; i < cand.size() && i < 3 && cand[i].first <= lim; ++i) {
                            const int c = cand[i].second;
                            float pen = 0.f;
                            if (is_xs(kCfg[c].base) && a.stats) pen = (float)((double)a.P * a.Q * 2.0 / 3.0e9);
                            (void)hipEventRecord(e0, st);
                            for (int r = 0; r < 10; ++r) (void)ladi_launch_igemm(a, batch, c, st, nullptr, ws, ws_bytes, sk_cnt);
                            (void)hipEventRecord(e1, st);
                            float ms = 0.f;
                            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
                            if (ms / 10.f + pen < best2) { best2 = ms / 10.f + pen; best2_cfg = c; }
                        }
                        if (best2_cfg) best_cfg = best2_cfg;
                    }
                    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
                    if (best_cfg) { g_tuned[key] = best_cfg; cfg = best_cfg; tune_cache_append(key, best_cfg); }
                }
                g_autotune = was;
            }
        }
    }
    if (cfg >= 1 && cfg <= NCFG && cfg_in == 0 && !cfg_admissible(a, batch, cfg, true)) cfg = 0;   // stale table entry / key collision: cost model
    if (cfg == 0) {
        // fallback cost model: (waves of workgroups over the chip) x (tile work) / (per-tile efficiency)
        double best = 1e300;
        for (int c = 1; c <= NCFG; ++c) {
            const CfgInfo& ci = kCfg[c];
            if (is_xs(ci.base) || ci.split > 1 || !(ci.eff > 0.f) || !cfg_admissible(a, batch, c, false)) continue;
            const long long tiles = (long long)((a.Q + ci.bq - 1) / ci.bq) * ((a.P + ci.bp - 1) / ci.bp) * batch;
            const long long slots = 256LL * ci.blocks_per_cu;
            const double waves = (double)((tiles + slots - 1) / slots);
            // partial last wave: count it in proportion but never below 35% (a lone straggler still takes a full tile time)
            const double frac = (double)(tiles % slots) / (double)slots;
            const double eff_waves = (tiles % slots) ? (waves - 1.0) + (frac < 0.35 ? 0.35 : frac) : waves;
            const double cost = eff_waves * (double)ci.bq * ci.bp * ci.blocks_per_cu / ci.eff;
            if (cost < best) { best = cost; cfg = c; }
        }
    }
    if (cfg == 0 && a.gn_ss) {      // no measured selection and the cost model does not rank the X-stationary kernel: first eligible form
        for (int c : {25, 26, 27, 23, 24, 93, 94, 95}) if (cfg_admissible(a, batch, c, true)) { cfg = c; break; }
    }
    if (cfg < 1 || cfg > NCFG) return -7;
    if (a.gn_ss && !is_xs(kCfg[cfg].base)) return -17;
    if (a.ln_gamma && !is_xs(kCfg[cfg].base)) {   // LayerNorm as its own kernel into the caller's scratch (timed by the tuner as part of cfg)
        if (!a.ln_scratch || a.C1 || a.src1 || a.ksize != 1 || batch != 1) return -15;
        const int lrc = ladi_launch_layernorm(a.src0, a.ld0, a.ln_gamma, a.ln_beta, a.ln_eps, a.P, a.C0, a.ln_scratch, a.C0, st);
        if (lrc != 0) return -15;
        a.src0 = a.ln_scratch; a.ld0 = a.C0; a.ln_gamma = nullptr; a.ln_beta = nullptr;
    }
    if (is_xs(kCfg[cfg].base)) {   // X-stationary linear kernel: no fused statistics (the consumer falls back to ladi_launch_gn_partial)
        a.stats = nullptr;
        if (!ladi_linear_xs_eligible(a, batch, kCfg[cfg].tp, kCfg[cfg].bq, xs_nst(kCfg[cfg].base))) return -14;
    }
    if (geglu && !kCfg[cfg].geglu_ok) return -8;
    if (kCfg[cfg].bk == 64 && ((a.C0 % 64) || (a.C1 % 64))) return -2;  // BK = 64 variants
    const int split = kCfg[cfg].split;
    bool sk_inline = false;
    if (split > 1) {
        if (batch != 1 || geglu || a.out_f32 || a.bias_per_pixel) return -9;
        const size_t need = splitk_cfg_bytes(a, cfg);
        if (!ws || ws_bytes < need) {           // no (or too small a) caller slab: process-wide grow-only fallback
            if (!ensure_ws(need, st)) return -13;
            ws = g_ws;
        }
        const long long tiles = (long long)((a.Q + kCfg[cfg].bq - 1) / kCfg[cfg].bq) * ((a.P + kCfg[cfg].bp - 1) / kCfg[cfg].bp);
        sk_inline = sk_inline_ok(kCfg[cfg].base) && !sk_two_pass_forced() && tiles <= SK_MAX_TILES;
        if (sk_inline && !sk_cnt) {
            sk_cnt = ensure_sk_cnt(st);
            if (!sk_cnt) sk_inline = false;     // first use inside a capture without caller counters: two-pass form
        }
    }
    if (a.stats) {  // fused output statistics need whole 32*TP-pixel row blocks inside one sample
        const int px = ((split > 1 && !sk_inline) ? 1 : kCfg[cfg].tp) * 32;
        if (geglu || batch != 1 || a.out_f32 || ((a.Ho * a.Wo) % px)) a.stats = nullptr;
        else if (stats_row_px) *stats_row_px = px;
    }
    ProfRec rec;
    const bool prof = g_prof;
    if (prof) {
        if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) return -12;
        rec.cfg = cfg; rec.P = a.P; rec.Q = a.Q; rec.K = a.K; rec.ks = a.ksize;
        if (is_xs(kCfg[cfg].base)) ladi_linear_xs_symbol(a, kCfg[cfg].tp, xs_nst(kCfg[cfg].base), rec.sym, (int)sizeof(rec.sym));
        else snprintf(rec.sym, sizeof(rec.sym), "%s", cfg_symbol(cfg).c_str());
        rec.flops = 2.0 * (double)a.P * (double)a.Q * (double)a.K * (double)batch;
        (void)hipEventRecord(rec.e0, st);
    }
    IGemmArgs full = a;   // epilogue parameters for the split-K reduce pass
    int lbatch = batch;
    a.sk_ws = nullptr; a.sk_cnt = nullptr;
    if (split > 1 && sk_inline) {            // in-launch combine: the kernel keeps the real epilogue, the last-arriving slice of a tile runs it
        a.splitk = split; lbatch = split; a.sk_ws = ws; a.sk_cnt = sk_cnt;
    } else if (split > 1) {
        a.out = ws; a.ldo = a.Q; a.out_f32 = 1; a.bs_out = (long long)a.P * a.Q; a.splitk = split; lbatch = split;
        a.bias = nullptr; a.rowadd = nullptr; a.act = LADI_ACT_NONE; a.out_scale = 1.f; a.res0 = nullptr; a.res1 = nullptr;
        a.mask = nullptr; a.stats = nullptr;
    } else a.splitk = 1;
    const int batch_l = lbatch;
    int rc;
    rc = launch_base(cfg, a, batch_l, st);
    if (rc == 0 && split > 1 && !sk_inline) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((full.P + 31) / 32), (unsigned)((full.Q + 63) / 64)), dim3(256), 0, st, ws, split, full);
        if (hipGetLastError() != hipSuccess) rc = -11;
    }
    // work-complete timing: a split-K launch is not done until its reduce pass has written the output, so the pass is charged to the
    // symbol of the kernel that needed it (rocprofv3 lists splitk_reduce_kernel separately; profiles/README.md shows how to add it back)
    if (prof) { (void)hipEventRecord(rec.e1, st); g_recs.push_back(rec); }
    return rc;
}

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream around every igemm launch
void ladi_igemm_profile_enable(int on) { g_prof = on != 0; }
void ladi_igemm_autotune(int on) { g_autotune = on != 0; }
void ladi_igemm_splitk_two_pass(int on) { g_sk_two_pass = on ? 1 : 0; }
int ladi_igemm_tuned_count() { return (int)g_tuned.size(); }
int ladi_igemm_profile_symbols(char* buf, int n) {
    std::map<std::string, std::array<double, 3>> by;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return -2;
        auto& e = by[r.sym];
        e[0] += ms; e[1] += r.flops; e[2] += 1.0;
    }
    std::string out;
    for (auto& kv : by) {
        char line[160];
        snprintf(line, sizeof(line), "%s\t%.6f\t%.6e\t%d\n", kv.first.c_str(), kv.second[0], kv.second[1], (int)kv.second[2]);
        out += line;
    }
    if (n > 0) { snprintf(buf, (size_t)n, "%s", out.c_str()); }
    return (int)out.size();
}
// out[cfg*3 + {0,1,2}] = {total ms, algorithmic FLOP (2*P*Q*K), launches} for cfg 1..NCFG (index 0 = all); clears the records
int ladi_igemm_profile_collect(double* out, int n_out) {
    for (int i = 0; i < n_out; ++i) out[i] = 0.0;
    std::map<std::string, std::pair<double, int>> by_shape;
    const bool dump = getenv("LADI_PROF_DUMP") != nullptr;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return -2;
        if (r.cfg * 3 + 2 < n_out) { out[r.cfg * 3 + 0] += ms; out[r.cfg * 3 + 1] += r.flops; out[r.cfg * 3 + 2] += 1.0; }
        out[0] += ms; out[1] += r.flops; out[2] += 1.0;
        if (dump) { char b[128]; snprintf(b, sizeof(b), "P=%d Q=%d K=%d ks=%d cfg=%d gflop=%.1f", r.P, r.Q, r.K, r.ks, r.cfg, r.flops / 1e9); auto& e = by_shape[b]; e.first += ms; e.second += 1; }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_recs.clear();
    if (dump) for (auto& kv : by_shape) fprintf(stderr, "[igemm-prof] %-60s calls=%3d total_ms=%8.3f avg_us=%8.1f\n", kv.first.c_str(), kv.second.second, kv.second.first, 1000.0 * kv.second.first / kv.second.second);
    return 0;
}
