// Halo-resident 3x3 convolution (round 3): eligibility tests and the dispatch to the sixteen forms of igemm_halo_kernel.h, which are compiled in
// the units igemm_halo_inst_{a..h}.hip.
#include "common.h"
#include "kernels.h"
#include <algorithm>

namespace { constexpr int HALO_WMAX = 48; }

int ladi_halo_launch_f128x256_d(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f256x256(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f320x256(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f128x128_d(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f256x128(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f128x128_w2(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f128x192_w2(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f128x128_w2n(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f128x192_w2n(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f320x192_w6(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f320x192_one(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_g128x256(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_g256x256(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_g320x256(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_g128x128_w2(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_f128x256_w2n(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_u128x192(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_u128x128(IGemmArgs a, int batch, hipStream_t st);
int ladi_halo_launch_u320x192_w6(IGemmArgs a, int batch, hipStream_t st);

bool ladi_igemm_halo_eligible(const IGemmArgs& a, int batch) {
    return a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.ups && a.Ws <= HALO_WMAX && a.Ho == a.Hs && a.Wo == a.Ws && !(a.C0 % 64) && !(a.C1 % 64) &&
           batch == 1 && (size_t)a.P * (size_t)std::max(a.ld0, a.ld1) * 2 < 0x7FFFFFFFull;
}

// folded-upsample form (round 6): nearest-2x upsample + 3x3 convolution, single source, output rows of at most 48 pixels, whole `bp`-pixel tiles
// inside a sample
bool ladi_igemm_halo_ups_eligible(const IGemmArgs& a, int batch, int bp) {
    return a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.ups == 1 && a.Ho == 2 * a.Hs && a.Wo == 2 * a.Ws && a.Wo <= HALO_WMAX && !a.C1 && !a.src1 &&
           !(a.C0 % 64) && batch == 1 && bp > 0 && !((a.Ho * a.Wo) % bp) && !(a.P % (a.Ho * a.Wo)) && (size_t)a.P * (size_t)a.ld0 * 2 < 0x7FFFFFFFull;
}

// 2-D blocked form: rows of any width that is a multiple of 32, whole blocks of `th` image rows
bool ladi_igemm_halo2d_eligible(const IGemmArgs& a, int batch, int th) {
    return a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.ups && a.Ho == a.Hs && a.Wo == a.Ws && !(a.C0 % 64) && !(a.C1 % 64) && batch == 1 &&
           !(a.Ws % 32) && th > 0 && !(a.Hs % th) && (size_t)a.P * (size_t)std::max(a.ld0, a.ld1) * 2 < 0x7FFFFFFFull;
}

// (tq, tp): wave tile in 32-blocks on the 2 x 4 wave grid -> workgroup tile (64 tq) x (128 tp); nxb: halo buffers (10: the 4-wave form,
// 2 x 2 waves -> (64 tq) x (64 tp), one halo buffer)
int ladi_launch_igemm_halo(const IGemmArgs& a, int tq, int tp, int nxb, int batch, hipStream_t st) {
    if (tq == 2 && tp == 2 && nxb == 2) return ladi_halo_launch_f128x256_d(a, batch, st);   // 128x256, 3 weight slots (144 KB)
    if (tq == 4 && tp == 2 && nxb == 1) return ladi_halo_launch_f256x256(a, batch, st);   // 256x256, single halo buffer, 3 weight slots (144 KB)
    if (tq == 5 && tp == 2 && nxb == 1) return ladi_halo_launch_f320x256(a, batch, st);   // 320x256, single halo buffer, 2 weight slots (128 KB)
    if (tq == 2 && tp == 1 && nxb == 2) return ladi_halo_launch_f128x128_d(a, batch, st);   // 128x128, 4 weight slots (128 KB)
    if (tq == 4 && tp == 1 && nxb == 1) return ladi_halo_launch_f256x128(a, batch, st);   // 256x128, single halo buffer, 3 weight slots (128 KB)
    // 4-wave workgroups, TWO per CU (64-72 KB each): their barriers are independent, so one workgroup multiplies while the other waits
    if (tq == 2 && tp == 2 && nxb == 10) return ladi_halo_launch_f128x128_w2(a, batch, st);  // 128x128
    if (tq == 2 && tp == 3 && nxb == 10) return ladi_halo_launch_f128x192_w2(a, batch, st);  // 128x192
    // round 4
    if (tq == 2 && tp == 2 && nxb == 11) return ladi_halo_launch_f128x128_w2n(a, batch, st);   // 128x128, 4 waves x 2 per CU, rows <= 24 pixels: 3 weight slots (72 KB)
    if (tq == 2 && tp == 3 && nxb == 11) return ladi_halo_launch_f128x192_w2n(a, batch, st);   // 128x192, same
    if (tq == 5 && tp == 1 && nxb == 12) return ladi_halo_launch_f320x192_w6(a, batch, st);       // 320x192, 12 waves (3 per SIMD), 144 KB
    // round 5
    if (tq == 5 && tp == 3 && nxb == 13) return ladi_halo_launch_f320x192_one(a, batch, st);   // 320x192, 4 waves, ONE per SIMD (240 accumulators), 120 KB
    // round 5: 2-D blocked halo tiles (nxb 20 + ...): images wider than 48 pixels
    if (tq == 2 && tp == 2 && nxb == 20) return ladi_halo_launch_g128x256(a, batch, st);   // 128x256 (8 rows x 32), 8 waves, 96 KB
    if (tq == 4 && tp == 2 && nxb == 20) return ladi_halo_launch_g256x256(a, batch, st);   // 256x256, 8 waves, 144 KB
    if (tq == 5 && tp == 2 && nxb == 20) return ladi_halo_launch_g320x256(a, batch, st);   // 320x256, 8 waves, 128 KB
    if (tq == 2 && tp == 2 && nxb == 21) return ladi_halo_launch_g128x128_w2(a, batch, st);   // 128x128 (4 rows x 32), 4 waves x 2 per CU, 60 KB
    if (tq == 2 && tp == 4 && nxb == 11) return ladi_halo_launch_f128x256_w2n(a, batch, st);      // 128x256, 4 waves (64 x 128 each: 0.75 KB of fragment reads per MFMA, half the weight DMA per MFMA of 128x128) x 2 per CU, rows <= 24 pixels (72 KB)
    // round 6: folded-upsample forms (nxb 30: 4 waves x 2 per CU; 31: 12 waves)
    if (tq == 2 && tp == 3 && nxb == 30) return ladi_halo_launch_u128x192(a, batch, st);
    if (tq == 2 && tp == 2 && nxb == 30) return ladi_halo_launch_u128x128(a, batch, st);
    if (tq == 5 && tp == 1 && nxb == 31) return ladi_halo_launch_u320x192_w6(a, batch, st);
    return -7;
}
