// The ring-staged implicit-GEMM tile shapes that are compiled into the library: X(base id, WQ, WP, TQ, TP, BK, NST, OCC, ILV), see
// igemm_kernel.h for the meaning of the parameters.  Grouped by translation unit (igemm_inst_<group>.hip) so the build parallelises;
// igemm.hip turns the same list into its dispatch table.
#pragma once

// ---- two workgroups (of 4 waves) or more per CU: the round-1/2 shapes
#define LADI_IGEMM_TILES_A(X) \
    X(1, 2, 2, 2, 4, 32, 3, 2, 0)  /* 128x256 */ \
    X(3, 2, 2, 2, 2, 32, 3, 2, 0)  /* 128x128 */ \
    X(4, 2, 2, 2, 1, 32, 3, 2, 0)  /* 128x64  */ \
    X(5, 2, 2, 1, 1, 32, 3, 2, 0)  /* 64x64   */
#define LADI_IGEMM_TILES_B(X) \
    X(2, 2, 2, 5, 2, 32, 2, 2, 0)  /* 320x128 (Cout = 320 layers, no padding waste) */ \
    X(6, 2, 2, 4, 2, 32, 3, 2, 0)  /* 256x128 */ \
    X(16, 2, 2, 1, 1, 32, 4, 2, 0) /* 64x64, deeper prefetch for shallow-K, latency-bound GEMMs */
#define LADI_IGEMM_TILES_C(X) \
    X(7, 2, 2, 2, 2, 64, 2, 2, 0)  /* 128x128 BK64 */ \
    X(8, 2, 2, 2, 4, 64, 2, 2, 0)  /* 128x256 BK64 */ \
    X(9, 2, 2, 2, 1, 64, 3, 2, 0)  /* 128x64  BK64 */
#define LADI_IGEMM_TILES_D(X) \
    X(10, 2, 2, 5, 2, 64, 2, 2, 0) /* 320x128 BK64 */ \
    X(17, 2, 2, 2, 1, 32, 4, 2, 0) \
    X(18, 2, 2, 2, 2, 32, 4, 2, 0)
#define LADI_IGEMM_TILES_E(X) \
    X(19, 2, 4, 2, 2, 32, 3, 2, 0) /* 128x256, 8 waves */ \
    X(20, 4, 2, 2, 2, 32, 3, 2, 0) /* 256x128, 8 waves */ \
    X(21, 2, 4, 4, 2, 32, 3, 2, 0) /* 256x256, 8 waves, 96 KB LDS */
#define LADI_IGEMM_TILES_F(X) \
    X(22, 2, 4, 5, 2, 64, 2, 2, 0) /* 320x256, 8 waves, 144 KB LDS */ \
    X(47, 2, 2, 2, 2, 64, 2, 2, 1) /* 128x128 BK64, DMA issue interleaved with the MFMA groups */ \
    X(48, 2, 2, 2, 1, 64, 3, 2, 1) /* 128x64  BK64, interleaved */
// ---- round 3: ONE workgroup of 4 waves per CU, one wave per SIMD with up to 240 accumulator registers: tiles whose grid fills the chip
//      at batch 8 (320x192: 256 tiles on the 49 152-pixel level) and deep rings (64-128 KB in flight per CU)
#define LADI_IGEMM_TILES_G(X) \
    X(39, 2, 2, 5, 3, 64, 2, 1, 1) /* 320x192, 128 KB ring, interleaved */ \
    X(40, 2, 2, 5, 3, 32, 4, 1, 1) /* 320x192, BK32 4-deep ring (128 KB): every stage is issued 3 K steps before it is awaited */
#define LADI_IGEMM_TILES_H(X) \
    X(41, 2, 2, 4, 4, 64, 2, 1, 1) /* 256x256, 128 KB ring */ \
    X(42, 2, 2, 2, 2, 64, 4, 1, 1) /* 128x128, 4-deep ring (128 KB) */
#define LADI_IGEMM_TILES_I(X) \
    X(43, 2, 2, 2, 4, 64, 3, 1, 1) /* 128x256, 3-deep ring (144 KB) */ \
    X(44, 2, 2, 2, 1, 64, 5, 1, 1) /* 128x64, 5-deep ring (120 KB) */ \
    X(45, 2, 2, 4, 3, 64, 2, 1, 1) /* 256x192, 112 KB ring */

#define LADI_IGEMM_TILES_ALL(X) \
    LADI_IGEMM_TILES_A(X) LADI_IGEMM_TILES_B(X) LADI_IGEMM_TILES_C(X) LADI_IGEMM_TILES_D(X) LADI_IGEMM_TILES_E(X) LADI_IGEMM_TILES_F(X) \
    LADI_IGEMM_TILES_G(X) LADI_IGEMM_TILES_H(X) LADI_IGEMM_TILES_I(X)
