// Instantiation unit of the ring-staged implicit-GEMM kernel: tile group A of igemm_tiles.h.
#include "igemm_kernel.h"
#include "igemm_tiles.h"
#define X(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV) LADI_IGEMM_INSTANTIATE(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV)
LADI_IGEMM_TILES_A(X)
#undef X
