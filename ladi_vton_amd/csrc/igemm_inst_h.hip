// Instantiation unit of the ring-staged implicit-GEMM kernel: tile group H of igemm_tiles.h.
#include "igemm_kernel.h"
#include "igemm_tiles.h"
#define X(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV) LADI_IGEMM_INSTANTIATE(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV)
LADI_IGEMM_TILES_H(X)
#undef X
