// Common device/host definitions for libladi_native (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LADI_WAVE 64

// ---------------------------------------------------------------------------------------------
// Activation codes shared by kernels
// ---------------------------------------------------------------------------------------------
enum {
    LADI_ACT_NONE = 0,
    LADI_ACT_SILU = 1,
    LADI_ACT_GELU = 2,   // exact erf GELU
    LADI_ACT_GEGLU = 3,  // igemm only: rows interleaved in blocks of 32 (u | g), out = u * gelu(g)
    LADI_ACT_RELU = 4,   // warping module (conv -> folded BatchNorm -> ReLU)
    LADI_ACT_TANH = 5    // small_linear only: control-point regression of the TPS network
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// x * sigmoid(x) with the hardware reciprocal (1 ulp) and exp2: 4 VALU instead of the ~14 of the IEEE division above; the result is
// rounded to fp16 by every caller
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of anything computed from it): one rcp, one
// exp2 and seven FMA-class operations instead of libm erff's ~40 VALU instructions
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(fmaf(-p, e, 1.0f), x);
}
// Exact (erf) GELU -- the GEGLU epilogues evaluate ~10^8 of them per UNet forward and are VALU-bound (linear_xs MODE 2: ~1 500 VALU cycles
// per 1 280 MFMA cycles of a 32 x 32 output block).  Round 6: written on the COMPLEMENTARY error function instead of 0.5 x (1 + erf):
//     gelu(x) = x Phi(x),  Phi(x) = 1 - Q(|x|) for x >= 0,  Q(|x|) for x < 0,  Q(a) = 0.5 erfc(a / sqrt 2) = (0.5 poly(t)) exp(-a^2 / 2)
//  => gelu(x) = max(x, 0) - |x| Q(|x|)
// with A&S 7.1.26's poly(t), t = 1 / (1 + p a / sqrt 2), the 0.5 and the 1 / sqrt 2 folded into the constants: 14 VALU operations instead
// of 17, and no 1 + erf cancellation in the negative tail (relative error 1.0 -> 3.6e-3 there; absolute 4.6e-7 -> 3.3e-7 over [-12, 12];
// against the float64 erf form the fp16-rounded results differ in 2.8 % of 2 M samples instead of 6.0 %).
__device__ __forceinline__ float gelu_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    p *= t;
    const float y = x * 0.84932180028801907f;            // sqrt(log2(e) / 2): exp(-x^2 / 2) = exp2(-y^2)
    const float q = p * __builtin_amdgcn_exp2f(-(y * y));
    return fmaf(-ax, q, fmaxf(x, 0.0f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: a process-wide `static bool` would leave the second GPU of a process
// with the 64 KB default and its first large-LDS launch failing.  `done` is the caller's static bit mask, one bit per device.
inline int ladi_ensure_dyn_lds(const void* kfn, int bytes, unsigned long long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -10;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return 0;
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -10;
    done |= bit;
    return 0;
}

// compile-time loop (keeps accumulator / fragment array indices constant so they stay in registers)
template <int V> struct IntC { static constexpr int value = V; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IntC<I>{}); static_for<I + 1, N>(f); }
}


// ---------------------------------------------------------------------------------------------
// Implicit-GEMM argument block (conv3x3 / conv1x1 / linear / batched GEMM), see igemm.hip
// ---------------------------------------------------------------------------------------------
struct IGemmArgs {
    // "pixel" operand: up to two NHWC fp16 sources concatenated along channels (C1 = 0 -> single)
    const h16* src0; const h16* src1;
    int C0, C1;          // channels taken from each source (multiples of 64)
    int ld0, ld1;        // row (pixel) stride of each source in elements
    int Hs, Ws;          // physical source spatial size (per sample)
    int Ho, Wo;          // output spatial size (per sample)
    int P;               // number of output pixels = n * Ho * Wo
    int ksize;           // 1 or 3
    int stride;          // 1 or 2
    int pad;             // leading pad (top/left); trailing pad is implied by bounds check
    int ups;             // 1: nearest-2x upsample folded into the gather (source is Hs x Ws, logical 2Hs x 2Ws)
    // "weight" operand: [Q][K] fp16 row-major, K = ksize*ksize*(C0+C1), tap-major / channel-minor
    const h16* W;
    int Q;               // output channels (rows of W)
    int K;
    int ldw;             // row stride of W in elements (0 -> K)
    // batching over grid.z (element strides); 0 = shared
    long long bs_src0, bs_w, bs_out, bs_res;
    // epilogue
    const h16* bias;     // [Q] or null (GEGLU: interleaved like W rows)
    int bias_per_pixel;  // 1: bias indexed by pixel (row-bias for transposed products)
    const float* rowadd; // [Q] fp32 added per output channel (time embedding) or null
    const int* rowadd_idx; int rowadd_stride; // optional device-side row selector: rowadd + (*idx)*stride
    int act;
    float out_scale;
    const h16* res0; const h16* res1; // residual tensors [P][ldr] or null
    int ldr0, ldr1;
    const h16* mask;     // [P] fp16; out *= (1 - mask[p]) or null
    void* out; int ldo;  // [P][ldo]
    int out_f32;         // 1: store fp32
    float* stats;        // optional per-channel partial statistics of the OUTPUT: rows of [Q][2] (sum, sumsq), see kernels.h
    int stats_groups;    // unused
    int splitk;          // set by the launcher: > 1 = grid.z slices K, fp32 partials to `out` (+ z*bs_out), reduced by a 2nd kernel
    int tile_map;        // set by the launcher: 0 plain, 1 pixel tiles split over XCDs, 2 channel tiles split over XCDs
    // optional LayerNorm of the pixel operand: x <- (x - mean) * rstd * gamma + beta over the C0 channels of every pixel, rounded to
    // fp16 (null = none).  The X-stationary linear kernel fuses it into its prologue (the wave holds whole rows in registers); every
    // other configuration runs the stand-alone LayerNorm kernel into ln_scratch ([P][C0] fp16) first -- the tuner times both forms
    const h16* ln_gamma; const h16* ln_beta;
    float ln_eps;
    float bias_mul;      // multiplier of `bias` (0 means 1): the VAE keeps its residual stream scaled by 2^-k when the fp16 range is tight
                         // (runtime_vae.cpp); a convolution whose INPUT is the scaled stream then needs bias * 2^-k
    h16* ln_scratch;
    // in-launch split-K combine (set by the launcher, igemm_common.h igemm_splitk_combine): fp32 slabs [splitk][tiles][workgroup image] and
    // the per-tile arrival counters (zero on entry, left zero); sk_cnt == nullptr with splitk > 1 = two-pass form (splitk_reduce_kernel)
    float* sk_ws; int* sk_cnt;
    // optional GroupNorm affine of the pixel operand (no activation): x <- x * scale + shift per (sample, channel), rounded to fp16;
    // gn_ss = [n][C0][2] floats (scale, shift) as gn_finalize writes them, gn_hw = pixels per sample (a multiple of 32).  Only the
    // X-stationary linear kernel implements it (the launcher offers no other configuration when it is set)
    const float* gn_ss; int gn_hw;
};
