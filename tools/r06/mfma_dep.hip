// Round 6: issue cost of v_mfma_f32_32x32x16_f16 chains on ONE wave per SIMD: the same accumulator back to back (what linear_xs's one-block-per-wave
// loop does: 20 dependent MFMAs per weight stage) against 2 and 4 independent accumulators in rotation.  cycles per MFMA from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    h16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[NACC] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    for (int waves = 1; waves <= 2; ++waves) {      // 256-thread blocks: one wave per SIMD; grid 256 or 512 -> one / two waves per SIMD
        hipLaunchKernelGGL(k<1>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<2>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<4>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%d wave(s) per SIMD: cycles per MFMA  same accumulator %.1f | 2 in rotation %.1f | 4 in rotation %.1f   (s_memtime ticks = shader cycles)\n", waves,
               (double)h[1] / (iters * 16.0), (double)h[2] / (iters * 16.0), (double)h[4] / (iters * 16.0));
    }
    return 0;
}
