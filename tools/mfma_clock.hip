// Sustained MFMA rate and shader clock under matrix load ("Sustained MFMA rate").
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
// Every wave runs ITERS x 8 independent v_mfma_f32_32x32x16_f16 (nothing else in the loop) and records the shader-clock counter
// (s_memtime) and the constant 100 MHz wall counter (s_memrealtime) around it: clock = d(cycles) / d(wall).  Reported for a full chip
// (256 CUs x 2 workgroups x 4 waves = 2 waves per SIMD, like the igemm8 tiles) and for a single workgroup (idle chip).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void mfma_loop(int iters, unsigned long long* out, float* sink) {
    h16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter();   // s_memtime: shader clock
    const unsigned long long w0 = wall_clock64();                 // s_memrealtime: constant 100 MHz
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][threadIdx.x & 15];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = w1 - w0;
    }
}

static void run(int blocks, int iters, const char* label) {
    unsigned long long* d = nullptr;
    float* sink = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&d), (size_t)blocks * 4 * 2 * sizeof(unsigned long long));
    (void)hipMalloc(reinterpret_cast<void**>(&sink), 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, iters / 8, d, sink);   // warm-up
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, iters, d, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 8);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> mhz, cyc_per_mfma;
    for (int w = 0; w < blocks * 4; ++w) {
        const double cyc = (double)h[2 * w], wall = (double)h[2 * w + 1];
        if (wall > 0) mhz.push_back(cyc / wall * 100.0);
        cyc_per_mfma.push_back(cyc / ((double)iters * 8.0));
    }
    std::sort(mhz.begin(), mhz.end());
    std::sort(cyc_per_mfma.begin(), cyc_per_mfma.end());
    const double flop = (double)blocks * 4 * (double)iters * 8.0 * 32768.0;
    printf("%-28s blocks=%4d  kernel %.3f ms  %.1f TFLOP/s  shader clock median %.0f MHz (min %.0f, max %.0f)  cycles per MFMA per wave median %.1f\n",
           label, blocks, ms, flop / ms / 1e9, mhz[mhz.size() / 2], mhz.front(), mhz.back(), cyc_per_mfma[cyc_per_mfma.size() / 2]);
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    run(1, 40000, "one workgroup (idle chip)");
    run(512, 40000, "full chip, 2 waves per SIMD");
    run(512, 200000, "full chip, 5x longer");
    run(256, 40000, "full chip, 1 wave per SIMD");
    return 0;
}
