// Global -> LDS staging rate of the LDS-DMA path (buffer_load_dwordx4 ... lds) for several source access patterns, source resident in
// L2 / Infinity Cache ("Global->LDS staging rate").
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
// Every wave issues BURST 1-KiB DMA instructions (16 B per lane), waits for them (vmcnt(0)) and repeats; nothing else runs, so
// BURST KiB per wave are in flight (BURST = 2 / 4 / 8 / 16: the rate against the bytes in flight is the second table).  Patterns:
//   0: 1 KiB contiguous per instruction            1: 8 rows x 128 B, row stride 640 B   (igemm BK = 64 on K = 320)
//   2: 8 rows x 128 B, row stride 5760 B (K = 2880) 3: 4 rows x 256 B, row stride 5760 B  (a BK = 128 tile)
//   4: 16 rows x 64 B, row stride 5760 B (BK = 32)  5: pattern 2 with the igemm XOR swizzle on the 16-byte chunks
// Reported: bytes per shader clock per CU (s_memtime) and aggregate TB/s (HIP events), with 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int BURST>
__global__ __launch_bounds__(256, 2) void dma_loop(const char* src, unsigned span, int pattern, int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7FFFFFFF, 0x00020000);
    unsigned rel, step;                               // step: source advance between consecutive instructions (the next K tile)
    switch (pattern) {
        case 0: rel = lane * 16; step = 1024; break;
        case 1: rel = (lane >> 3) * 640 + (lane & 7) * 16; step = 128; break;
        case 2: rel = (lane >> 3) * 5760 + (lane & 7) * 16; step = 128; break;
        case 3: rel = (lane >> 4) * 5760 + (lane & 15) * 16; step = 256; break;
        case 4: rel = (lane >> 2) * 5760 + (lane & 3) * 16; step = 64; break;
        default: rel = (lane >> 3) * 5760 + (((lane & 7) ^ ((lane >> 4) & 7)) * 16); step = 128; break;
    }
    // every workgroup walks its own window of the source (windows overlap across workgroups: the data stays cache resident)
    unsigned base = (unsigned)(((size_t)blockIdx.x * 36864 + (size_t)wave * 9216) % span);
    char* dst = smem + wave * (BURST * 1024);
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < BURST; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + j * 1024), 16, rel + base + (unsigned)j * step, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        base += 46080;
        if (base >= span) base -= span;
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (lane == 0) out[(size_t)blockIdx.x * 4 + wave] = c1 - c0;
}

template <int BURST>
static void run(const char* src, unsigned span, int pattern, int blocks, int iters) {
    unsigned long long* d = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&d), (size_t)blocks * 4 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int smem = 4 * BURST * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_loop<BURST>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(dma_loop<BURST>, dim3(blocks), dim3(256), smem, 0, src, span, pattern, iters / 4, d);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(dma_loop<BURST>, dim3(blocks), dim3(256), smem, 0, src, span, pattern, iters, d);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double bytes_wave = (double)iters * BURST * 1024.0;
    const double per_cu = bytes_wave * 4.0 * (blocks > 256 ? 2.0 : 1.0) / (double)h[h.size() / 2];
    printf("pattern %d  blocks %3d  in flight per CU %3d KiB  %.3f ms  %6.2f TB/s aggregate  %6.1f B/clk/CU (median wave)\n", pattern, blocks,
           BURST * 4 * (blocks > 256 ? 2 : 1), ms, bytes_wave * 4.0 * blocks / ms / 1e9, per_cu);
    (void)hipFree(d);
}

int main() {
    const unsigned span = 24u << 20;                 // 24 MiB source: beyond one XCD's 4 MiB L2, inside the 256 MiB Infinity Cache
    char* src = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&src), (size_t)span + (1 << 20));
    (void)hipMemset(src, 1, (size_t)span + (1 << 20));
    for (int blocks : {256, 512})
        for (int p = 0; p < 6; ++p) run<16>(src, span, p, blocks, 2000);
    const unsigned small = 2u << 20;                 // 2 MiB source: L2 resident on every XCD
    for (int p = 0; p < 6; ++p) run<16>(src, small, p, 512, 2000);
    printf("-- rate against bytes in flight (pattern 2, 24 MiB source)\n");
    for (int blocks : {256, 512}) {
        run<2>(src, span, 2, blocks, 16000);
        run<4>(src, span, 2, blocks, 8000);
        run<8>(src, span, 2, blocks, 4000);
        run<16>(src, span, 2, blocks, 2000);
    }
    return 0;
}
