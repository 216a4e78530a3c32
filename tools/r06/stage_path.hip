// Round 6: the two ways of getting a K tile from global memory into LDS, under MFMA load.  Every GEMM-family kernel of this library stages through
// the LDS-DMA path (buffer_load_dwordx4 ... lds) and DESIGN.md section 3 found "time = t_DMA + t_MFMA": a DMA instruction holds its wave and the wave's
// MFMAs queue behind it.  hipBLASLt on the same box runs the 16x12-level GEMMs 10-27 % faster (profiles/r06_vendor_gemm_shapes.txt), so the question is
// whether the classic path -- global_load_dwordx4 into registers, ds_write_b128 one step later -- overlaps with the MFMAs where the DMA path does not.
// One K loop of a BQ x BP x 64 tile (4 waves, 2 x 2, wave tile BQ/2 x BP/2, v_mfma_f32_32x32x16_f16, swizzled 128-byte rows, one barrier per step):
//   mode 0  fragments + MFMAs only (no staging)            mode 1  LDS-DMA staging only            mode 2  LDS-DMA ring (depth D) + MFMAs
//   mode 5  weights DIRECT to the MFMA operand registers from a fragment-packed layout (no LDS for them), pixels through the LDS-DMA ring + MFMAs;  mode 6  same with the pixels resident (no staging, no barrier)
//   mode 3  register staging only (loads, ds_write)        mode 4  register staging + MFMAs (loads of step s+2 and ds_writes of step s+1 issued ahead of the MFMAs of step s)
// Results are timing only (operands are whatever the buffers hold).   hipcc --offload-arch=gfx950 -O3 tools/r06/stage_path.hip -o tools/r06/bin/stage_path
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N> __device__ __forceinline__ void wait_vm() {       // s_waitcnt vmcnt(N), lgkmcnt / expcnt untouched (gfx9 encoding: vmcnt low 4 bits [3:0], high 2 bits [15:14])
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
// one stage's 1-KiB DMA instructions of this wave (a free function: inside a lambda of the kernel the builtin makes hipcc's host pass drop the kernel's stub silently)
template <int BQ, int NI>
__device__ __forceinline__ void dma_stage(const char* Wm, const char* Xm, char* dst, int wave, const unsigned* goff, unsigned koff) {
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wm), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Xm), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const bool isw = (wave * NI + j) * 8 < BQ;          // wave-uniform
        if (isw) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(dst + j * 1024), 16, goff[j] + koff, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(dst + j * 1024), 16, goff[j] + koff, 0, 0, 0);
    }
}
template <int MODE, int BQ, int BP, int D, int WPC, int WQN = 2>
__global__ __launch_bounds__(256, WPC) void kloop(const char* __restrict__ Wm, const char* __restrict__ Xm, int K, int nbq, int xmap, int passes, float* out,
                                                   unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = BQ + BP;                      // rows of 128 bytes per stage
    constexpr int STAGE = ROWS * 128;
    constexpr int NI = ROWS / 32;                      // 1-KiB wave-instructions per wave per stage (8 rows each, 4 waves)
    constexpr int WPN_ = 4 / WQN;                      // waves along the pixel direction (WQN along the channel direction: 2 x 2, or 4 x 1)
    constexpr int TQ = BQ / (32 * WQN), TP = BP / (32 * WPN_);   // 32x32 MFMA tiles per wave in each direction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave % WQN, wp = wave / WQN;
    // xmap: the library's tile_map 1 -- block b runs on XCD b % 8; the pixel tiles are split over the XCDs (each XCD's L2 holds all of W and 1/8 of X), q fastest inside an XCD
    int bq = blockIdx.x % nbq, bp = blockIdx.x / nbq;
    if (xmap) { const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3, nbp8 = (int)(gridDim.x / nbq) >> 3; bq = i % nbq; bp = xcd * nbp8 + i / nbq; }
    const unsigned rowb = (unsigned)K * 2u;            // bytes per source row
    // source offsets of this lane's 16-byte pieces: piece (j, lane) of wave w covers stage row r = (w * NI + j) * 8 + (lane >> 3), chunk (lane & 7) ^ (r & 7)
    unsigned goff[NI];
    const char* gsrc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int r = (wave * NI + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        const bool isw = r < BQ;
        const unsigned row = isw ? (unsigned)(bq * BQ + r) : (unsigned)(bp * BP + r - BQ);
        goff[j] = row * rowb + (unsigned)c * 16u;
        gsrc[j] = isw ? Wm : Xm;
    }
    // fragment addresses: A (weights) rows wq*BQ/2 + i*32 + (lane & 31); B (pixels) rows BQ + wp*BP/2 + i*32 + (lane & 31); chunk (kk*2 + (lane >> 5)) ^ (row & 7)
    unsigned fa[TQ], fb[TP];
#pragma unroll
    for (int i = 0; i < TQ; ++i) fa[i] = (unsigned)(wq * (BQ / WQN) + i * 32 + (lane & 31));
#pragma unroll
    for (int i = 0; i < TP; ++i) fb[i] = (unsigned)(BQ + wp * (BP / WPN_) + i * 32 + (lane & 31));
    f32x16 acc[TQ][TP];
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int steps = K / 64;

    auto compute = [&](int buf) {
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            h16x8 a[TQ], b[TP];
#pragma unroll
            for (int i = 0; i < TQ; ++i) a[i] = *reinterpret_cast<const h16x8*>(base + fa[i] * 128u + ((unsigned)((kk * 2 + (lane >> 5)) ^ (fa[i] & 7)) * 16u));
#pragma unroll
            for (int i = 0; i < TP; ++i) b[i] = *reinterpret_cast<const h16x8*>(base + fb[i] * 128u + ((unsigned)((kk * 2 + (lane >> 5)) ^ (fb[i] & 7)) * 16u));
#pragma unroll
            for (int i = 0; i < TQ; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
#define dma_issue(s_, buf_) dma_stage<BQ, NI>(Wm, Xm, smem + (buf_) * STAGE + wave * (NI * 1024), wave, goff, (unsigned)(s_) * 128u)
    f32x4 regs[NI];
    auto reg_load = [&](int s) {
#pragma unroll
        for (int j = 0; j < NI; ++j) regs[j] = *reinterpret_cast<const f32x4*>(gsrc[j] + goff[j] + (unsigned)s * 128u);
    };
    auto reg_store = [&](int buf) {
        char* dst = smem + buf * STAGE + wave * (NI * 1024) + lane * 16;
#pragma unroll
        for (int j = 0; j < NI; ++j) *reinterpret_cast<f32x4*>(dst + j * 1024) = regs[j];
    };

    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int pass = 0; pass < passes; ++pass) {
        if (MODE == 0) {
            for (int s = 0; s < steps; ++s) { __builtin_amdgcn_s_barrier(); compute(s % D); }
        } else if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int s = 0; s < D - 1; ++s) dma_issue(s, s);
            for (int s = 0; s < steps; ++s) {
                // stage s landed: at most (D - 2) younger stages may stay outstanding
                if (D == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (D == 3) { if (s + 1 < steps) wait_vm<NI>(); else wait_vm<0>(); }
                else { if (s + 2 < steps) wait_vm<2 * NI>(); else wait_vm<0>(); }
                __builtin_amdgcn_s_barrier();
                if (s + D - 1 < steps) dma_issue(s + D - 1, (s + D - 1) % D);
                if (MODE == 2) compute(s % D);
            }
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 5 || MODE == 6) {
            // The weights never touch LDS: packed in global memory in MFMA-fragment order ([q / 32][k / 16][lane][16 B]), ONE coalesced 1-KiB load per 32 x 16 fragment
            // straight into the MFMA operand registers, requested one step ahead (two register sets, the loop unrolled by two so that no copy is needed).
            // MODE 5: the pixels go through the LDS-DMA ring (depth 2).  MODE 6: the pixels are resident in LDS (the halo kernel between two chunk boundaries): no staging, no barrier.
            constexpr int NIX = BP / 32;
            const char* wf = Wm + (size_t)((bq * (BQ / 32) + wq * TQ) * (K / 16)) * 1024 + lane * 16;
            unsigned gx[NIX];
#pragma unroll
            for (int j = 0; j < NIX; ++j) {
                const int r = (wave * NIX + j) * 8 + (lane >> 3);
                gx[j] = (unsigned)(bp * BP + r) * rowb + (unsigned)((lane & 7) ^ (r & 7)) * 16u;
            }
            h16x8 A0[TQ][4], A1[TQ][4];
            auto a_load = [&](int s_, h16x8 (&dst)[TQ][4]) {
#pragma unroll
                for (int i = 0; i < TQ; ++i)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        // xmap >= 2: the (chunk, tap) walk of a 3x3 convolution over tap-major storage -- step s reads k16 blocks ((s % 9) * K / 9 + (s / 9) * 64) / 16 ..+3
                        const int kb0 = xmap >= 2 ? (((s_ % 9) * (K / 9) + (s_ / 9) * 64) >> 4) : s_ * 4;
                        dst[i][kk] = *reinterpret_cast<const h16x8*>(wf + ((size_t)i * (K / 16) + (size_t)(kb0 + kk)) * 1024);
                    }
            };
            auto compute_a = [&](int buf, h16x8 (&av)[TQ][4]) {
                const char* base = smem + buf * STAGE;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    h16x8 b[TP];
#pragma unroll
                    for (int i = 0; i < TP; ++i) b[i] = *reinterpret_cast<const h16x8*>(base + fb[i] * 128u + ((unsigned)((kk * 2 + (lane >> 5)) ^ (fb[i] & 7)) * 16u));
#pragma unroll
                    for (int i = 0; i < TQ; ++i)
#pragma unroll
                        for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i][kk], b[j], acc[i][j], 0, 0, 0);
                }
            };
#define x_issue(s_, buf_) dma_stage<0, NIX>(Wm, Xm, smem + (buf_) * STAGE + BQ * 128 + wave * (NIX * 1024), wave, gx, (unsigned)(s_) * 128u)
            auto step = [&](int s_, h16x8 (&cur)[TQ][4], h16x8 (&nxt)[TQ][4]) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // weights of step s_ (and, MODE 5, the pixel stage) requested one step ago
                if (MODE == 5) __builtin_amdgcn_s_barrier();
                // unconditional (the last step re-requests its own operands): behind a branch the compiler's own s_waitcnt pass must assume the loads were NOT issued and
                // makes the MFMAs of this step wait for the loads of the next one
                const int sn = s_ + 1 < steps ? s_ + 1 : s_;
                if (MODE == 5) x_issue(sn, (s_ + 1) & 1);
                a_load(sn, nxt);
                __builtin_amdgcn_sched_barrier(0);                            // keep the requests AHEAD of this step's MFMAs (hipcc otherwise sinks them to their first use: one register set, no prefetch)
                compute_a(MODE == 5 ? (s_ & 1) : 0, cur);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (MODE == 5) x_issue(0, 0);
            a_load(0, A0);
            for (int s = 0; s < steps; s += 2) {
                step(s, A0, A1);
                step(s + 1, A1, A0);                                          // steps is even here
            }
            if (MODE == 5) __builtin_amdgcn_s_barrier();
        } else {
            reg_load(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            reg_store(0);
            if (steps > 1) reg_load(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int s = 0; s < steps; ++s) {
                if (s + 1 < steps) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stage s + 1 is in the registers (requested one step ago)
                    reg_store((s + 1) & 1);                               // buffer (s + 1) & 1 was last read in step s - 1: free since the barrier
                    if (s + 2 < steps) reg_load(s + 2);
                }
                if (MODE == 4) compute(s & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    if (MODE == 3) for (int j = 0; j < NI; ++j) sum += regs[j][0];
    if (sum == 123.456f) out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int BQ, int BP, int D, int WPC, int WQN = 2>
static void run(const char* W, const char* X, int Q, int P, int K, float* out, unsigned long long* cyc, const char* what, int xmap = 1) {
    const int nbq = Q / BQ, nbp = P / BP, blocks = nbq * nbp, passes = 20;
    const int smem = D * (BQ + BP) * 128;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kloop<MODE, BQ, BP, D, WPC, WQN>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kloop<MODE, BQ, BP, D, WPC, WQN>), dim3(blocks), dim3(256), smem, 0, W, X, K, nbq, xmap, 2, out, cyc);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kloop<MODE, BQ, BP, D, WPC, WQN>), dim3(blocks), dim3(256), smem, 0, W, X, K, nbq, xmap, passes, out, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed: %s\n", what); return; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double steps = (double)passes * (K / 64);
    const double us_pass = ms * 1e3 / passes;
    const double flop = 2.0 * Q * P * K;
    printf("%-34s map %d waves %dx%d %3dx%-3d D=%d wg/CU<=%d blocks %4d  %7.1f us per K loop  %6.0f TFLOP/s-equivalent  %6.0f cycles per step (median wg, s_memtime)  LDS %3d KB\n", what, xmap, WQN, 4 / WQN, BQ, BP, D,
           WPC, blocks, us_pass, (MODE == 0 || MODE == 2 || MODE >= 4) ? flop / us_pass / 1e6 : 0.0, (double)h[blocks / 2] / steps, smem / 1024);
}

int main() {
    const int K = 1280, Qmax = 3840, Pmax = 12288, KBIG = 5760;
    char *W = nullptr, *X = nullptr;
    float* out = nullptr;
    unsigned long long* cyc = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&W), (size_t)Qmax * KBIG * 2 + 4096);
    (void)hipMalloc(reinterpret_cast<void**>(&X), (size_t)Pmax * KBIG * 2 + 4096);
    (void)hipMemset(W, 0, (size_t)Qmax * KBIG * 2 + 4096);
    (void)hipMemset(X, 0, (size_t)Pmax * KBIG * 2 + 4096);
    (void)hipMalloc(reinterpret_cast<void**>(&out), 4096 * 256 * 4);
    (void)hipMalloc(reinterpret_cast<void**>(&cyc), 4096 * 8);
    printf("== 3072 x 1280 x 1280 (the 16x12 projection: 240 tiles of 128x128, one workgroup per CU)\n");
    run<0, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "fragments + MFMA only");
    run<1, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA staging only");
    run<1, 128, 128, 3, 1>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA staging only");
    run<2, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA ring + MFMA");
    run<2, 128, 128, 3, 1>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA ring + MFMA");
    run<2, 128, 128, 4, 1>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA ring + MFMA");
    run<3, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "register staging only");
    run<4, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "register staging + MFMA");
    run<5, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "W direct-to-VGPR, X ring + MFMA");
    run<6, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "W direct-to-VGPR, X resident");
    printf("-- same, blocks in plain order (every XCD touches all of W and X: 11 MB against a 4 MB L2)\n");
    run<1, 128, 128, 3, 1>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA staging only", 0);
    run<2, 128, 128, 3, 1>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA ring + MFMA", 0);
    run<3, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "register staging only", 0);
    run<4, 128, 128, 2, 1>(W, X, 1280, 3072, K, out, cyc, "register staging + MFMA", 0);
    printf("== 6144 x 1280 x 1280 (480 tiles of 128x128, two workgroups per CU)\n");
    run<0, 128, 128, 2, 2>(W, X, 1280, 6144, K, out, cyc, "fragments + MFMA only");
    run<1, 128, 128, 2, 2>(W, X, 1280, 6144, K, out, cyc, "LDS-DMA staging only");
    run<2, 128, 128, 2, 2>(W, X, 1280, 6144, K, out, cyc, "LDS-DMA ring + MFMA");
    run<3, 128, 128, 2, 2>(W, X, 1280, 6144, K, out, cyc, "register staging only");
    run<4, 128, 128, 2, 2>(W, X, 1280, 6144, K, out, cyc, "register staging + MFMA");
    run<5, 128, 128, 2, 2>(W, X, 1280, 6144, K, out, cyc, "W direct-to-VGPR, X ring + MFMA");
    run<6, 128, 128, 2, 2>(W, X, 1280, 6144, K, out, cyc, "W direct-to-VGPR, X resident");
    printf("== 3072 x 1280 x 1280 as 128 x 64 tiles (480 tiles, two workgroups per CU: what the tile table picks for this shape)\n");
    run<0, 128, 64, 2, 2>(W, X, 1280, 3072, K, out, cyc, "fragments + MFMA only");
    run<1, 128, 64, 3, 2>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA staging only");
    run<2, 128, 64, 3, 2>(W, X, 1280, 3072, K, out, cyc, "LDS-DMA ring + MFMA");
    run<3, 128, 64, 2, 2>(W, X, 1280, 3072, K, out, cyc, "register staging only");
    run<4, 128, 64, 2, 2>(W, X, 1280, 3072, K, out, cyc, "register staging + MFMA");
    run<5, 128, 64, 2, 2>(W, X, 1280, 3072, K, out, cyc, "W direct-to-VGPR, X ring + MFMA");
    run<6, 128, 64, 2, 2>(W, X, 1280, 3072, K, out, cyc, "W direct-to-VGPR, X resident");
    printf("== 3072 x 3840 x 1280 as 128 x 128 tiles (720 tiles, two workgroups per CU) -- the fused q/k/v of the 16x12 level\n");
    run<0, 128, 128, 2, 2>(W, X, 3840, 3072, K, out, cyc, "fragments + MFMA only");
    run<2, 128, 128, 2, 2>(W, X, 3840, 3072, K, out, cyc, "LDS-DMA ring + MFMA");
    run<4, 128, 128, 2, 2>(W, X, 3840, 3072, K, out, cyc, "register staging + MFMA");
    run<5, 128, 128, 2, 2>(W, X, 3840, 3072, K, out, cyc, "W direct-to-VGPR, X ring + MFMA");
    run<6, 128, 128, 2, 2>(W, X, 3840, 3072, K, out, cyc, "W direct-to-VGPR, X resident");
    printf("== 12288 x 640 x 5760-like K loop (K = 1280 here): 480 tiles of 128x128 at two workgroups per CU is the dominant halo form's grid\n");
    run<0, 128, 128, 2, 2>(W, X, 640, 12288, K, out, cyc, "fragments + MFMA only");
    run<2, 128, 128, 2, 2>(W, X, 640, 12288, K, out, cyc, "LDS-DMA ring + MFMA");
    run<6, 128, 128, 2, 2>(W, X, 640, 12288, K, out, cyc, "W direct-to-VGPR, X resident");
    printf("-- the same grid with the four waves as 4 x 1 (wave tile 32 channels x 128 pixels: every weight fragment is fetched by ONE wave, every pixel fragment read by all four)\n");
    run<0, 128, 128, 2, 2, 4>(W, X, 640, 12288, K, out, cyc, "fragments + MFMA only");
    run<2, 128, 128, 2, 2, 4>(W, X, 640, 12288, K, out, cyc, "LDS-DMA ring + MFMA");
    run<6, 128, 128, 2, 2, 4>(W, X, 640, 12288, K, out, cyc, "W direct-to-VGPR, X resident");
    run<5, 128, 128, 2, 2, 4>(W, X, 640, 12288, K, out, cyc, "W direct-to-VGPR, X ring + MFMA");
    printf("-- 256 x 128 tiles, 4 x 1 waves of 64 x 128 (128 accumulators; 240 tiles, one workgroup per CU)\n");
    run<0, 256, 128, 2, 1, 4>(W, X, 640 * 2, 6144, K, out, cyc, "fragments + MFMA only");
    run<2, 256, 128, 2, 1, 4>(W, X, 640 * 2, 6144, K, out, cyc, "LDS-DMA ring + MFMA");
    run<6, 256, 128, 2, 1, 4>(W, X, 640 * 2, 6144, K, out, cyc, "W direct-to-VGPR, X resident");
    printf("-- K = 5760 (the real reduction length of conv 640 -> 640: 7.4 MB of weights against a 4 MB L2), 128 x 128 tiles, two workgroups per CU\n");
    run<2, 128, 128, 2, 2>(W, X, 640, 12288, KBIG, out, cyc, "LDS-DMA ring + MFMA");
    run<2, 128, 128, 2, 2, 4>(W, X, 640, 12288, KBIG, out, cyc, "LDS-DMA ring + MFMA");
    run<6, 128, 128, 2, 2>(W, X, 640, 12288, KBIG, out, cyc, "W direct-to-VGPR, X resident");
    run<6, 128, 128, 2, 2, 4>(W, X, 640, 12288, KBIG, out, cyc, "W direct-to-VGPR, X resident");
    run<0, 128, 128, 2, 2, 4>(W, X, 640, 12288, KBIG, out, cyc, "fragments + MFMA only");
    run<6, 128, 128, 2, 2, 4>(W, X, 640, 12288, KBIG, out, cyc, "W direct, X resident, TAP-MAJOR walk", 2);
    printf("-- does a third workgroup per CU (three waves per SIMD) hide a step's non-MFMA instructions?  768 tiles (Q = 1024), K = 5760, one LDS stage; the TAP-MAJOR walk carries a dozen extra address instructions per request\n");
    run<6, 128, 128, 1, 2, 4>(W, X, 1024, 12288, KBIG, out, cyc, "W direct, X resident (2 per CU)");
    run<6, 128, 128, 1, 2, 4>(W, X, 1024, 12288, KBIG, out, cyc, "  same, TAP-MAJOR walk (2 per CU)", 2);
    run<6, 128, 128, 1, 3, 4>(W, X, 1024, 12288, KBIG, out, cyc, "W direct, X resident (3 per CU)");
    run<6, 128, 128, 1, 3, 4>(W, X, 1024, 12288, KBIG, out, cyc, "  same, TAP-MAJOR walk (3 per CU)", 2);
    run<6, 128, 128, 1, 4, 4>(W, X, 1024, 12288, KBIG, out, cyc, "W direct, X resident (4 per CU)");
    run<6, 128, 128, 1, 4, 4>(W, X, 1024, 12288, KBIG, out, cyc, "  same, TAP-MAJOR walk (4 per CU)", 2);
    return 0;
}
