// Which part of the implicit-GEMM staging stream is slow?  The LDS-DMA path alone sustains 40-48 B/clk/CU (tools/lds_dma_rate.hip), the
// conv kernels stage at 13-17 B/clk/CU even with every MFMA removed (profiles/r02_igemm_microbench.txt, ABL = 4).  This program replays
// the staging stream of the 320x256 tile on the 3x3 conv 320 -> 320 @ 64x48, n = 16 (192 workgroups x 45 K tiles x 9 pieces of 8 KiB) with
// NOTHING but the DMA instructions, a counted wait and one barrier per K tile, and switches parts of the stream off:
//   mode bit 0: weight pieces   bit 1: pixel pieces   bit 2: pixel pieces read the SAME tap every step (no window shift)
//   bit 3: every workgroup reads pixel tile 0 (L2-resident X)   bit 4: no halo masking (all lanes in range)
//   hipcc --offload-arch=gfx950 -O3 tools/dma_conv_pattern.hip -o /tmp/dma_conv && /tmp/dma_conv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef _Float16 h16;

template <int DEPTH>   // K tiles in flight (ring slots - 1)
__global__ __launch_bounds__(512, 2) void stage_loop(const h16* X, const h16* W, int mode, int H, int Wd, int C, int Q, int P, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BQ = 320, BP = 256, BK = 64, NB = BP / 64, NA = BQ / 64, PIECE = 8192, TILE = (NA + NB) * PIECE;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = tid >> 3, clog = (tid & 7) ^ ((r0 >> 1) & 7);
    const int np = P / BP;
    const int b = blockIdx.x, xcd = b & 7, loc = b >> 3, npx = (np + 7) >> 3;
    int pt = xcd * npx + loc;
    if (pt >= np) return;
    if (mode & 8) pt = 0;
    const int p0 = pt * BP, HW = H * Wd, n_first = p0 / HW;
    const int back = Wd + 1;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(X + (size_t)n_first * HW * C) - (ptrdiff_t)back * C, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(W), 0, 0x7FFFFFFF, 0x00020000);
    unsigned vox[NB], vmask[NB], wb[NA];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int p = p0 + r0 + 64 * i, n = p / HW, rem = p - n * HW, oy = rem / Wd, ox = rem - oy * Wd;
        const int y0 = oy - 1, x0 = ox - 1;
        vox[i] = (unsigned)((((n - n_first) * HW + y0 * Wd + x0 + back) * C + clog * 8) * 2);
        unsigned m = 0;
        for (int t = 0; t < 9; ++t)
            if ((unsigned)(y0 + t / 3) < (unsigned)H && (unsigned)(x0 + t % 3) < (unsigned)Wd) m |= 1u << t;
        vmask[i] = (mode & 16) ? 0x1ffu : m;
    }
    const int K = 9 * C;
#pragma unroll
    for (int i = 0; i < NA; ++i) wb[i] = (unsigned)((((size_t)(r0 + 64 * i)) * K + clog * 8) * 2);
    const int nk = K / BK;
    int tap = 0, cb = 0;
    auto issue = [&](int slot) {
        char* buf = smem + slot * TILE + wave * 1024;
        const int t = (mode & 4) ? 4 : tap;
        const unsigned so_x = (unsigned)((((t / 3) * Wd + (t % 3)) * C + cb) * 2), so_w = (unsigned)((tap * C + cb) * 2), bit = 1u << t;
        if (mode & 2) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(buf + (NA + i) * PIECE), 16, (vmask[i] & bit) ? vox[i] : 0x80000000u, so_x, 0, 0);
        }
        if (mode & 1) {
#pragma unroll
            for (int i = 0; i < NA; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(buf + i * PIECE), 16, wb[i], so_w, 0, 0);
        }
        if (++tap == 9) { tap = 0; cb += BK; }
    };
    const int per = ((mode & 2) ? NB : 0) + ((mode & 1) ? NA : 0);
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int s = 0; s < DEPTH && s < nk; ++s) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        const int later = min(DEPTH - 1, nk - 1 - kt);
        // wait until only `later` K tiles of mine are outstanding, then rendezvous (the slot read by "compute" is free again)
        if (later >= 2) { if (per == 9) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else if (per == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (later == 1) { if (per == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else if (per == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt + DEPTH < nk) issue((kt + DEPTH) % (DEPTH + 1));
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (tid == 0) out[blockIdx.x] = c1 - c0;
}

template <int DEPTH>
static void run(const h16* X, const h16* W, int mode, int blocks) {
    const int H = 64, Wd = 48, C = 320, Q = 320, P = 16 * H * Wd;
    unsigned long long* d = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&d), blocks * sizeof(unsigned long long));
    (void)hipMemset(d, 0, blocks * sizeof(unsigned long long));
    const int smem = (DEPTH + 1) * 9 * 8192;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stage_loop<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(stage_loop<DEPTH>, dim3(blocks), dim3(512), smem, 0, X, W, mode, H, Wd, C, Q, P, d);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(stage_loop<DEPTH>, dim3(blocks), dim3(512), smem, 0, X, W, mode, H, Wd, C, Q, P, d);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), d, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<unsigned long long> v;
    for (auto x : h) if (x) v.push_back(x);
    std::sort(v.begin(), v.end());
    const int per = ((mode & 2) ? 4 : 0) + ((mode & 1) ? 5 : 0);
    const double bytes = 45.0 * per * 8192.0;
    printf("depth %d mode %2d (%s%s%s%s%s): %7.1f us per launch, median workgroup %8llu cycles = %5.1f B/clk/CU, %5.2f TB/s aggregate\n", DEPTH, mode,
           (mode & 1) ? "W " : "", (mode & 2) ? "X " : "", (mode & 4) ? "same-tap " : "", (mode & 8) ? "one-X-tile " : "", (mode & 16) ? "no-halo " : "",
           ms * 100.0, v[v.size() / 2], bytes / (double)v[v.size() / 2], bytes * v.size() / (ms / 10.0) / 1e9);
    (void)hipFree(d);
}

int main() {
    const size_t nx = (size_t)16 * 64 * 48 * 320, nw = (size_t)320 * 2880;
    h16 *X = nullptr, *W = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&X), (nx + (1 << 20)) * 2);
    (void)hipMalloc(reinterpret_cast<void**>(&W), nw * 2);
    (void)hipMemset(X, 0, (nx + (1 << 20)) * 2);
    (void)hipMemset(W, 0, nw * 2);
    X += 1 << 19;     // room in front for the rebased descriptor
    for (int mode : {3, 1, 2, 6, 10, 18, 11, 7})
        run<1>(X, W, mode, 192);
    run<1>(X, W, 3, 256);   // tile_map pads the grid to 8 * 24 = 192 anyway; 256 blocks: the extra ones return at once
    return 0;
}
