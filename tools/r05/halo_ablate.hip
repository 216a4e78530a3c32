// What bounds the dominant kernel of the UNet forward, igemm_halo_kernel<2,2,1,3,2,24> (128x128 tile, 4 waves, two workgroups per CU)?
// The SAME kernel source compiled with parts switched off (-DLADI_HALO_ABL=<mask>, see igemm_halo.hip) on the 3x3 convolution 640 -> 640 at
// 32x24, n = 16 (P = 12 288, K = 5 760: 6 of its 20 launches per forward; 90.6 GFLOP).  Built HERE for every mask by tools/r05/build_ablate.sh
// (cross-compiled, the binaries travel with the snapshot), run on the GPU box by tools/r05/call7.sh.
//   mask 0 full | 1 no weight DMA | 2 no halo DMA | 3 no DMA at all | 4 no fragment reads | 8 no MFMA | 16 no wait + barrier | combinations
#define LADI_HALO_TOOL 1
#include "../../ladi_vton_amd/csrc/igemm_halo_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = 16, H = argc > 1 ? atoi(argv[1]) : 32, W = argc > 2 ? atoi(argv[2]) : 24, C = argc > 3 ? atoi(argv[3]) : 640, Q = argc > 4 ? atoi(argv[4]) : 640;
    const int P = n * H * W, K = 9 * C;
    std::vector<h16> hx((size_t)P * C), hw((size_t)Q * K), hb(Q);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)((s >> 9) & 0xffff) / 65536.f - 0.5f); };
    for (auto& v : hx) v = (h16)(rnd() * 2.f);
    for (auto& v : hw) v = (h16)(rnd() * 0.05f);
    for (auto& v : hb) v = (h16)rnd();
    h16 *dx, *dw, *db, *dout;
    CK(hipMalloc((void**)&dx, hx.size() * 2)); CK(hipMalloc((void**)&dw, hw.size() * 2)); CK(hipMalloc((void**)&db, hb.size() * 2));
    CK(hipMalloc((void**)&dout, (size_t)P * Q * 2));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    IGemmArgs a = {};
    a.src0 = dx; a.C0 = C; a.ld0 = C; a.Hs = H; a.Ws = W; a.Ho = H; a.Wo = W; a.P = P; a.ksize = 3; a.stride = 1; a.pad = 1;
    a.W = dw; a.Q = Q; a.K = K; a.bias = db; a.act = LADI_ACT_NONE; a.out_scale = 1.f; a.out = dout; a.ldo = Q; a.splitk = 1;
    hipStream_t st; CK(hipStreamCreate(&st));
    auto launch = [&]() { return W <= 24 ? launch_halo<2, 2, 1, 3, 2, 24>(a, 1, st) : launch_halo<2, 2, 1, 2, 2>(a, 1, st); };
    for (int i = 0; i < 3; ++i) if (launch() != 0) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 30;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters, gf = 2.0 * P * Q * (double)K / 1e9;
    printf("ABL %2d  %dx%d %d->%d  %8.1f us  %6.0f TFLOP/s-equivalent\n", LADI_HALO_ABL, H, W, C, Q, us, gf / us * 1e3);
    return 0;
}
