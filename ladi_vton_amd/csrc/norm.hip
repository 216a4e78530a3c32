// GroupNorm (statistics + apply), LayerNorm and row-softmax kernels. All HBM-bound: 16-byte vector
// accesses, fp32 statistics.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int GN_PIX_PER_BLOCK = 64;

// stats[n][g][0] += sum, [1] += sumsq over the virtual concat (src0 | src1), NHWC.
__global__ __launch_bounds__(256) void gn_stats_kernel(const h16* __restrict__ src0, int C0, int ld0,
                                                       const h16* __restrict__ src1, int C1, int ld1, int HW, int groups,
                                                       float* __restrict__ stats) {
    __shared__ float bins[4][64][2];  // per-wave private bins (groups <= 64)
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 4 * 64 * 2; i += 256) (&bins[0][0][0])[i] = 0.f;
    __syncthreads();
    const int n = blockIdx.y;
    const int Ct = C0 + C1;
    const int octs = Ct >> 3;
    const int gs = Ct / groups;
    const int pix0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int npix = min(GN_PIX_PER_BLOCK, HW - pix0);
    const int total = npix * octs;
    for (int idx = tid; idx < total; idx += 256) {
        const int pix = idx / octs, oc = idx - pix * octs;
        const int c = oc << 3;
        const size_t row = (size_t)n * HW + pix0 + pix;
        h16x8 v;
        if (c < C0) v = *reinterpret_cast<const h16x8*>(src0 + row * ld0 + c);
        else v = *reinterpret_cast<const h16x8*>(src1 + row * ld1 + (c - C0));
        int g = c / gs;
        int gend = (g + 1) * gs;  // first channel of next group
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (c + e >= gend) {
                atomicAdd(&bins[wave][g][0], s); atomicAdd(&bins[wave][g][1], ss);
                s = 0.f; ss = 0.f; ++g; gend += gs;
            }
            float x = (float)v[e];
            s += x; ss += x * x;
        }
        atomicAdd(&bins[wave][g][0], s); atomicAdd(&bins[wave][g][1], ss);
    }
    __syncthreads();
    if (tid < groups * 2) {
        const int g = tid >> 1, w = tid & 1;
        float t = bins[0][g][w] + bins[1][g][w] + bins[2][g][w] + bins[3][g][w];
        atomicAdd(&stats[((size_t)n * groups + g) * 2 + w], t);
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const h16* __restrict__ src0, int C0, int ld0,
                                                       const h16* __restrict__ src1, int C1, int ld1, int HW, int groups,
                                                       const float* __restrict__ stats, const h16* __restrict__ gamma,
                                                       const h16* __restrict__ beta, float eps, int silu,
                                                       const h16* __restrict__ add, h16* __restrict__ out, int pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sc_sh[];  // [Ct] scale, [Ct] shift
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int Ct = C0 + C1;
    const int gs = Ct / groups;
    float* scale = sc_sh;
    float* shift = sc_sh + Ct;
    const float inv_cnt = 1.f / ((float)gs * (float)HW);
    for (int c = tid; c < Ct; c += 256) {
        const int g = c / gs;
        const float s = stats[((size_t)n * groups + g) * 2 + 0];
        const float ss = stats[((size_t)n * groups + g) * 2 + 1];
        const float mean = s * inv_cnt;
        const float var = fmaxf(ss * inv_cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        const float ga = (float)gamma[c] * rstd;
        scale[c] = ga;
        shift[c] = (float)beta[c] - mean * ga;
    }
    __syncthreads();
    const int octs = Ct >> 3;
    const int pix0 = blockIdx.x * pix_per_block;
    const int npix = min(pix_per_block, HW - pix0);
    const int total = npix * octs;
    for (int idx = tid; idx < total; idx += 256) {
        const int pix = idx / octs, oc = idx - pix * octs;
        const int c = oc << 3;
        const size_t row = (size_t)n * HW + pix0 + pix;
        h16x8 v;
        if (c < C0) v = *reinterpret_cast<const h16x8*>(src0 + row * ld0 + c);
        else v = *reinterpret_cast<const h16x8*>(src1 + row * ld1 + (c - C0));
        h16x8 o;
        h16x8 ad;
        if (add) ad = *reinterpret_cast<const h16x8*>(add + row * Ct + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[e] * scale[c + e] + shift[c + e];
            if (silu) y = silu_f(y);
            if (add) y += (float)ad[e];
            o[e] = (h16)y;
        }
        *reinterpret_cast<h16x8*>(out + row * Ct + c) = o;
    }
}

// one wave per row; C % 8 == 0, C <= 4096
__global__ __launch_bounds__(256) void layernorm_kernel(const h16* __restrict__ x, int ldx, const h16* __restrict__ gamma,
                                                        const h16* __restrict__ beta, float eps, int rows, int C,
                                                        h16* __restrict__ out, int ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int octs = C >> 3;
    constexpr int MAXO = 8;  // up to 8 octets per lane -> C <= 4096
    h16x8 v[MAXO];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int oc = lane + 64 * i;
        if (oc < octs) {
            v[i] = *reinterpret_cast<const h16x8*>(x + (size_t)row * ldx + oc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[i][e];
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int oc = lane + 64 * i;
        if (oc < octs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float d = (float)v[i][e] - mean; ss += d * d; }
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int oc = lane + 64 * i;
        if (oc < octs) {
            const h16x8 g = *reinterpret_cast<const h16x8*>(gamma + oc * 8);
            const h16x8 b = *reinterpret_cast<const h16x8*>(beta + oc * 8);
            h16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)v[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            *reinterpret_cast<h16x8*>(out + (size_t)row * ldo + oc * 8) = o;
        }
    }
}

// one block per row: P = softmax(scale * S) ; cols % 4 == 0
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int cols, float scale,
                                                           h16* __restrict__ P) {
    __shared__ float red[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* s = S + (size_t)blockIdx.x * cols;
    h16* p = P + (size_t)blockIdx.x * cols;
    const int q4 = cols >> 2;
    float m = -3.0e38f;
    for (int i = tid; i < q4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(s)[i];
        m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float sc = scale * 1.4426950408889634f;
    const float msc = m * sc;
    float sum = 0.f;
    for (int i = tid; i < q4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(s)[i];
        sum += exp2f(v.x * sc - msc) + exp2f(v.y * sc - msc) + exp2f(v.z * sc - msc) + exp2f(v.w * sc - msc);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.f / sum;
    for (int i = tid; i < q4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(s)[i];
        h16x4 o;
        o[0] = (h16)(exp2f(v.x * sc - msc) * inv); o[1] = (h16)(exp2f(v.y * sc - msc) * inv);
        o[2] = (h16)(exp2f(v.z * sc - msc) * inv); o[3] = (h16)(exp2f(v.w * sc - msc) * inv);
        reinterpret_cast<h16x4*>(p)[i] = o;
    }
}

}  // namespace

int ladi_launch_gn_stats(const h16* src0, int C0, int ld0, const h16* src1, int C1, int ld1, int n, int HW, int groups,
                         float* stats, hipStream_t st) {
    const int Ct = C0 + C1;
    if ((C0 & 7) || (C1 & 7) || groups > 64 || (Ct % groups) || (ld0 & 7) || (C1 && (ld1 & 7))) return -1;
    dim3 grid((HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK, n);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, src0, C0, ld0, src1, C1, ld1, HW, groups, stats);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_gn_apply(const h16* src0, int C0, int ld0, const h16* src1, int C1, int ld1, int n, int HW, int groups,
                         const float* stats, const h16* gamma, const h16* beta, float eps, int silu, const h16* add,
                         h16* out, hipStream_t st) {
    const int Ct = C0 + C1;
    if ((C0 & 7) || (C1 & 7) || (Ct % groups) || (ld0 & 7) || (C1 && (ld1 & 7))) return -1;
    // enough pixels per block to amortise the per-block scale/shift setup (Ct rsqrt's)
    int ppb = 64;
    while (ppb * (Ct >> 3) < 4096 && ppb < HW) ppb <<= 1;
    dim3 grid((HW + ppb - 1) / ppb, n);
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), (size_t)Ct * 2 * sizeof(float), st, src0, C0, ld0, src1, C1, ld1, HW,
                       groups, stats, gamma, beta, eps, silu, add, out, ppb);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_layernorm(const h16* x, int ldx, const h16* gamma, const h16* beta, float eps, int rows, int C, h16* out,
                          int ldo, hipStream_t st) {
    if ((C & 7) || C > 4096 || (ldx & 7) || (ldo & 7)) return -1;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, gamma, beta, eps, rows, C, out, ldo);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_softmax_rows(const float* S, int rows, int cols, float scale, h16* P, hipStream_t st) {
    if (cols & 3) return -1;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, S, cols, scale, P);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}
