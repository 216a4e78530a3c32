// fp32 path of the warping module.  The reference forces this stage to fp32 (src/inference.py:253 `tps(low_cloth.to(torch.float32),
// agnostic.to(torch.float32))`, :264 `refinement(warped_cloth.to(torch.float32))`): a geometric regression whose 50 control-point
// outputs move every pixel of the warped cloth.  When the caller passes fp32 tensors -- as inference.py does -- the TPS network and the
// refinement UNet therefore run with fp32 weights, fp32 activations (NHWC) and fp32 accumulation on the f32-input matrix instruction
// v_mfma_f32_32x32x2_f32 (exact fp32: bit-for-bit a k-ordered fmaf chain, cdna_hip_programming.md section 3) -- 1/16 of the fp16 MFMA
// rate, for a stage that runs once per batch at 256x192 / 512x384.  fp16 callers keep the fp16 kernels.
//
// conv_f32_kernel: implicit GEMM, D[q][p] = sum_k W[q][k] X[p][k], no LDS staging: a 32x32x2 MFMA takes one float per lane per
// operand (lane l: row l & 31, k = l >> 5), so a lane loads float4 = 4 consecutive channels of ITS weight row and ITS pixel -- lanes
// 0-31 channels c..c+3, lanes 32-63 channels c+4..c+7 -- and feeds element e of both vectors to MFMA e: the four MFMAs of a chunk cover
// 8 channels (k pairs {c+e, c+4+e}; any pairing is valid as long as both operands use the same one).  At 64 cycles per MFMA the loads
// (4 x 16 B per 16 MFMAs) are far off the critical path; 128-B lines are re-used from L1 across the 4 chunks that share them.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float act_f32(float x, int act) {
    if (act == LADI_ACT_RELU) return fmaxf(x, 0.f);
    if (act == LADI_ACT_TANH) return tanhf(x);
    if (act == LADI_ACT_SILU) return x / (1.f + expf(-x));
    return x;
}

// 4 waves as WQ x WP, wave tile 64 channels x 64 pixels (2 x 2 MFMA tiles)
template <int WQ, int WP>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvF32Args a) {
    constexpr int BQ = WQ * 64, BP = WP * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wq = wave / WP, wp = wave % WP;
    const int l31 = lane & 31, hh = lane >> 5;
    const int nq = (a.Q + BQ - 1) / BQ;
    const int qt = blockIdx.x % nq, pt = blockIdx.x / nq, z = blockIdx.z;
    const int q0 = qt * BQ + wq * 64, p0 = pt * BP + wp * 64;
    const float* __restrict__ W = a.W + (size_t)z * a.bs_w;
    const float* __restrict__ s0 = a.src0 + (size_t)z * a.bs_src0;
    const float* __restrict__ s1 = a.src1;
    const int HoWo = a.Ho * a.Wo, Ct = a.C0 + a.C1;
    const int ldw = a.ldw ? a.ldw : a.K;

    size_t wrow[2]; bool wok[2];
    int pn[2], py[2], px[2]; bool pok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = q0 + i * 32 + l31;
        wok[i] = q < a.Q;
        wrow[i] = (size_t)(wok[i] ? q : 0) * ldw;
        const int p = p0 + i * 32 + l31;
        pok[i] = p < a.P;
        const int pp = pok[i] ? p : 0;
        pn[i] = pp / HoWo;
        const int rem = pp - pn[i] * HoWo;
        py[i] = (rem / a.Wo) * a.stride - a.pad;
        px[i] = (rem % a.Wo) * a.stride - a.pad;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int tap = 0;
    for (int dy = 0; dy < a.ksize; ++dy)
        for (int dx = 0; dx < a.ksize; ++dx, ++tap) {
            size_t xoff0[2], xoff1[2]; bool xv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int iy = py[j] + dy, ix = px[j] + dx;
                xv[j] = pok[j] && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws;
                const size_t pix = ((size_t)pn[j] * a.Hs + (xv[j] ? iy : 0)) * a.Ws + (xv[j] ? ix : 0);
                xoff0[j] = pix * a.ld0; xoff1[j] = pix * a.ld1;
            }
            const size_t wk = (size_t)tap * Ct + 4 * hh;
            for (int cb = 0; cb < Ct; cb += 8) {
                const bool first = cb < a.C0;
                const int c = (first ? cb : cb - a.C0) + 4 * hh;
                f32x4 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = wok[i] ? *reinterpret_cast<const f32x4*>(W + wrow[i] + wk + cb) : zero4;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bf[j] = xv[j] ? *reinterpret_cast<const f32x4*>((first ? s0 + xoff0[j] : s1 + xoff1[j]) + c) : zero4;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            }
        }

    // epilogue: lane owns pixel column l31 of each pixel sub-tile and 4 consecutive channels per register group
    float* __restrict__ out = a.out + (size_t)z * a.bs_out;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = p0 + j * 32 + l31;
        if (p >= a.P) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q = q0 + i * 32 + 8 * g + 4 * hh;
                if (q >= a.Q) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][4 * g + e];
                    if (a.bias && q + e < a.Q) x += a.bias[q + e];
                    v[e] = act_f32(x, a.act);
                }
                float* op = out + (size_t)p * a.ldo + q;
                if (q + 3 < a.Q && !(a.ldo & 3)) *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (q + e < a.Q) op[e] = v[e];
                }
            }
    }
}

// NCHW (fp32 / fp16) -> NHWC fp32, channels >= C zero
__global__ void nchw_to_nhwc_f32_kernel(const void* __restrict__ src, int in_f32, int C, int HW, float* __restrict__ dst, int ld, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % ld);
    const size_t pix = idx / ld;
    const size_t b = pix / HW, s = pix - b * HW;
    float v = 0.f;
    if (c < C) {
        const size_t i = (b * C + c) * HW + s;
        v = in_f32 ? reinterpret_cast<const float*>(src)[i] : (float)reinterpret_cast<const h16*>(src)[i];
    }
    dst[idx] = v;
}
__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ src, int ld, int C, int HW, void* __restrict__ dst, int out_f32, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [B][C][HW]
    if (idx >= total) return;
    const size_t s = idx % HW, bc = idx / HW;
    const size_t c = bc % C, b = bc / C;
    const float v = src[(b * HW + s) * ld + c];
    if (out_f32) reinterpret_cast<float*>(dst)[idx] = v; else reinterpret_cast<h16*>(dst)[idx] = (h16)v;
}
__global__ void channel_affine_f32_kernel(float* __restrict__ x, int ld, size_t n_pix, int C, const float* __restrict__ scale, const float* __restrict__ shift) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix * C) return;
    const int c = (int)(idx % C);
    float* p = x + (idx / C) * ld + c;
    *p = *p * scale[c] + shift[c];
}
// y = x / sqrt(sum_c x^2 + 1e-6)  (FeatureL2Norm, ConvNet_TPS.py:59-66); one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_f32_kernel(float* __restrict__ x, int ld, int rows, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + (size_t)row * ld;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += p[c] * p[c];
    s = wave_sum(s);
    const float inv = 1.f / sqrtf(s + 1e-6f);
    for (int c = lane; c < C; c += 64) p[c] *= inv;
}
__global__ void gather_rows_f32_kernel(const float* __restrict__ src, const int* __restrict__ rows, int H, float* __restrict__ dst) {
    const float* s = src + (size_t)rows[blockIdx.x] * H;
    float* d = dst + (size_t)blockIdx.x * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) d[i] = s[i];
}
__global__ void maxpool2_f32_kernel(const float* __restrict__ src, int lds_, int n, int H, int W, int C, float* __restrict__ dst, int ldd) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * Ho * Wo * C) return;
    const int c = (int)(idx % C);
    size_t p = idx / C;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const float* base = src + (((size_t)b * H + 2 * oy) * W + 2 * ox) * lds_ + c;
    dst[(((size_t)b * Ho + oy) * Wo + ox) * ldd + c] = fmaxf(fmaxf(base[0], base[lds_]), fmaxf(base[(size_t)W * lds_], base[(size_t)W * lds_ + lds_]));
}
// bilinear x2 with align_corners=True (unet_parts.py Up)
__global__ void upsample2x_bilinear_ac_f32_kernel(const float* __restrict__ src, int lds_, int n, int H, int W, int C, float* __restrict__ dst, int ldd) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * Ho * Wo * C) return;
    const int c = (int)(idx % C);
    size_t p = idx / C;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const float sy = Ho > 1 ? (float)oy * (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)ox * (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float* base = src + (size_t)b * H * W * lds_ + c;
    const float v00 = base[((size_t)y0 * W + x0) * lds_], v01 = base[((size_t)y0 * W + x1) * lds_];
    const float v10 = base[((size_t)y1 * W + x0) * lds_], v11 = base[((size_t)y1 * W + x1) * lds_];
    const float top = v00 + (v01 - v00) * fx, bot = v10 + (v11 - v10) * fx;
    dst[(((size_t)b * Ho + oy) * Wo + ox) * ldd + c] = top + (bot - top) * fy;
}
// out[m][n] = act(sum_k x[m][k] W[n][k] + b[n]), everything fp32; one wave per output element row block (tiny: 50 x 768)
__global__ __launch_bounds__(64) void linear_f32_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W, const float* __restrict__ b, int K,
                                                        int act, float* __restrict__ out, int ldo) {
    const int m = blockIdx.y, n = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += x[(size_t)m * ldx + k] * W[(size_t)n * K + k];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)m * ldo + n] = act_f32(s + (b ? b[n] : 0.f), act);
}

inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -1; }

}  // namespace

int ladi_launch_conv_f32(const ConvF32Args& a, int batch, hipStream_t st) {
    if (a.ksize < 1 || a.ksize > 4 || (a.C0 % 8) || (a.C1 % 8) || (a.ld0 % 4) || (a.C1 && (a.ld1 % 4)) || a.P <= 0 || a.Q <= 0) return -1;
    if (a.K != a.ksize * a.ksize * (a.C0 + a.C1) || ((a.ldw ? a.ldw : a.K) % 4)) return -3;
    if ((reinterpret_cast<uintptr_t>(a.src0) | reinterpret_cast<uintptr_t>(a.src1) | reinterpret_cast<uintptr_t>(a.W) | reinterpret_cast<uintptr_t>(a.out)) & 15) return -4;
    if (a.Q <= 64) {       // narrow outputs (the 3-channel OutConv, 64-channel first layers): all four waves along the pixels
        const int np = (a.P + 255) / 256;
        hipLaunchKernelGGL((conv_f32_kernel<1, 4>), dim3((unsigned)np, 1, (unsigned)batch), dim3(256), 0, st, a);
    } else {
        const int nq = (a.Q + 127) / 128, np = (a.P + 127) / 128;
        hipLaunchKernelGGL((conv_f32_kernel<2, 2>), dim3((unsigned)(nq * np), 1, (unsigned)batch), dim3(256), 0, st, a);
    }
    return ok();
}
int ladi_launch_nchw_to_nhwc_f32(const void* src, int in_f32, int n, int C, int H, int W, float* dst, int ld, hipStream_t st) {
    const size_t total = (size_t)n * H * W * ld;
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, in_f32, C, H * W, dst, ld, total);
    return ok();
}
int ladi_launch_nhwc_to_nchw_f32(const float* src, int ld, int n, int C, int H, int W, void* dst, int out_f32, hipStream_t st) {
    const size_t total = (size_t)n * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, ld, C, H * W, dst, out_f32, total);
    return ok();
}
int ladi_launch_channel_affine_f32(float* x, int ld, size_t n_pix, int C, const float* scale, const float* shift, hipStream_t st) {
    const size_t total = n_pix * C;
    hipLaunchKernelGGL(channel_affine_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ld, n_pix, C, scale, shift);
    return ok();
}
int ladi_launch_l2norm_rows_f32(float* x, int ld, int rows, int C, hipStream_t st) {
    hipLaunchKernelGGL(l2norm_rows_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, ld, rows, C);
    return ok();
}
int ladi_launch_gather_rows_f32(const float* src, const int* rows, int n, int H, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((unsigned)n), dim3(256), 0, st, src, rows, H, dst);
    return ok();
}
int ladi_launch_maxpool2_f32(const float* src, int lds_, int n, int H, int W, int C, float* dst, int ldd, hipStream_t st) {
    if ((H & 1) || (W & 1)) return -1;
    const size_t total = (size_t)n * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(maxpool2_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds_, n, H, W, C, dst, ldd);
    return ok();
}
int ladi_launch_upsample2x_bilinear_ac_f32(const float* src, int lds_, int n, int H, int W, int C, float* dst, int ldd, hipStream_t st) {
    const size_t total = (size_t)n * (2 * H) * (2 * W) * C;
    hipLaunchKernelGGL(upsample2x_bilinear_ac_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds_, n, H, W, C, dst, ldd);
    return ok();
}
int ladi_launch_linear_f32(const float* x, int ldx, const float* W, const float* b, int M, int N, int K, int act, float* out, int ldo, hipStream_t st) {
    hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)N, (unsigned)M), dim3(64), 0, st, x, ldx, W, b, K, act, out, ldo);
    return ok();
}
