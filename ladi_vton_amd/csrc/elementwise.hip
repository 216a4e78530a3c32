// HBM-bound / tiny kernels of the try-on hot path: layout conversion at the NCHW boundary, small-M
// weight-streaming linear (time embedding, inversion-adapter head), the fused CFG + DDIM/PLMS scheduler
// step + next-UNet-input assembly, posterior sampling, mask / pose resizes.
#include "common.h"
#include "kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
// small-M linear: one wave per output column, MT rows of x per pass. W [N][K] fp16 (K % 8 == 0).
// ------------------------------------------------------------------------------------------------
template <typename XT, int MT>
__global__ __launch_bounds__(256) void small_linear_kernel(const XT* __restrict__ x, int ldx, const h16* __restrict__ W,
                                                           const h16* __restrict__ bias, const h16* __restrict__ res, int ldr,
                                                           int M, int N, int K, int act, int pre_silu, void* __restrict__ out,
                                                           int out_f32, int ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nn = blockIdx.x * 4 + wave;
    const int m0 = blockIdx.y * MT;
    if (nn >= N) return;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    const h16* wrow = W + (size_t)nn * K;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const h16x8 w = *reinterpret_cast<const h16x8*>(wrow + k);
        float wf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[e] = (float)w[e];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + i;
            if (m < M) {
                const XT* xr = x + (size_t)m * ldx + k;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xv = (float)xr[e];
                    if (pre_silu) xv = silu_f(xv);
                    s += xv * wf[e];
                }
                acc[i] += s;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float s = wave_sum(acc[i]);
        const int m = m0 + i;
        if (lane == 0 && m < M) {
            if (bias) s += (float)bias[nn];
            if (act == LADI_ACT_SILU) s = silu_f(s);
            else if (act == LADI_ACT_GELU) s = gelu_f(s);
            else if (act == LADI_ACT_RELU) s = fmaxf(s, 0.f);
            else if (act == LADI_ACT_TANH) s = tanhf(s);
            if (res) s += (float)res[(size_t)m * ldr + nn];
            if (out_f32) reinterpret_cast<float*>(out)[(size_t)m * ldo + nn] = s;
            else reinterpret_cast<h16*>(out)[(size_t)m * ldo + nn] = (h16)s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NCHW (fp32/fp16) -> NHWC fp16 with channel padding to ld (zeros); LDS-tiled transpose over (C, W)
// ------------------------------------------------------------------------------------------------
template <typename ST>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const ST* __restrict__ src, int C, int HW, h16* __restrict__ dst,
                                                           int ld) {
    // block handles 64 pixels x all channels (in chunks of 64 channels)
    __shared__ h16 tile[64][65];
    const int n = blockIdx.y;
    const int pix0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // ty 0..3
    for (int c0 = 0; c0 < ld; c0 += 64) {
        for (int cc = ty; cc < 64; cc += 4) {
            const int c = c0 + cc, pix = pix0 + tx;
            h16 v = (h16)0.f;
            if (c < C && pix < HW) v = (h16)(float)src[((size_t)n * C + c) * HW + pix];
            tile[cc][tx] = v;
        }
        __syncthreads();
        for (int pp = ty; pp < 64; pp += 4) {
            const int pix = pix0 + pp, c = c0 + tx;
            if (pix < HW && c < ld) dst[((size_t)n * HW + pix) * ld + c] = tile[tx][pp];
        }
        __syncthreads();
    }
}

template <typename DT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const h16* __restrict__ src, int ld, int C, int HW,
                                                           DT* __restrict__ dst) {
    __shared__ h16 tile[64][65];
    const int n = blockIdx.y;
    const int pix0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c0 = 0; c0 < C; c0 += 64) {
        for (int pp = ty; pp < 64; pp += 4) {
            const int pix = pix0 + pp, c = c0 + tx;
            h16 v = (h16)0.f;
            if (pix < HW && c < C) v = src[((size_t)n * HW + pix) * ld + c];
            tile[pp][tx] = v;
        }
        __syncthreads();
        for (int cc = ty; cc < 64; cc += 4) {
            const int c = c0 + cc, pix = pix0 + tx;
            if (c < C && pix < HW) dst[((size_t)n * C + c) * HW + pix] = (DT)(float)tile[tx][cc];
        }
        __syncthreads();
    }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int count, int dim, float* __restrict__ out) {
    const int i = blockIdx.x, j = threadIdx.x + blockIdx.y * blockDim.x;
    const int half = dim / 2;
    if (i >= count || j >= half) return;
    const float freq = __expf(-9.210340371976184f * (float)j / (float)half);  // ln(10000)
    const float arg = t[i] * freq;
    // flip_sin_to_cos: [cos | sin]
    out[(size_t)i * dim + j] = cosf(arg);
    out[(size_t)i * dim + half + j] = sinf(arg);
}

// ------------------------------------------------------------------------------------------------
// fused CFG combine + scheduler update (DDIM or PLMS table entry) + UNet-input refresh
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sched_step_body(const StepArgs& a, const int idx, const int total, const int step) {
    const StepTable T = a.table[step];
    const int b = idx / a.hw;
    const size_t row_u = (size_t)idx;                       // uncond (or only) row
    const size_t row_c = (size_t)idx + (size_t)total;       // cond row when cfg
    float e[4];
    {
        const h16x4 eu = *reinterpret_cast<const h16x4*>(a.eps + row_u * a.ld_eps);
        if (a.cfg) {
            const h16x4 ec = *reinterpret_cast<const h16x4*>(a.eps + row_c * a.ld_eps);
#pragma unroll
            for (int c = 0; c < 4; ++c) { float u = (float)eu[c]; e[c] = u + a.guidance * ((float)ec[c] - u); }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = (float)eu[c];
        }
    }
    (void)b;
    const size_t plane = (size_t)total * 4;
    float4 x = reinterpret_cast<const float4*>(a.latents)[idx];
    if (T.save_cur) reinterpret_cast<float4*>(a.cur_sample)[idx] = x;
    if (T.mode == 1) x = reinterpret_cast<const float4*>(a.cur_sample)[idx];
    float ec[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ec[c] = T.w[0] * e[c];
    // history slots are encoded in the table through w[] order: slot ids packed in push field's upper bits
    const int s1 = (T.push >> 8) & 3, s2 = (T.push >> 10) & 3, s3 = (T.push >> 12) & 3, sp = (T.push >> 4) & 3;
    if (T.w[1] != 0.f) { const float4 h = reinterpret_cast<const float4*>(a.ets + s1 * plane)[idx];
        ec[0] += T.w[1] * h.x; ec[1] += T.w[1] * h.y; ec[2] += T.w[1] * h.z; ec[3] += T.w[1] * h.w; }
    if (T.w[2] != 0.f) { const float4 h = reinterpret_cast<const float4*>(a.ets + s2 * plane)[idx];
        ec[0] += T.w[2] * h.x; ec[1] += T.w[2] * h.y; ec[2] += T.w[2] * h.z; ec[3] += T.w[2] * h.w; }
    if (T.w[3] != 0.f) { const float4 h = reinterpret_cast<const float4*>(a.ets + s3 * plane)[idx];
        ec[0] += T.w[3] * h.x; ec[1] += T.w[3] * h.y; ec[2] += T.w[3] * h.z; ec[3] += T.w[3] * h.w; }
    if (T.push & 1) reinterpret_cast<float4*>(a.ets + sp * plane)[idx] = make_float4(e[0], e[1], e[2], e[3]);
    float4 xn;
    xn.x = T.c_x * x.x + T.c_e * ec[0]; xn.y = T.c_x * x.y + T.c_e * ec[1];
    xn.z = T.c_x * x.z + T.c_e * ec[2]; xn.w = T.c_x * x.w + T.c_e * ec[3];
    reinterpret_cast<float4*>(a.latents)[idx] = xn;
    if (step < a.trace_cap) {
        if (a.trace_eps) reinterpret_cast<float4*>(a.trace_eps)[(size_t)step * total + idx] = make_float4(e[0], e[1], e[2], e[3]);
        if (a.trace_lat) reinterpret_cast<float4*>(a.trace_lat)[(size_t)step * total + idx] = xn;
    }
    if (a.unet_in) {
        const float is = T.in_scale_next;
        h16x4 o; o[0] = (h16)(xn.x * is); o[1] = (h16)(xn.y * is); o[2] = (h16)(xn.z * is); o[3] = (h16)(xn.w * is);
        *reinterpret_cast<h16x4*>(a.unet_in + row_u * a.ld_in) = o;
        if (a.cfg) *reinterpret_cast<h16x4*>(a.unet_in + row_c * a.ld_in) = o;
        if (T.zero_cloth_next) {
            h16x4 zz = {0, 0, 0, 0};
            // cloth channels are not 8-byte aligned (27..30): scalar stores
            h16* pu = a.unet_in + row_u * a.ld_in + a.cloth_ch0;
            pu[0] = zz[0]; pu[1] = zz[1]; pu[2] = zz[2]; pu[3] = zz[3];
            if (a.cfg) { h16* pc = a.unet_in + row_c * a.ld_in + a.cloth_ch0; pc[0] = zz[0]; pc[1] = zz[1]; pc[2] = zz[2]; pc[3] = zz[3]; }
        }
    }
}
// Every block reads the evaluation index first; the LAST block to finish (arrival ticket at step_idx[1]) advances it for the next
// evaluation and re-arms the ticket: the counter needs no launch of its own and cannot change under a block that has not read it yet.
__global__ __launch_bounds__(256) void sched_step_kernel(const StepArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // (b, pixel)
    const int total = a.B * a.hw;
    const int step = *a.step_idx;
    if (idx < total) sched_step_body(a, idx, total, step);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(a.step_idx + 1, 1) == (int)gridDim.x - 1) { a.step_idx[1] = 0; a.step_idx[0] = step + 1; }
    }
}

__global__ __launch_bounds__(256) void assemble_static_kernel(h16* __restrict__ unet_in, int ld_in, int B, int hw, int cfg,
                                                              const float* __restrict__ latents,
                                                              const h16* __restrict__ mask_lat,
                                                              const float* __restrict__ masked_lat,
                                                              const h16* __restrict__ pose, int pose_ch,
                                                              const float* __restrict__ cloth_lat, int has_cloth, float lat_scale) {
    const int rows = (cfg ? 2 : 1) * B * hw;
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int total = B * hw;
    const bool cond = cfg ? (row >= total) : true;
    const int src = cfg ? (row % total) : row;
    h16* o = unet_in + (size_t)row * ld_in;
    int c = 0;
    for (int i = 0; i < 4; ++i) o[c++] = (h16)(latents[(size_t)src * 4 + i] * lat_scale);   // scale_model_input of evaluation 0
    o[c++] = mask_lat[src];
    for (int i = 0; i < 4; ++i) o[c++] = (h16)masked_lat[(size_t)src * 4 + i];
    for (int i = 0; i < pose_ch; ++i) o[c++] = cond ? pose[(size_t)src * pose_ch + i] : (h16)0.f;
    if (has_cloth) for (int i = 0; i < 4; ++i) o[c++] = cond ? (h16)cloth_lat[(size_t)src * 4 + i] : (h16)0.f;
    for (; c < ld_in; ++c) o[c] = (h16)0.f;
}

__global__ __launch_bounds__(256) void posterior_sample_kernel(const h16* __restrict__ moments, int ldm,
                                                               const float* __restrict__ noise, int B, int hw, float scaling,
                                                               float* __restrict__ lat) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * hw) return;
    const int b = idx / hw, p = idx - b * hw;
    const h16* m = moments + (size_t)idx * ldm;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float mean = (float)m[c];
        float lv = fminf(fmaxf((float)m[4 + c], -30.f), 20.f);
        const float std = __expf(0.5f * lv);
        const float nz = noise ? noise[((size_t)b * 4 + c) * hw + p] : 0.f;
        lat[(size_t)idx * 4 + c] = (mean + std * nz) * scaling;
    }
}

template <typename IT, typename MT>
__global__ __launch_bounds__(256) void prepare_mask_kernel(const IT* __restrict__ image, const MT* __restrict__ mask, int B,
                                                           int HW, h16* __restrict__ masked_img, int ld,
                                                           h16* __restrict__ mask_bin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW, p = idx - b * HW;
    const float mv = (float)mask[idx];
    const float mb = mv >= 0.5f ? 1.f : 0.f;
    mask_bin[idx] = (h16)mb;
    h16* o = masked_img + (size_t)idx * ld;
    for (int c = 0; c < 3; ++c) {
        const float v = (float)image[((size_t)b * 3 + c) * HW + p];
        o[c] = (h16)(mb < 0.5f ? v : 0.f);
    }
    for (int c = 3; c < ld; ++c) o[c] = (h16)0.f;
}

__global__ void mask_down_kernel(const h16* __restrict__ src, int B, int H, int W, int s, h16* __restrict__ dst) {
    const int h = H / s, w = W / s;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * h * w) return;
    const int b = idx / (h * w), r = idx - b * h * w, y = r / w, x = r - y * w;
    dst[idx] = src[((size_t)b * H + y * s) * W + x * s];
}

template <typename PT>
__global__ void pose_down8_kernel(const PT* __restrict__ pose, int B, int C, int H, int W, h16* __restrict__ dst) {
    const int h = H / 8, w = W / 8;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (b, y, x, c)
    if (idx >= B * h * w * C) return;
    const int c = idx % C, r = idx / C, x = r % w, y = (r / w) % h, b = r / (w * h);
    const PT* p = pose + ((size_t)b * C + c) * H * W;
    const int y0 = 8 * y + 3, x0 = 8 * x + 3;
    const float v = 0.25f * ((float)p[(size_t)y0 * W + x0] + (float)p[(size_t)y0 * W + x0 + 1] +
                             (float)p[(size_t)(y0 + 1) * W + x0] + (float)p[(size_t)(y0 + 1) * W + x0 + 1]);
    dst[idx] = (h16)v;
}

// decode_latents' tail (tryon_pipe.py:356-358): (image / 2 + 0.5).clamp(0, 1), channels last.  U8: followed by numpy_to_pil's rounding
// (tryon_pipe.py:357-360 `(images * 255).round().astype("uint8")`, round-half-to-even like numpy) so that the batch leaves the library
// in the dtype the all-gather and the JPEG encoder want
template <bool U8>
__global__ void image_post_kernel(const h16* __restrict__ src, int ld, int n_pix, void* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix) return;
    const h16* s = src + (size_t)idx * ld;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fminf(fmaxf((float)s[c] * 0.5f + 0.5f, 0.f), 1.f);
        if constexpr (U8) reinterpret_cast<unsigned char*>(dst)[(size_t)idx * 3 + c] = (unsigned char)rintf(v * 255.0f);
        else reinterpret_cast<float*>(dst)[(size_t)idx * 3 + c] = v;
    }
}

__global__ __launch_bounds__(256) void mask_mul_kernel(h16* __restrict__ feat, int C, int n_pix, const h16* __restrict__ mask) {
    const size_t octs = (size_t)(C >> 3);
    const size_t total = (size_t)n_pix * octs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / octs;
        const float mk = 1.f - (float)mask[pix];
        h16x8 v = reinterpret_cast<h16x8*>(feat)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * mk);
        reinterpret_cast<h16x8*>(feat)[i] = v;
    }
}

__global__ void fill_f32_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// z' = post_quant_conv(lat * inv_sf) (4x4 1x1 conv + bias; pq = 16 weights + 4 biases), written NHWC padded to ld
__global__ void post_quant_kernel(const float* __restrict__ lat, const float* __restrict__ pq, float inv_sf, int n,
                                  h16* __restrict__ dst, int ld) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 v = reinterpret_cast<const float4*>(lat)[idx];
    const float z[4] = {v.x * inv_sf, v.y * inv_sf, v.z * inv_sf, v.w * inv_sf};
    h16* o = dst + (size_t)idx * ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = pq ? pq[16 + c] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (pq ? pq[c * 4 + j] : (c == j ? 1.f : 0.f)) * z[j];
        o[c] = (h16)s;
    }
    for (int c = 4; c < ld; ++c) o[c] = (h16)0.f;
}

// fp32 latents NCHW [B][4][hw] <-> pixel-major [B][hw][4]
__global__ void lat_nchw_to_pix_kernel(const float* __restrict__ src, int B, int hw, float scale, float* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * hw) return;
    const int b = idx / hw, p = idx - b * hw;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[(size_t)idx * 4 + c] = src[((size_t)b * 4 + c) * hw + p] * scale;
}
__global__ void lat_pix_to_nchw_kernel(const float* __restrict__ src, int B, int hw, float* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * hw) return;
    const int b = idx / hw, p = idx - b * hw;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[((size_t)b * 4 + c) * hw + p] = src[(size_t)idx * 4 + c];
}

inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -11; }

}  // namespace

int ladi_launch_small_linear(const void* x, int x_f32, int ldx, const h16* W, const h16* bias, const h16* res, int ldr, int M,
                             int N, int K, int act, int pre_silu, void* out, int out_f32, int ldo, hipStream_t st) {
    if ((K & 7) || (ldx & 7)) return -1;
    constexpr int MT = 8;
    dim3 grid((N + 3) / 4, (M + MT - 1) / MT);
    if (x_f32)
        hipLaunchKernelGGL((small_linear_kernel<float, MT>), grid, dim3(256), 0, st, (const float*)x, ldx, W, bias, res, ldr, M, N, K,
                           act, pre_silu, out, out_f32, ldo);
    else
        hipLaunchKernelGGL((small_linear_kernel<h16, MT>), grid, dim3(256), 0, st, (const h16*)x, ldx, W, bias, res, ldr, M, N, K,
                           act, pre_silu, out, out_f32, ldo);
    return ok();
}

int ladi_launch_nchw_to_nhwc(const void* src, int src_f32, int n, int C, int H, int W, h16* dst, int ld, hipStream_t st) {
    dim3 grid((H * W + 63) / 64, n);
    if (src_f32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, st, (const float*)src, C, H * W, dst, ld);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<h16>, grid, dim3(256), 0, st, (const h16*)src, C, H * W, dst, ld);
    return ok();
}

int ladi_launch_nhwc_to_nchw(const h16* src, int ld, int n, int C, int H, int W, void* dst, int dst_f32, hipStream_t st) {
    dim3 grid((H * W + 63) / 64, n);
    if (dst_f32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, st, src, ld, C, H * W, (float*)dst);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<h16>, grid, dim3(256), 0, st, src, ld, C, H * W, (h16*)dst);
    return ok();
}

int ladi_launch_timestep_embedding(const float* t, int count, int dim, float* out, hipStream_t st) {
    const int half = dim / 2;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(count, (half + 63) / 64), dim3(64), 0, st, t, count, dim, out);
    return ok();
}

int ladi_launch_sched_step(const StepArgs& a, hipStream_t st) {
    const int total = a.B * a.hw;
    hipLaunchKernelGGL(sched_step_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
    return ok();
}

int ladi_launch_assemble_static(h16* unet_in, int ld_in, int B, int hw, int cfg, const float* latents, const h16* mask_lat,
                                const float* masked_lat, const h16* pose, int pose_ch, const float* cloth_lat, int has_cloth,
                                float lat_scale, hipStream_t st) {
    const int rows = (cfg ? 2 : 1) * B * hw;
    if (9 + pose_ch + (has_cloth ? 4 : 0) > ld_in) return -1;
    hipLaunchKernelGGL(assemble_static_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, unet_in, ld_in, B, hw, cfg, latents,
                       mask_lat, masked_lat, pose, pose_ch, cloth_lat, has_cloth, lat_scale);
    return ok();
}

int ladi_launch_posterior_sample(const h16* moments, int ldm, const float* noise_nchw, int B, int hw, float scaling, float* lat,
                                 hipStream_t st) {
    hipLaunchKernelGGL(posterior_sample_kernel, dim3((B * hw + 255) / 256), dim3(256), 0, st, moments, ldm, noise_nchw, B, hw,
                       scaling, lat);
    return ok();
}

int ladi_launch_prepare_mask(const void* image_nchw, int img_f32, const void* mask_nchw, int mask_f32, int B, int H, int W,
                             h16* masked_img, int ld, h16* mask_bin, hipStream_t st) {
    dim3 grid((B * H * W + 255) / 256);
    const int HW = H * W;
    if (img_f32 && mask_f32)
        hipLaunchKernelGGL((prepare_mask_kernel<float, float>), grid, dim3(256), 0, st, (const float*)image_nchw,
                           (const float*)mask_nchw, B, HW, masked_img, ld, mask_bin);
    else if (img_f32)
        hipLaunchKernelGGL((prepare_mask_kernel<float, h16>), grid, dim3(256), 0, st, (const float*)image_nchw,
                           (const h16*)mask_nchw, B, HW, masked_img, ld, mask_bin);
    else if (mask_f32)
        hipLaunchKernelGGL((prepare_mask_kernel<h16, float>), grid, dim3(256), 0, st, (const h16*)image_nchw,
                           (const float*)mask_nchw, B, HW, masked_img, ld, mask_bin);
    else
        hipLaunchKernelGGL((prepare_mask_kernel<h16, h16>), grid, dim3(256), 0, st, (const h16*)image_nchw,
                           (const h16*)mask_nchw, B, HW, masked_img, ld, mask_bin);
    return ok();
}

int ladi_launch_mask_down(const h16* src, int B, int H, int W, int s, h16* dst, hipStream_t st) {
    const int total = B * (H / s) * (W / s);
    hipLaunchKernelGGL(mask_down_kernel, dim3((total + 255) / 256), dim3(256), 0, st, src, B, H, W, s, dst);
    return ok();
}

int ladi_launch_pose_down8(const void* pose_nchw, int f32, int B, int C, int H, int W, h16* dst, hipStream_t st) {
    const int total = B * (H / 8) * (W / 8) * C;
    if (f32) hipLaunchKernelGGL(pose_down8_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)pose_nchw, B,
                                C, H, W, dst);
    else hipLaunchKernelGGL(pose_down8_kernel<h16>, dim3((total + 255) / 256), dim3(256), 0, st, (const h16*)pose_nchw, B, C, H,
                            W, dst);
    return ok();
}

int ladi_launch_image_post(const h16* src, int ld, int n_pix, void* dst, int dst_u8, hipStream_t st) {
    if (dst_u8) hipLaunchKernelGGL(image_post_kernel<true>, dim3((n_pix + 255) / 256), dim3(256), 0, st, src, ld, n_pix, dst);
    else hipLaunchKernelGGL(image_post_kernel<false>, dim3((n_pix + 255) / 256), dim3(256), 0, st, src, ld, n_pix, dst);
    return ok();
}

int ladi_launch_mask_mul(h16* feat, int C, int n_pix, const h16* mask, hipStream_t st) {
    if (C & 7) return -1;
    hipLaunchKernelGGL(mask_mul_kernel, dim3(2048), dim3(256), 0, st, feat, C, n_pix, mask);
    return ok();
}

// CLIP text embeddings with the pseudo-word splice (encode_text_word_embedding.py:26-38): row (b,t) = token_embedding[ids[b][t]]
// unless first[b] <= t < first[b] + nv (first[b] >= 0: position of sentence b's first '$'), then word_emb[b][t - first[b]]; plus
// the position embedding of t.  One block per row, 8 channels per thread.
__global__ void text_embed_kernel(const int* __restrict__ ids, const int* __restrict__ first, int nv, const h16* __restrict__ tok,
                                  const h16* __restrict__ pos, const h16* __restrict__ wemb, int T, int H, int vocab, h16* __restrict__ out) {
    const int row = blockIdx.x, b = row / T, t = row - b * T;
    const int f = first[b];
    const bool sp = wemb && f >= 0 && t >= f && t < f + nv;
    const h16* src = sp ? wemb + ((size_t)b * nv + (t - f)) * H : tok + (size_t)min(max(ids[row], 0), vocab - 1) * H;
    const h16* pp = pos + (size_t)t * H;
    for (int c = threadIdx.x * 8; c < H; c += blockDim.x * 8) {
        const h16x8 a = *reinterpret_cast<const h16x8*>(src + c);
        const h16x8 p8 = *reinterpret_cast<const h16x8*>(pp + c);
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)((float)a[e] + (float)p8[e]);
        *reinterpret_cast<h16x8*>(out + (size_t)row * H + c) = o;
    }
}
// device side of encode_text_word_embedding.py:12-19,62-65 for ids that already live on the device: per sentence the position of the first
// '$' (vstar) token (-1: none, or no pseudo-words given) and the row of the end-of-text token = first maximum of the ids (torch.argmax).
// One wave per sentence; ids are clamped to the vocabulary by the caller's embedding lookup (no host round trip, hence no host error)
__global__ void text_meta_kernel(const int* __restrict__ ids, int T, int vstar, int use_words, int* __restrict__ first, int* __restrict__ eot) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int f = 0x7fffffff, best = (int)0x80000000, arg = 0x7fffffff;   // best = INT_MIN: a sentence of negative ids still finds its maximum
    for (int t = lane; t < T; t += 64) {
        const int id = ids[(size_t)b * T + t];
        if (id == vstar && t < f) f = t;
        if (id > best) { best = id; arg = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int f2 = __shfl_xor(f, o), b2 = __shfl_xor(best, o), a2 = __shfl_xor(arg, o);
        f = min(f, f2);
        if (b2 > best || (b2 == best && a2 < arg)) { best = b2; arg = a2; }
    }
    if (lane == 0) { first[b] = (use_words && f != 0x7fffffff) ? f : -1; eot[b] = b * T + min(max(arg, 0), T - 1); }   // row index always inside the sentence
}
// dst[i][:] = src[rows[i]][:]  (pooled output: the eot row of every sentence)
__global__ void gather_rows_kernel(const h16* __restrict__ src, const int* __restrict__ rows, int H, h16* __restrict__ dst) {
    const h16* s = src + (size_t)rows[blockIdx.x] * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) dst[(size_t)blockIdx.x * H + c] = s[c];
}

// y = x * scale[c] + shift[c] per channel (inference-mode BatchNorm that FOLLOWS a ReLU: ConvNet_TPS.py:31-43), NHWC fp16, in place ok
__global__ void channel_affine_kernel(const h16* __restrict__ x, int ldx, size_t n_pix, int C, const float* __restrict__ scale,
                                      const float* __restrict__ shift, h16* __restrict__ y, int ldy) {
    const int oc = C / 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix * oc) return;
    const int c8 = (int)(idx % oc);
    const size_t p = idx / oc;
    const h16x8 v = *reinterpret_cast<const h16x8*>(x + p * ldx + c8 * 8);
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)((float)v[e] * scale[c8 * 8 + e] + shift[c8 * 8 + e]);
    *reinterpret_cast<h16x8*>(y + p * ldy + c8 * 8) = o;
}
// FeatureL2Norm (ConvNet_TPS.py:59-66): every pixel's channel vector divided by sqrt(sum of squares + 1e-6); one wave per pixel
__global__ void l2norm_rows_kernel(const h16* __restrict__ x, int ldx, int rows, int C, h16* __restrict__ y, int ldy) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const h16* xr = x + (size_t)row * ldx;
    float ss = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        const h16x8 v = *reinterpret_cast<const h16x8*>(xr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
    }
    ss = wave_sum(ss);
    const float inv = 1.f / sqrtf(ss + 1e-6f);
    for (int c = lane * 8; c < C; c += 512) {
        const h16x8 v = *reinterpret_cast<const h16x8*>(xr + c);
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)((float)v[e] * inv);
        *reinterpret_cast<h16x8*>(y + (size_t)row * ldy + c) = o;
    }
}
// TPSGridGen.forward (ConvNet_TPS.py:172-185): grid[b][y][x] = [phi(p, c_0..c_{N-1}), 1, X, Y] . (inverse_kernel . [coor_b; 0; 0; 0]),
// phi(r^2) = 0.5 r^2 log r^2 (0 at r = 0), p = (X, Y) = (2x/(W-1) - 1, 2y/(H-1) - 1).  N <= 32 control points.
__global__ void tps_grid_kernel(const float* __restrict__ coor, const float* __restrict__ inv, const float* __restrict__ ctrl, int N,
                                int H, int W, float* __restrict__ grid) {
    __shared__ float map[35][2];
    __shared__ float sc[32][2];
    const int b = blockIdx.y, M = N + 3;
    for (int i = threadIdx.x; i < M * 2; i += blockDim.x) {
        const int r = i >> 1, d = i & 1;
        float s = 0.f;
        for (int k = 0; k < N; ++k) s += inv[r * M + k] * coor[((size_t)b * N + k) * 2 + d];
        map[r][d] = s;
    }
    for (int i = threadIdx.x; i < N * 2; i += blockDim.x) sc[i >> 1][i & 1] = ctrl[i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const float X = (float)x * 2.f / (float)(W - 1) - 1.f, Y = (float)y * 2.f / (float)(H - 1) - 1.f;
    float gx = map[N][0] + X * map[N + 1][0] + Y * map[N + 2][0];
    float gy = map[N][1] + X * map[N + 1][1] + Y * map[N + 2][1];
    for (int k = 0; k < N; ++k) {
        const float dx = X - sc[k][0], dy = Y - sc[k][1];
        const float r2 = dx * dx + dy * dy;
        const float phi = r2 > 0.f ? 0.5f * r2 * logf(r2) : 0.f;
        gx += phi * map[k][0];
        gy += phi * map[k][1];
    }
    float* o = grid + ((size_t)b * H * W + p) * 2;
    o[0] = gx; o[1] = gy;
}
int ladi_launch_channel_affine(const h16* x, int ldx, size_t n_pix, int C, const float* scale, const float* shift, h16* y, int ldy, hipStream_t st) {
    if ((C & 7) || (ldx & 7) || (ldy & 7)) return -1;
    const size_t total = n_pix * (size_t)(C / 8);
    hipLaunchKernelGGL(channel_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ldx, n_pix, C, scale, shift, y, ldy);
    return ok();
}
int ladi_launch_l2norm_rows(const h16* x, int ldx, int rows, int C, h16* y, int ldy, hipStream_t st) {
    if ((C & 7) || (ldx & 7) || (ldy & 7)) return -1;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, ldx, rows, C, y, ldy);
    return ok();
}
int ladi_launch_tps_grid(const float* coor, const float* inv, const float* ctrl, int N, int B, int H, int W, float* grid, hipStream_t st) {
    if (N < 1 || N > 32 || H < 2 || W < 2) return -1;
    hipLaunchKernelGGL(tps_grid_kernel, dim3((unsigned)((H * W + 255) / 256), (unsigned)B), dim3(256), 0, st, coor, inv, ctrl, N, H, W, grid);
    return ok();
}

// 2x2 max pooling, NHWC fp16 (nn.MaxPool2d(2) of the refinement UNet, unet_parts.py:33-36); 8 channels per thread
__global__ void maxpool2_kernel(const h16* __restrict__ src, int lds_, int n, int H, int W, int C, h16* __restrict__ dst, int ldd) {
    const int Ho = H / 2, Wo = W / 2, oc = C / 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * Ho * Wo * oc) return;
    const int c8 = (int)(idx % oc);
    size_t p = idx / oc;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const h16* s0 = src + (((size_t)b * H + 2 * oy) * W + 2 * ox) * lds_ + c8 * 8;
    const h16x8 a = *reinterpret_cast<const h16x8*>(s0), bq = *reinterpret_cast<const h16x8*>(s0 + lds_);
    const h16x8 c = *reinterpret_cast<const h16x8*>(s0 + (size_t)W * lds_), d = *reinterpret_cast<const h16x8*>(s0 + (size_t)W * lds_ + lds_);
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)fmaxf(fmaxf((float)a[e], (float)bq[e]), fmaxf((float)c[e], (float)d[e]));
    *reinterpret_cast<h16x8*>(dst + (((size_t)b * Ho + oy) * Wo + ox) * ldd + c8 * 8) = o;
}
// bilinear x2 upsampling with align_corners=True, NHWC fp16 (nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
// unet_parts.py:48): source coordinate = dst * (in - 1) / (out - 1)
__global__ void upsample2x_bilinear_ac_kernel(const h16* __restrict__ src, int lds_, int n, int H, int W, int C, h16* __restrict__ dst, int ldd) {
    const int Ho = 2 * H, Wo = 2 * W, oc = C / 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * Ho * Wo * oc) return;
    const int c8 = (int)(idx % oc);
    size_t p = idx / oc;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
    const float sy = Ho > 1 ? (float)oy * (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)ox * (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const h16* base = src + (size_t)b * H * W * lds_ + c8 * 8;
    const h16x8 v00 = *reinterpret_cast<const h16x8*>(base + ((size_t)y0 * W + x0) * lds_), v01 = *reinterpret_cast<const h16x8*>(base + ((size_t)y0 * W + x1) * lds_);
    const h16x8 v10 = *reinterpret_cast<const h16x8*>(base + ((size_t)y1 * W + x0) * lds_), v11 = *reinterpret_cast<const h16x8*>(base + ((size_t)y1 * W + x1) * lds_);
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float top = (float)v00[e] + ((float)v01[e] - (float)v00[e]) * fx;
        const float bot = (float)v10[e] + ((float)v11[e] - (float)v10[e]) * fx;
        o[e] = (h16)(top + (bot - top) * fy);
    }
    *reinterpret_cast<h16x8*>(dst + (((size_t)b * Ho + oy) * Wo + ox) * ldd + c8 * 8) = o;
}
int ladi_launch_maxpool2(const h16* src, int lds_, int n, int H, int W, int C, h16* dst, int ldd, hipStream_t st) {
    if ((C & 7) || (lds_ & 7) || (ldd & 7) || (H & 1) || (W & 1)) return -1;
    const size_t total = (size_t)n * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds_, n, H, W, C, dst, ldd);
    return ok();
}
int ladi_launch_upsample2x_bilinear_ac(const h16* src, int lds_, int n, int H, int W, int C, h16* dst, int ldd, hipStream_t st) {
    if ((C & 7) || (lds_ & 7) || (ldd & 7)) return -1;
    const size_t total = (size_t)n * (2 * H) * (2 * W) * (C / 8);
    hipLaunchKernelGGL(upsample2x_bilinear_ac_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds_, n, H, W, C, dst, ldd);
    return ok();
}

// ViT patch extraction for the bias-free patch-embedding conv (kernel = stride = ps): rows [B][1 + G*G][KP], row 0 (class-token slot)
// and columns >= 3*ps*ps zero; column c*ps*ps + ky*ps + kx = pixel[b][c][gy*ps + ky][gx*ps + kx] (the flatten order of the conv weight)
__global__ void patchify_kernel(const void* __restrict__ px, int in_f32, int S, int ps, int G, int KP, h16* __restrict__ out) {
    const int T = 1 + G * G;
    const int row = blockIdx.x, b = row / T, t = row - b * T;
    h16* o = out + (size_t)row * KP;
    const int nk = 3 * ps * ps;
    for (int k = threadIdx.x; k < KP; k += blockDim.x) {
        float v = 0.f;
        if (t > 0 && k < nk) {
            const int c = k / (ps * ps), r = k - c * ps * ps, ky = r / ps, kx = r - ky * ps;
            const int gy = (t - 1) / G, gx = (t - 1) - gy * G;
            const size_t idx = (((size_t)b * 3 + c) * S + (gy * ps + ky)) * S + gx * ps + kx;
            v = in_f32 ? reinterpret_cast<const float*>(px)[idx] : (float)reinterpret_cast<const h16*>(px)[idx];
        }
        o[k] = (h16)v;
    }
}
int ladi_launch_patchify(const void* px, int in_f32, int B, int S, int ps, int KP, h16* out, hipStream_t st) {
    if (ps <= 0 || S % ps || KP < 3 * ps * ps) return -1;
    const int G = S / ps;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)(B * (1 + G * G))), dim3(256), 0, st, px, in_f32, S, ps, G, KP, out);
    return ok();
}

int ladi_launch_text_embed(const int* ids, const int* first, int nv, const h16* tok, const h16* pos, const h16* wemb, int B, int T,
                           int H, int vocab, h16* out, hipStream_t st) {
    if (H % 8 || vocab <= 0) return -1;
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)(B * T)), dim3(128), 0, st, ids, first, nv, tok, pos, wemb, T, H, vocab, out);
    return ok();
}
int ladi_launch_text_meta(const int* ids, int B, int T, int vstar, int use_words, int* first, int* eot, hipStream_t st) {
    hipLaunchKernelGGL(text_meta_kernel, dim3((unsigned)B), dim3(64), 0, st, ids, T, vstar, use_words, first, eot);
    return ok();
}
int ladi_launch_gather_rows(const h16* src, const int* rows, int n, int H, h16* dst, hipStream_t st) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, src, rows, H, dst);
    return ok();
}

int ladi_launch_fill_f32(float* p, size_t n, float v, hipStream_t st) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(256), dim3(256), 0, st, p, n, v);
    return ok();
}

int ladi_launch_post_quant(const float* lat, const float* pq, float inv_sf, int n, h16* dst, int ld, hipStream_t st) {
    hipLaunchKernelGGL(post_quant_kernel, dim3((n + 255) / 256), dim3(256), 0, st, lat, pq, inv_sf, n, dst, ld);
    return ok();
}
int ladi_launch_lat_nchw_to_pix(const float* src, int B, int hw, float scale, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(lat_nchw_to_pix_kernel, dim3((B * hw + 255) / 256), dim3(256), 0, st, src, B, hw, scale, dst);
    return ok();
}
int ladi_launch_lat_pix_to_nchw(const float* src, int B, int hw, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(lat_pix_to_nchw_kernel, dim3((B * hw + 255) / 256), dim3(256), 0, st, src, B, hw, dst);
    return ok();
}

// ------------------------------------------------------------------------------------------------
// Glue of the warping module (src/inference.py:242-260): antialiased bilinear resize and border-padded grid_sample, NCHW planes.
// ------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float ld_any(const void* p, int f32, size_t i) {
    return f32 ? reinterpret_cast<const float*>(p)[i] : (float)reinterpret_cast<const h16*>(p)[i];
}
__device__ __forceinline__ void st_any(void* p, int f32, size_t i, float v) {
    if (f32) reinterpret_cast<float*>(p)[i] = v; else reinterpret_cast<h16*>(p)[i] = (h16)v;
}

// torchvision.transforms.functional.resize(..., BILINEAR, antialias=True) == aten _upsample_bilinear2d_aa (align_corners=False):
// separable triangle filter whose support is widened by the scale factor when down-sampling; per axis
//   scale = in / out, support = max(scale, 1), centre = scale * (o + 0.5), taps [int(centre - support + 0.5), int(centre + support + 0.5))
//   clipped to the image, weight = max(0, 1 - |(tap + 0.5 - centre) / max(scale, 1)|), normalised to sum 1.
// One thread per output element evaluates the 2-D product of the two 1-D filters in fp32 (<= 7 x 7 taps for the 2.3x reductions here).
// Optional value epilogue `e` (the CLIP image pre-processing of src/inference.py:268-272 in the same pass): v = v * pre_mul + pre_add,
// clamp to [0, 1], floor to 1 / quant steps (quant > 0), then (v - sub[c]) / div[c] with c = plane % C.
__global__ __launch_bounds__(256) void resize_bilinear_aa_kernel(const void* __restrict__ src, int in_f32, int planes, int H, int W,
                                                                 void* __restrict__ dst, int out_f32, int Ho, int Wo, const ResizeEpi e) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)planes * Ho * Wo;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho);
    const size_t pl = idx / ((size_t)Wo * Ho);
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const float supy = fmaxf(sy, 1.f), supx = fmaxf(sx, 1.f);
    const float cy = sy * ((float)oy + 0.5f), cx = sx * ((float)ox + 0.5f);
    const int y0 = max(0, (int)(cy - supy + 0.5f)), y1 = min(H, (int)(cy + supy + 0.5f));
    const int x0 = max(0, (int)(cx - supx + 0.5f)), x1 = min(W, (int)(cx + supx + 0.5f));
    const float iy = 1.f / supy, ix = 1.f / supx;
    float wys = 0.f, wxs = 0.f;
    for (int y = y0; y < y1; ++y) wys += fmaxf(0.f, 1.f - fabsf(((float)y - cy + 0.5f) * iy));
    for (int x = x0; x < x1; ++x) wxs += fmaxf(0.f, 1.f - fabsf(((float)x - cx + 0.5f) * ix));
    const size_t base = pl * (size_t)H * W;
    float acc = 0.f;
    for (int y = y0; y < y1; ++y) {
        const float wy = fmaxf(0.f, 1.f - fabsf(((float)y - cy + 0.5f) * iy));
        float row = 0.f;
        for (int x = x0; x < x1; ++x)
            row += fmaxf(0.f, 1.f - fabsf(((float)x - cx + 0.5f) * ix)) * ld_any(src, in_f32, base + (size_t)y * W + x);
        acc += wy * row;
    }
    float v = acc / (wys * wxs);
    if (e.on) {
        v = fminf(fmaxf(v * e.pre_mul + e.pre_add, 0.f), 1.f);
        // CLIPImageProcessor (transformers 4.27.3, the version the reference pins) sends a float image through to_pil_image on its way to
        // `resize`: (x * 255).astype(uint8) -- a truncation -- and rescales by 1 / 255 afterwards, so the pixel values the vision encoder sees
        // are floor-quantised to 8 bits
        if (e.quant > 0.f) v = floorf(v * e.quant) / e.quant;
        const int c = (int)(pl % (size_t)e.C);
        v = (v - e.sub[c]) / e.div[c];
    }
    st_any(dst, out_f32, idx, v);
}

// F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=False) (src/inference.py:260): x = ((g + 1) * size - 1) / 2
// clipped to [0, size - 1]; the four corners are weighted bilinearly, corners outside the image (only possible with zero weight after
// the clip) are skipped.  One thread per output pixel walks the channels (the grid sample is shared by them).
__global__ __launch_bounds__(256) void grid_sample_border_kernel(const void* __restrict__ src, int in_f32, int B, int C, int H, int W,
                                                                 const float* __restrict__ grid, int Ho, int Wo,
                                                                 void* __restrict__ dst, int out_f32) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo;
    if (idx >= total) return;
    const size_t b = idx / ((size_t)Ho * Wo), pix = idx % ((size_t)Ho * Wo);
    const float gx = grid[idx * 2], gy = grid[idx * 2 + 1];
    float x = ((gx + 1.f) * (float)W - 1.f) * 0.5f, y = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = x - xf, ty = y - yf;
    const float wnw = (1.f - tx) * (1.f - ty), wne = tx * (1.f - ty), wsw = (1.f - tx) * ty, wse = tx * ty;
    const bool inx1 = x1 < W, iny1 = y1 < H;
    for (int c = 0; c < C; ++c) {
        const size_t base = (b * C + c) * (size_t)H * W;
        float v = wnw * ld_any(src, in_f32, base + (size_t)y0 * W + x0);
        if (inx1) v += wne * ld_any(src, in_f32, base + (size_t)y0 * W + x1);
        if (iny1) v += wsw * ld_any(src, in_f32, base + (size_t)y1 * W + x0);
        if (inx1 && iny1) v += wse * ld_any(src, in_f32, base + (size_t)y1 * W + x1);
        st_any(dst, out_f32, (b * C + c) * (size_t)Ho * Wo + pix, v);
    }
}

}  // namespace

int ladi_launch_resize_bilinear_aa(const void* src, int in_f32, int planes, int H, int W, void* dst, int out_f32, int Ho, int Wo,
                                   hipStream_t st, const ResizeEpi* epi) {
    if (planes <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return -1;
    ResizeEpi e; e.on = 0; e.C = 1; e.pre_mul = 1.f; e.pre_add = 0.f; e.quant = 0.f;
    for (int i = 0; i < 4; ++i) { e.sub[i] = 0.f; e.div[i] = 1.f; }
    if (epi) { e = *epi; if (e.C < 1 || e.C > 4) return -1; }
    const size_t total = (size_t)planes * Ho * Wo;
    hipLaunchKernelGGL(resize_bilinear_aa_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, in_f32, planes, H, W, dst,
                       out_f32, Ho, Wo, e);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_grid_sample_border(const void* src, int in_f32, int B, int C, int H, int W, const float* grid, int Ho, int Wo, void* dst,
                                   int out_f32, hipStream_t st) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return -1;
    const size_t total = (size_t)B * Ho * Wo;
    hipLaunchKernelGGL(grid_sample_border_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, in_f32, B, C, H, W, grid,
                       Ho, Wo, dst, out_f32);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

// Shader-clock probe (bench.py `clock`): ONE wave spins for `wall_ticks` ticks of the constant 100 MHz counter (s_memrealtime) and
// reports how many shader cycles (s_memtime) went by -- launched on a side stream next to the kernels being measured, it reads the clock
// the chip actually sustains under that load (DVFS: 2.4 GHz nominal, ~1.9 GHz under dense MFMA work, MI355X_MICROARCH.md).
__global__ void clock_probe_kernel(unsigned long long wall_ticks, unsigned long long* out) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(32); w1 = wall_clock64(); }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
int ladi_launch_clock_probe(unsigned long long wall_ticks, unsigned long long* out2, hipStream_t st) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, st, wall_ticks, out2);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// dst = src * s (fp16 NHWC rows of C channels; the VAE range guard's scaled copies of the EMASC skips)
__global__ void scale_h16_kernel(const h16* __restrict__ src, int lds_, h16* __restrict__ dst, int ldd, size_t n_pix, int C8, float s) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix * C8) return;
    const size_t p = idx / C8; const int c = (int)(idx % C8) * 8;
    h16x8 v = *reinterpret_cast<const h16x8*>(src + p * lds_ + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * s);
    *reinterpret_cast<h16x8*>(dst + p * ldd + c) = v;
}
int ladi_launch_scale_h16(const h16* src, int lds_, h16* dst, int ldd, size_t n_pix, int C, float s, hipStream_t st) {
    if ((C & 7) || (lds_ & 7) || (ldd & 7)) return -1;
    const size_t total = n_pix * (C / 8);
    hipLaunchKernelGGL(scale_h16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds_, dst, ldd, n_pix, C / 8, s);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
