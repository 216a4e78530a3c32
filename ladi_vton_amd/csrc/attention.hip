// Fused (flash-style) attention for head_dim = 64 on gfx950 MFMA, plus a single-query VALU kernel.
//
// flash_attn64: used by all 16 self- and 16 cross-attention layers of the UNet.
//   Block = 4 waves = 128 queries of one (sample, head); each wave owns 32 queries.
//   Swapped product S^T = K Q^T (A = K tile from LDS, B = Q held in registers) so that each lane owns ONE
//   query column: the online-softmax max / sum / rescale are lane-local (one cross-half shuffle), and the
//   exponentiated P registers are directly the B operand of O^T += V^T P^T (k index permutation
//   key = 4*half + (j&3) + 8*(j>>2) is applied to the V^T A-operand reads instead of moving P).
//   K tile: [64 keys][64 d] fp16, 16-byte chunks XOR-swizzled; V tile is transposed while staging
//   (4 keys x 8 d micro-tiles per thread, 8-byte LDS writes) into V^T[d][64 keys + 4 pad].
#include "common.h"
#include "kernels.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int VT_LD = 68;  // halves per V^T row (136 B: conflict-free 8-byte column reads)

__device__ __forceinline__ int kswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

__global__ __launch_bounds__(256, 2) void flash_attn64_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) h16 sK[KV_TILE * 64];
    __shared__ __attribute__((aligned(16))) h16 sVt[64 * VT_LD];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, n = blockIdx.z;
    const int qbase = blockIdx.x * 128 + wave * 32;

    const h16* __restrict__ qp = a.q + (size_t)n * a.sq + head * 64;
    const h16* __restrict__ kp = a.k + (size_t)n * a.sk + head * 64;
    const h16* __restrict__ vp = a.v + (size_t)n * a.sv + head * 64;

    // ---- Q fragments (B operand): lane = query l31, k-half hh
    h16x8 qf[4];
    {
        const int qrow = qbase + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qrow < a.Nq) v = *reinterpret_cast<const h16x8*>(qp + (size_t)qrow * a.ldq + ks * 16 + hh * 8);
            qf[ks] = v;
        }
    }

    f32x16 o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;
    const float sc = a.scale * 1.4426950408889634f;

    // staging roles: threads 0..127 stage K (4 x 16B chunks), threads 128..255 stage V (4 keys x 8 d)
    const bool is_k = tid < 128;
    const int kt_r0 = (tid & 127) >> 3, kt_c8 = tid & 7;         // K: rows kt_r0 + 16*i
    const int v_quad = (tid & 127) & 15, v_oct = (tid & 127) >> 4;  // V: keys 4*quad.., d = 8*oct..
    uint4 stg[4];

    auto gload = [&](int key0) {
        if (is_k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = key0 + kt_r0 + 16 * i;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (key < a.Nk) v = *reinterpret_cast<const uint4*>(kp + (size_t)key * a.ldk + kt_c8 * 8);
                stg[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = key0 + v_quad * 4 + i;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (key < a.Nk) v = *reinterpret_cast<const uint4*>(vp + (size_t)key * a.ldv + v_oct * 8);
                stg[i] = v;
            }
        }
    };
    auto lstore = [&]() {
        if (is_k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = kt_r0 + 16 * i;
                *reinterpret_cast<uint4*>(sK + kswz(r, kt_c8)) = stg[i];
            }
        } else {
            const h16* s0 = reinterpret_cast<const h16*>(&stg[0]);
            const h16* s1 = reinterpret_cast<const h16*>(&stg[1]);
            const h16* s2 = reinterpret_cast<const h16*>(&stg[2]);
            const h16* s3 = reinterpret_cast<const h16*>(&stg[3]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h16x4 t; t[0] = s0[e]; t[1] = s1[e]; t[2] = s2[e]; t[3] = s3[e];
                *reinterpret_cast<h16x4*>(sVt + (v_oct * 8 + e) * VT_LD + v_quad * 4) = t;
            }
        }
    };

    const int ntiles = (a.Nk + KV_TILE - 1) / KV_TILE;
    gload(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();  // previous tile fully consumed
        lstore();
        __syncthreads();
        if (t + 1 < ntiles) gload((t + 1) * KV_TILE);
        const int key0 = t * KV_TILE;

        // ---- S^T = K Q^T : two 32-key blocks
        f32x16 s_acc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int r = kb * 32 + l31;
                const h16x8 kf = *reinterpret_cast<const h16x8*>(sK + kswz(r, ks * 2 + hh));
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s_acc[kb], 0, 0, 0);
            }
        }
        // ---- online softmax (lane-local query)
        float mt = -1.0e30f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float s = s_acc[kb][r] * sc;
                s = (key < a.Nk) ? s : -1.0e30f;
                s_acc[kb][r] = s;
                mt = fmaxf(mt, s);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        h16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(s_acc[kb][r] - m_new);
                psum += p;
                pf[kb][r >> 3][r & 7] = (h16)p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        // ---- O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int kofs = kb * 32 + k2 * 16 + 4 * hh;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const h16* vrow = sVt + (d * 32 + l31) * VT_LD + kofs;
                    const h16x4 lo = *reinterpret_cast<const h16x4*>(vrow);
                    const h16x4 hi = *reinterpret_cast<const h16x4*>(vrow + 8);
                    h16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][k2], o_acc[d], 0, 0, 0);
                }
            }
    }

    // ---- normalise and store: lane owns query l31, d = 32*dblk + 8g + 4hh + e
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    const int qrow = qbase + l31;
    if (qrow < a.Nq) {
        h16* op = a.o + (size_t)n * a.so + (size_t)qrow * a.ldo + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (h16)(o_acc[d][4 * g + e] * inv);
                *reinterpret_cast<h16x4*>(op + d * 32 + 8 * g + 4 * hh) = o;
            }
    }
}

// One wave per (sample, head): a single query row against Nk keys; head dim d <= 128 (lanes own d and d+64).
__global__ __launch_bounds__(64) void attn_single_query_kernel(const h16* __restrict__ q, int ldq, const h16* __restrict__ k,
                                                               int ldk, const h16* __restrict__ v, int ldv,
                                                               h16* __restrict__ o, int ldo, int d, int Nk, long long sk,
                                                               long long sv, float scale) {
    extern __shared__ float sc_buf[];  // Nk scores
    const int lane = threadIdx.x;
    const int head = blockIdx.x, n = blockIdx.y;
    const h16* qp = q + (size_t)n * ldq + head * d;
    const h16* kp = k + (size_t)n * sk + head * d;
    const h16* vp = v + (size_t)n * sv + head * d;
    const float q0 = lane < d ? (float)qp[lane] : 0.f;
    const float q1 = lane + 64 < d ? (float)qp[lane + 64] : 0.f;
    float m = -3.0e38f;
    for (int j = 0; j < Nk; ++j) {
        const h16* kr = kp + (size_t)j * ldk;
        float s = (lane < d ? q0 * (float)kr[lane] : 0.f) + (lane + 64 < d ? q1 * (float)kr[lane + 64] : 0.f);
        s = wave_sum(s) * scale;
        if (lane == 0) sc_buf[j] = s;
        m = fmaxf(m, s);
    }
    __syncthreads();
    float sum = 0.f, a0 = 0.f, a1 = 0.f;
    for (int j = 0; j < Nk; ++j) {
        const float p = __expf(sc_buf[j] - m);
        sum += p;
        const h16* vr = vp + (size_t)j * ldv;
        if (lane < d) a0 += p * (float)vr[lane];
        if (lane + 64 < d) a1 += p * (float)vr[lane + 64];
    }
    const float inv = 1.f / sum;
    h16* op = o + (size_t)n * ldo + head * d;
    if (lane < d) op[lane] = (h16)(a0 * inv);
    if (lane + 64 < d) op[lane + 64] = (h16)(a1 * inv);
}

}  // namespace

int ladi_launch_flash_attn64(const AttnArgs& a, hipStream_t st) {
    if ((a.ldq & 7) || (a.ldk & 7) || (a.ldv & 7) || (a.ldo & 3) || a.Nk <= 0 || a.Nq <= 0) return -1;
    dim3 grid((a.Nq + 127) / 128, a.heads, a.n);
    hipLaunchKernelGGL(flash_attn64_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_attn_single_query(const h16* q, int ldq, const h16* k, int ldk, const h16* v, int ldv, h16* o, int ldo,
                                  int n, int heads, int d, int Nk, long long sk, long long sv, float scale, hipStream_t st) {
    if (d > 128 || Nk <= 0) return -1;
    hipLaunchKernelGGL(attn_single_query_kernel, dim3(heads, n), dim3(64), (size_t)Nk * sizeof(float), st, q, ldq, k, ldk, v,
                       ldv, o, ldo, d, Nk, sk, sv, scale);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}
