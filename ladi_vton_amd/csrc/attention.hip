// Fused (flash-style) attention for head_dim = 64 on gfx950 MFMA, plus a single-query VALU kernel.
//
// flash_attn64: used by all 16 self- and 16 cross-attention layers of the UNet.
//   Block = 4 waves = 128 queries of one (sample, head); each wave owns 32 queries.
//   Swapped product S^T = K Q^T (A = K tile from LDS, B = Q held in registers) so that each lane owns ONE
//   query column: the online-softmax max / sum / rescale are lane-local (one cross-half shuffle), and the
//   exponentiated P registers are directly the B operand of O^T += V^T P^T (k index permutation
//   key = 4*half + (j&3) + 8*(j>>2) is applied to the V^T A-operand reads instead of moving P).
//   K stage: [128 keys][64 d] fp16, 16-byte chunks XOR-swizzled; the V stage is transposed while staging
//   (4 keys x 8 d micro-tiles per thread, 8-byte LDS writes) into V^T[d][128 keys + 4 pad]; two 64-key compute
//   sub-tiles per barrier pair.  Softmax: raw v_exp_f32, masking only on the ragged last sub-tile, and the O/l
//   rescale deferred until the running max grows by more than 2^8 (wave-uniform decision; P <= 256 fits fp16).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int KV_STAGE = 128;        // keys staged per barrier pair (two 64-key compute sub-tiles)
constexpr int VT_LD = KV_STAGE + 4;  // halves per V^T row (8-byte aligned rows, conflict-free 8-byte column reads)
constexpr float RESCALE_THR = 8.0f;  // defer the O/l rescale until the running max grows by more than 2^8 (P <= 256 fits fp16)

template <int V> struct SubIdx { static constexpr int value = V; };
__device__ __forceinline__ int kswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

__global__ __launch_bounds__(256, 3) void flash_attn64_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) h16 sK[KV_STAGE * 64];
    __shared__ __attribute__((aligned(16))) h16 sVt[64 * VT_LD];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, n = blockIdx.z;
    const int qbase = blockIdx.x * 128 + wave * 32;

    const h16* __restrict__ qp = a.q + (size_t)n * a.sq + head * 64;
    const h16* __restrict__ kp = a.k + (size_t)n * a.sk + head * 64;
    const h16* __restrict__ vp = a.v + (size_t)n * a.sv + head * 64;

    const float qscale = a.scale * 1.4426950408889634f;
    // ---- Q fragments (B operand): lane = query l31, k-half hh
    h16x8 qf[4];
    {
        const int qrow = qbase + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qrow < a.Nq) v = *reinterpret_cast<const h16x8*>(qp + (size_t)qrow * a.ldq + ks * 16 + hh * 8);
            // fold softmax scale * log2(e) into Q once (0.125 * log2e: the scaled value keeps full fp16 relative precision)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * qscale);
            qf[ks] = v;
        }
    }

    f32x16 o_acc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;

    // staging: every thread moves 4 x 16 B of K (rows r0 + 32 i) and one 4-key x 8-d micro-tile of V (transposed on the way)
    const int k_r0 = tid >> 3, k_c8 = tid & 7;
    const int v_quad = tid & 31, v_oct = tid >> 5;   // keys 4*quad.., d = 8*oct..
    uint4 kst0, kst1, kst2, kst3, vst0, vst1, vst2, vst3;   // named scalars: arrays captured by lambdas were demoted to scratch

#define FA_KLOAD(dst, i)                                                                                   \
    {                                                                                                      \
        const int key = key0_ + k_r0 + 32 * (i);                                                           \
        const int kc = key < a.Nk ? key : a.Nk - 1; /* clamp (masked later) instead of branching */        \
        dst = *reinterpret_cast<const uint4*>(kp + (size_t)kc * a.ldk + k_c8 * 8);                         \
    }
#define FA_VLOAD(dst, i)                                                                                   \
    {                                                                                                      \
        const int key = key0_ + v_quad * 4 + (i);                                                          \
        dst = make_uint4(0, 0, 0, 0); /* V rows past Nk must be exact zeros (0 * garbage could be NaN) */  \
        if (key < a.Nk) dst = *reinterpret_cast<const uint4*>(vp + (size_t)key * a.ldv + v_oct * 8);       \
    }
#define FA_GLOAD(k0v)                                                                                      \
    {                                                                                                      \
        const int key0_ = (k0v);                                                                           \
        FA_KLOAD(kst0, 0) FA_KLOAD(kst1, 1) FA_KLOAD(kst2, 2) FA_KLOAD(kst3, 3)                            \
        FA_VLOAD(vst0, 0) FA_VLOAD(vst1, 1) FA_VLOAD(vst2, 2) FA_VLOAD(vst3, 3)                            \
    }
#define FA_VROW(e, comp, odd)                                                                              \
    {                                                                                                      \
        uint2 t_;                                                                                          \
        if (odd) { t_.x = (vst0.comp >> 16) | (vst1.comp & 0xffff0000u); t_.y = (vst2.comp >> 16) | (vst3.comp & 0xffff0000u); } \
        else { t_.x = (vst0.comp & 0xffffu) | (vst1.comp << 16); t_.y = (vst2.comp & 0xffffu) | (vst3.comp << 16); }           \
        *reinterpret_cast<uint2*>(sVt + (v_oct * 8 + (e)) * VT_LD + v_quad * 4) = t_;                      \
    }
#define FA_LSTORE()                                                                                        \
    {                                                                                                      \
        *reinterpret_cast<uint4*>(sK + kswz(k_r0, k_c8)) = kst0;                                           \
        *reinterpret_cast<uint4*>(sK + kswz(k_r0 + 32, k_c8)) = kst1;                                      \
        *reinterpret_cast<uint4*>(sK + kswz(k_r0 + 64, k_c8)) = kst2;                                      \
        *reinterpret_cast<uint4*>(sK + kswz(k_r0 + 96, k_c8)) = kst3;                                      \
        FA_VROW(0, x, 0) FA_VROW(1, x, 1) FA_VROW(2, y, 0) FA_VROW(3, y, 1)                                \
        FA_VROW(4, z, 0) FA_VROW(5, z, 1) FA_VROW(6, w, 0) FA_VROW(7, w, 1)                                \
    }

    const int nstages = (a.Nk + KV_STAGE - 1) / KV_STAGE;
    FA_GLOAD(0)
    for (int t = 0; t < nstages; ++t) {
        __syncthreads();  // previous stage fully consumed
        FA_LSTORE()
        __syncthreads();
        if (t + 1 < nstages) FA_GLOAD((t + 1) * KV_STAGE)
        auto process = [&](auto SubC) __attribute__((always_inline)) {
            constexpr int sub = decltype(SubC)::value;
            const int key0 = t * KV_STAGE + sub * 64;
            // ---- S^T = K Q^T : two 32-key blocks
            f32x16 s_acc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int r = sub * 64 + kb * 32 + l31;
                    const h16x8 kf = *reinterpret_cast<const h16x8*>(sK + kswz(r, ks * 2 + hh));
                    s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s_acc[kb], 0, 0, 0);
                }
            }
            // ---- online softmax (lane-local query); masking only on the ragged last sub-tile
            float mt = -1.0e30f;
            if (key0 + 64 > a.Nk) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const float sv = (key < a.Nk) ? s_acc[kb][r] : -1.0e30f;
                        s_acc[kb][r] = sv;
                        mt = fmaxf(mt, sv);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s_acc[kb][r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            // deferred rescale (wave-uniform decision): everything still at the old max is scaled exactly once
            if (__any(mt > m_run + RESCALE_THR)) {
                const float m_new = fmaxf(m_run, mt);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
            }
            float psum = 0.f;
            h16x8 pf[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s_acc[kb][r] - m_run);
                    psum += p;
                    pf[kb][r >> 3][r & 7] = (h16)p;
                }
            l_run += psum;
            // ---- O^T += V^T P^T
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int kofs = sub * 64 + kb * 32 + k2 * 16 + 4 * hh;
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
        };
        process(SubIdx<0>{});
        if (t * KV_STAGE + 64 < a.Nk) process(SubIdx<1>{});
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
