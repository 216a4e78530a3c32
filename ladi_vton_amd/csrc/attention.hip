// Fused (flash-style) attention for head_dim = 64 on gfx950 MFMA, plus a single-query VALU kernel.
//
// flash_attn64: used by all 16 self- and 16 cross-attention layers of the UNet.
//   Block = 4 waves; each wave owns QB x 32 queries of one (sample, head) (QB = 2 for the long self-attention layers: every
//   K / V^T fragment read from LDS then feeds two MFMAs).
//   Swapped product S^T = K Q^T (A = K tile from LDS, B = Q held in registers) so that each lane owns ONE
//   query column: the online-softmax max / sum / rescale are lane-local (one cross-half shuffle), and the
//   exponentiated P registers are directly the B operand of O^T += V^T P^T (k index permutation
//   key = 4*half + (j&3) + 8*(j>>2) is applied to the V^T A-operand reads instead of moving P).
//   K stage: [128 keys][64 d] fp16, 16-byte chunks XOR-swizzled, double-buffered and filled by LDS-DMA one stage ahead
//   (no staging registers); the V stage is transposed while staging (4 keys x 8 d micro-tiles per thread, 8-byte LDS
//   writes) into V^T[d][128 keys + 4 pad]; two 64-key compute sub-tiles per barrier pair.
//   Softmax: the running reference m_run is folded into the S accumulator init (the MFMA delivers s - m_run, no
//   per-element subtract), raw v_exp_f32, masking only on the ragged last sub-tile, and the O/l rescale deferred
//   until the tile maximum exceeds the reference by more than 2^8 (wave-uniform decision; P <= 256 fits fp16).
#include "common.h"
#include "kernels.h"
#pragma clang diagnostic ignored "-Winline-asm"   // the "m0" clobber of FA_DMA (see there)
#include <cstdlib>

namespace {

constexpr float RESCALE_THR = 8.0f;  // defer the O/l rescale until the running max grows by more than 2^8 (P <= 256 fits fp16)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __fp16 f16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));   // operand type of the transposed LDS read builtin
typedef __attribute__((address_space(3))) f16x4v* lds_f16x4_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct SubIdx { static constexpr int value = V; };
__device__ __forceinline__ int kswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// QB = 32-query blocks per wave: every K / V^T fragment read from LDS feeds QB MFMAs (QB = 2 halves the LDS read traffic per
// MFMA for the long self-attention layers; QB = 1 keeps more blocks in flight for short sequences)
// KV_STAGE = keys staged per barrier (one or two 64-key compute sub-tiles); WPS = waves per SIMD the register budget is cut for
template <int QB, int KV_STAGE, int WPS>
__global__ __launch_bounds__(256, WPS) void flash_attn64_kernel(const AttnArgs a) {
    // dynamic LDS (64 KiB): K double buffer | V double buffer, both filled by LDS-DMA one stage ahead
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    h16* sKbuf = reinterpret_cast<h16*>(fa_smem);                              // [2][KV_STAGE][64]
    h16* sVbuf = sKbuf + 2 * KV_STAGE * 64;                                    // [2][KV_STAGE][64]

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, hh = lane >> 5;
    // 1-D launch, XCD-aware: workgroup b runs on XCD b % 8 (round-robin dispatch), and all query tiles of one (sample, head) are
    // placed on the SAME XCD so its K / V (re-read by every query tile) are fetched into one L2 instead of eight
    int qt, head, n;
    {
        const int T = a.qtiles, G = a.heads * a.n;
        const int b = blockIdx.x;
        int g, t;
        if (a.xcd_map && (G & 7) == 0) { const int x = b & 7, sl = b >> 3; g = x + 8 * (sl / T); t = sl - (sl / T) * T; }
        else { g = b / T; t = b - g * T; }
        qt = t; head = g % a.heads; n = g / a.heads;
    }
    const int qbase = qt * (128 * QB) + wave * (32 * QB);

    const h16* __restrict__ qp = a.q + (size_t)n * a.sq + head * 64;
    const h16* __restrict__ kp = a.k + (size_t)n * a.sk + head * 64;
    const h16* __restrict__ vp = a.v + (size_t)n * a.sv + head * 64;

    const float qscale = a.scale * 1.4426950408889634f;
    // ---- Q fragments (B operand): lane = query l31, k-half hh
    h16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = qbase + qb * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qrow < a.Nq) v = *reinterpret_cast<const h16x8*>(qp + (size_t)qrow * a.ldq + ks * 16 + hh * 8);
            // fold softmax scale * log2(e) into Q once (0.125 * log2e: the scaled value keeps full fp16 relative precision)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * qscale);
            qf[qb][ks] = v;
        }
    }

    f32x16 o_acc[QB][2];
    // The reference m_run is kept INTEGER-valued (rounded up when it moves), so -m_run = -(1024 a) - b with |b| <= 512 is exact in two
    // fp16 values, and it enters the score accumulators through one extra MFMA k-step (K side: ones in k slots 0 and 1) instead of 16
    // v_mov per accumulator: the kernel is VALU-bound (10 VALU per MFMA before this change, PMC round 3), the MFMA pipe has the room.
    h16x8 ones_f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hh == 0) { ones_f[0] = (h16)1.f; ones_f[1] = (h16)1.f; }
    h16x8 mneg[QB];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mneg[qb] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        m_run[qb] = 0.f;   // reference offset of the exponent; set by the first sub-tile
        l_run[qb] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[qb][d][r] = 0.f;
    }

    // staging (round 3).  K AND V: 4 x 16 B per thread and tensor by LDS-DMA (buffer_load ... lds) straight into the next stage's
    // buffers, row-major [key][64 d] like global memory.  The LDS image of a DMA is lane-linear, so the XOR swizzle is applied to
    // the per-lane SOURCE chunk (same scheme as igemm.hip): K chunk ^= (row >> 1) & 7 (read back as 16-byte B fragments), V chunk ^=
    // row & 7 (read back TRANSPOSED by ds_read_b64_tr_b16, which hands a lane four keys of one d: the V^T fragment of the PV
    // product without the register transposition that cost 21 % of the round-2 kernel; profiles/r03_attention_ablation.txt).
    const int k_r0 = tid >> 3, k_c8 = tid & 7;
    const int k_clog = k_c8 ^ ((k_r0 >> 1) & 7);     // rows k_r0 + 32 i share (row >> 1) & 7
    const int v_clog = k_c8 ^ (k_r0 & 7);            // rows k_r0 + 32 i share row & 7
    auto make_rsrc = [](const h16* p) {   // raw buffer descriptor (base, stride 0, 2 GiB - 1 records, dword data format) in SGPRs
        const unsigned long long b = (unsigned long long)p;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
        r[2] = 0x7FFFFFFFu;
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 rk = make_rsrc(kp), rv = make_rsrc(vp);
    // transposed read: in each 16-lane group lane k hands in the address of 4 d of key (k >> 2); it gets back d = (its index) of 4
    // keys (measured layout: tools/experiments/ds_read_tr16.hip).  Group g: d half g & 1, key half hh = g >> 1 (keys +4 hh, the
    // accumulator-row permutation of the S^T tile).  The swizzle term row & 7 = 4 hh + (k >> 2) is a lane constant.
    int tr_off[2];
    {
        const int tk = lane & 15, trow = 4 * hh + (tk >> 2), c0 = 2 * ((lane >> 4) & 1) + ((tk & 3) >> 1);
#pragma unroll
        for (int d = 0; d < 2; ++d) tr_off[d] = trow * 64 + (((d * 4 + c0) ^ trow) << 3) + 4 * (tk & 1);
    }
    // The DMA is issued from inline asm on purpose: with the builtin the compiler knows an LDS-DMA is in flight and, unable to prove
    // that the transposed reads of THIS stage do not alias the buffers being filled for the next one, puts s_waitcnt vmcnt(0) in
    // front of the first V read of the stage: the prefetch then costs a full memory round trip in the middle of every stage.
    // Synchronisation is explicit instead (vmcnt(0) + barrier at the top of each stage).
    const unsigned lds_k0 = (unsigned)(size_t)(lds_ptr_t)sKbuf, lds_v0 = (unsigned)(size_t)(lds_ptr_t)sVbuf;
#define FA_DMA(rsrc, ld, clog, lds0, buf, i)                                                               \
    {                                                                                                      \
        const int key = key0_ + k_r0 + 32 * (i);                                                           \
        const int kc = key < a.Nk ? key : a.Nk - 1; /* clamp: scores of keys >= Nk are masked, P = 0 exactly */ \
        const unsigned m0v = __builtin_amdgcn_readfirstlane((lds0) + (buf) * (KV_STAGE * 128) + (i) * 4096 + wave * 1024); \
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"                          \
                     :: "s"(m0v), "v"((unsigned)((kc * (ld) + (clog) * 8) * 2)), "s"(rsrc) : "memory", "m0"); /* m0 is RESERVED for hipcc (never allocated; the compiler re-materialises it right before each of its own uses), so naming it in the clobber list draws -Winline-asm "reserved registers" -- silenced for this file below; the constraint is stated anyway */ \
    }
#define FA_STAGE(k0v, buf)                                                                                 \
    {                                                                                                      \
        const int key0_ = (k0v);                                                                           \
        FA_DMA(rk, a.ldk, k_clog, lds_k0, buf, 0) FA_DMA(rk, a.ldk, k_clog, lds_k0, buf, 1)                \
        if constexpr (KV_STAGE == 128) { FA_DMA(rk, a.ldk, k_clog, lds_k0, buf, 2) FA_DMA(rk, a.ldk, k_clog, lds_k0, buf, 3) } \
        FA_DMA(rv, a.ldv, v_clog, lds_v0, buf, 0) FA_DMA(rv, a.ldv, v_clog, lds_v0, buf, 1)                \
        if constexpr (KV_STAGE == 128) { FA_DMA(rv, a.ldv, v_clog, lds_v0, buf, 2) FA_DMA(rv, a.ldv, v_clog, lds_v0, buf, 3) } \
    }

    // ONE barrier per 128-key stage: stage t multiplies from buffers t & 1 while the DMA of stage t + 1 fills the other pair.  The
    // barrier at the top of a stage says: my K/V pieces of stage t have landed (vmcnt), everybody's have (barrier), and everybody
    // is done reading the buffers of stage t - 1, which are the ones refilled now.
    const int nstages = (a.Nk + KV_STAGE - 1) / KV_STAGE;
    FA_STAGE(0, 0)
    for (int t = 0; t < nstages; ++t) {
        const h16* sK = sKbuf + (t & 1) * (KV_STAGE * 64);
        const h16* sV = sVbuf + (t & 1) * (KV_STAGE * 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler does not track DMA -> LDS
        __syncthreads();
        if (t + 1 < nstages) FA_STAGE((t + 1) * KV_STAGE, (t + 1) & 1)
        auto process = [&](auto SubC) __attribute__((always_inline)) {
            constexpr int sub = decltype(SubC)::value;
            const int key0 = t * KV_STAGE + sub * 64;
            // (Round 6: hipcc schedules the two products below as [fragment read -> s_waitcnt lgkmcnt(0) -> QB MFMAs] x 8.  Rotating three fragment
            // registers with the order pinned by sched_barrier -- reads two fragments ahead, lgkmcnt(2) -- was built and measured on one box against
            // this form: 3 072-token self-attention 0.252 / 0.249 ms (this form) vs 0.247 / 0.256 ms, i.e. no difference for 20 more registers (226 ->
            // 246): the partner wave on the SIMD already covers the round trips.  profiles/r06_ab_vs_r05.txt; tools/experiments/attn_pipelined_reads.patch)
            // ---- S^T = K Q^T - m_run : two 32-key blocks x QB query blocks (one K fragment read per QB MFMAs).  The running
            // reference m_run of the lane's query is folded into the accumulator init, so the exponent argument comes straight
            // out of the MFMA (no per-element subtract); the very first sub-tile starts from 0 and sets the reference.
            const bool first = key0 == 0;
            f32x16 s_acc[QB][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    s_acc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones_f, mneg[qb], z, 0, 0, 0);   // = -m_run of the lane's query, exactly
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int r = sub * 64 + kb * 32 + l31;
                    const h16x8 kf = *reinterpret_cast<const h16x8*>(sK + kswz(r, ks * 2 + hh));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        s_acc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], s_acc[qb][kb], 0, 0, 0);
                }
            }
            // ---- online softmax (lane-local query); masking only on the ragged last sub-tile
            h16x8 pf[QB][2][2];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float mt = -1.0e30f;   // maximum of THIS LANE's 32 keys relative to m_run (the other k-half of the query sits in lane ^ 32)
                if (key0 + 64 > a.Nk || a.causal) {   // wave-uniform; the mask compares against compile-time key offsets
                    // causal (CLIP text encoder): query i attends to keys <= i; key 0 is visible to every query, so the first
                    // sub-tile always yields a finite reference
                    const int lim = (a.causal ? min(a.Nk, qbase + qb * 32 + l31 + 1) : a.Nk) - key0 - 4 * hh;   // compile-time key offsets below
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float sv = (kb * 32 + (r & 3) + 8 * (r >> 2) < lim) ? s_acc[qb][kb][r] : -1.0e30f;
                            s_acc[qb][kb][r] = sv;
                            mt = fmaxf(mt, sv);
                        }
                } else {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s_acc[qb][kb][r]);
                }
                // deferred rescale (wave-uniform decision): everything still at the old reference is scaled exactly once.  The two
                // lanes of a query exchange their maxima only here (ds_bpermute round trip), not on the common path.
                if (first || __any(mt > RESCALE_THR)) {
                    mt = fmaxf(mt, __shfl_xor(mt, 32));
                    // integer steps; |m_run| <= 64000 keeps the fp16 pair exact (scores of that size mean overflowed fp16 inputs anyway)
                    float delta = ceilf(first ? mt : fmaxf(mt, 0.f));
                    delta = fminf(fmaxf(m_run[qb] + delta, -64000.f), 64000.f) - m_run[qb];
                    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);   // nothing accumulated yet on the first tile
                    m_run[qb] += delta;
                    {
                        const float hi = 1024.f * rintf(m_run[qb] * (1.f / 1024.f));
                        if (hh == 0) { mneg[qb][0] = (h16)(-hi); mneg[qb][1] = (h16)(hi - m_run[qb]); }
                    }
                    l_run[qb] *= alpha;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o_acc[qb][d][r] *= alpha;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s_acc[qb][kb][r] -= delta;
                }
                float psum = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(s_acc[qb][kb][r]);
                        psum += p;
                        pf[qb][kb][r >> 3][r & 7] = (h16)p;
                    }
                l_run[qb] += psum;
            }
            // ---- O^T += V^T P^T (one V^T fragment read per QB MFMAs)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int kbase = sub * 64 + kb * 32 + k2 * 16;   // + 4 hh + (k >> 2) sits in tr_off
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const h16* vb = sV + kbase * 64 + tr_off[d];
                        const f16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4_t)vb);
                        const f16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4_t)(vb + 8 * 64));
                        h16x8 vf;
                        vf[0] = (h16)lo[0]; vf[1] = (h16)lo[1]; vf[2] = (h16)lo[2]; vf[3] = (h16)lo[3];
                        vf[4] = (h16)hi[0]; vf[5] = (h16)hi[1]; vf[6] = (h16)hi[2]; vf[7] = (h16)hi[3];
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
                            o_acc[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb][kb][k2], o_acc[qb][d], 0, 0, 0);
                    }
                }
        };
        process(SubIdx<0>{});
        if constexpr (KV_STAGE == 128) { if (t * KV_STAGE + 64 < a.Nk) process(SubIdx<1>{}); }
    }

    // ---- normalise and store: lane owns query l31, d = 32*dblk + 8g + 4hh + e
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float inv = 1.f / l_tot;
        const int qrow = qbase + qb * 32 + l31;
        if (qrow < a.Nq) {
            h16* op = a.o + (size_t)n * a.so + (size_t)qrow * a.ldo + head * 64;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (h16)(o_acc[qb][d][4 * g + e] * inv);
                    *reinterpret_cast<h16x4*>(op + d * 32 + 8 * g + 4 * hh) = o;
                }
        }
    }
}

// Generic-head-dim attention (HD a multiple of 16, <= 128) for the CLIP ViT-H/14 vision tower (16 heads of d = 80, 257 tokens, once per
// batch).  One wave per 32 queries; the same swapped product / lane-local online softmax / P-as-B-operand scheme as flash_attn64, but with
// plain register staging of 64-key tiles (no DMA ring, no deferred-rescale threshold games beyond the shared reference trick): the
// problem is tiny, simplicity wins.  O^T rows beyond HD (HD = 80 -> 96 rows of V^T) stay zero and are never stored.
template <int HD>
__global__ __launch_bounds__(64) void attn_generic_kernel(const AttnArgs a) {
    constexpr int DK = HD / 16;              // k16 steps of S^T = K Q^T
    constexpr int DB = (HD + 31) / 32;       // 32-row blocks of O^T
    constexpr int KLD = HD + 8;              // halves per sK row (rows stay 16-byte aligned; 176 B stride is conflict-free for b128 reads)
    constexpr int VLD = 64 + 4;              // halves per sVt row
    __shared__ __attribute__((aligned(16))) h16 sK[64 * KLD];
    __shared__ __attribute__((aligned(16))) h16 sVt[DB * 32 * VLD];
    const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, n = blockIdx.z, qbase = blockIdx.x * 32;
    const h16* __restrict__ qp = a.q + (size_t)n * a.sq + head * HD;
    const h16* __restrict__ kp = a.k + (size_t)n * a.sk + head * HD;
    const h16* __restrict__ vp = a.v + (size_t)n * a.sv + head * HD;
    const float qscale = a.scale * 1.4426950408889634f;
    h16x8 qf[DK];
    {
        const int qrow = qbase + l31;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qrow < a.Nq) v = *reinterpret_cast<const h16x8*>(qp + (size_t)qrow * a.ldq + ks * 16 + hh * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * qscale);
            qf[ks] = v;
        }
    }
    f32x16 o_acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    for (int i = lane; i < DB * 32 * VLD; i += 64) sVt[i] = (h16)0.f;   // rows >= HD stay zero for the whole kernel

    for (int key0 = 0; key0 < a.Nk; key0 += 64) {
        __syncthreads();   // previous tile consumed
        {
            const int key = key0 + lane;
            const bool valid = key < a.Nk;
#pragma unroll
            for (int c = 0; c < HD / 8; ++c) {
                h16x8 kk = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (valid) {
                    kk = *reinterpret_cast<const h16x8*>(kp + (size_t)key * a.ldk + c * 8);
                    vv = *reinterpret_cast<const h16x8*>(vp + (size_t)key * a.ldv + c * 8);
                }
                *reinterpret_cast<h16x8*>(sK + lane * KLD + c * 8) = kk;
#pragma unroll
                for (int e = 0; e < 8; ++e) sVt[(c * 8 + e) * VLD + lane] = vv[e];
            }
        }
        __syncthreads();
        const bool first = key0 == 0;
        f32x16 s_acc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float init = -m_run;
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[kb][r] = init;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(sK + (kb * 32 + l31) * KLD + ks * 16 + hh * 8);
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s_acc[kb], 0, 0, 0);
            }
        }
        float mt = -1.0e30f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float sv = (key < a.Nk) ? s_acc[kb][r] : -1.0e30f;
                s_acc[kb][r] = sv;
                mt = fmaxf(mt, sv);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        if (first || __any(mt > RESCALE_THR)) {
            const float delta = first ? mt : fmaxf(mt, 0.f);
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_acc[kb][r] -= delta;
        }
        float psum = 0.f;
        h16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s_acc[kb][r]);
                psum += p;
                pf[kb][r >> 3][r & 7] = (h16)p;
            }
        l_run += psum;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int kofs = kb * 32 + k2 * 16 + 4 * hh;
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const h16* vrow = sVt + (d * 32 + l31) * VLD + kofs;
                    const h16x4 lo = *reinterpret_cast<const h16x4*>(vrow);
                    const h16x4 hi = *reinterpret_cast<const h16x4*>(vrow + 8);
                    h16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][k2], o_acc[d], 0, 0, 0);
                }
            }
    }
    const float inv = 1.f / (l_run + __shfl_xor(l_run, 32));
    const int qrow = qbase + l31;
    if (qrow < a.Nq) {
        h16* op = a.o + (size_t)n * a.so + (size_t)qrow * a.ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dc = d * 32 + 8 * g + 4 * hh;
                if (dc < HD) {
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (h16)(o_acc[d][4 * g + e] * inv);
                    *reinterpret_cast<h16x4*>(op + dc) = o;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Flash attention for ONE WIDE head (HD = 256 / 512): the VAE mid-block AttentionBlock (single head of d = C = 512 over the 64x48 or
// 128x96 latent grid; diffusers 0.14 AttentionBlock).  Scores are never materialised (T^2 fp32 would be 57 MB per
// sample at 512x384 and 0.9 GB at 1024x768).  A 512-wide head does not fit one wave (O^T alone = 256 accumulator registers), so the
// head dimension is split over the 4 waves of a workgroup, which share 32 queries:
//   * wave w owns d in [w*HD/4, (w+1)*HD/4): it holds that slice of Q as MFMA B fragments and the matching rows of O^T;
//   * per tile of 32 keys every wave multiplies its K slice with its Q slice (partial S^T, swapped product as in flash_attn64), the four
//     partial tiles are summed through a double-buffered LDS exchange (ONE barrier per tile), each wave then runs the same lane-local
//     online softmax on the full scores and multiplies its own V^T rows with P.
// V is consumed TRANSPOSED (vt[d][key], produced directly in that layout by the value projection GEMM), so its A fragments are plain
// row segments; K / V^T fragments come straight from L2 (per sample K + V^T = 6 MB at T = 3072).  a.v = V^T, a.ldv = its row stride
// (>= Nk), a.sv = per-sample stride.
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void flash_attn_wide_kernel(const AttnArgs a) {
    constexpr int DS = HD / 4;               // head-dim slice of one wave
    constexpr int DK = DS / 16;              // k16 steps of the partial S^T
    constexpr int DB = DS / 32;              // 32-row blocks of this wave's O^T rows
    __shared__ __attribute__((aligned(16))) float sx[2][4][4][64][4];   // [buffer][wave][reg / 4][lane][reg % 4] partial scores
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y, qbase = blockIdx.x * 32;
    const h16* __restrict__ qp = a.q + (size_t)n * a.sq + w * DS;
    const h16* __restrict__ kp = a.k + (size_t)n * a.sk + w * DS;
    const h16* __restrict__ vt = a.v + (size_t)n * a.sv + (size_t)(w * DS) * a.ldv;
    const float qscale = a.scale * 1.4426950408889634f;
    h16x8 qf[DK];
    {
        const int qrow = qbase + l31;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qrow < a.Nq) v = *reinterpret_cast<const h16x8*>(qp + (size_t)qrow * a.ldq + ks * 16 + hh * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] * qscale);
            qf[ks] = v;
        }
    }
    f32x16 o_acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;
    int buf = 0;
    for (int key0 = 0; key0 < a.Nk; key0 += 32, buf ^= 1) {
        // partial S^T[key][query] over this wave's slice of d
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
            const int key = key0 + l31;
            const bool valid = key < a.Nk;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
                h16x8 kf = {0, 0, 0, 0, 0, 0, 0, 0};
                if (valid) kf = *reinterpret_cast<const h16x8*>(kp + (size_t)key * a.ldk + ks * 16 + hh * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(&sx[buf][w][g][lane][0]) = make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
        __syncthreads();       // the only barrier of the tile: a wave writes buffer buf^1 next, which nobody can still be reading
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = *reinterpret_cast<const float4*>(&sx[buf][0][g][lane][0]);
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const float4 u = *reinterpret_cast<const float4*>(&sx[buf][ww][g][lane][0]);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            s[4 * g] = t.x; s[4 * g + 1] = t.y; s[4 * g + 2] = t.z; s[4 * g + 3] = t.w;
        }
        // online softmax, lane-local per query (the two half-waves hold the two halves of a query's 32 keys)
        float mt = -1.0e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            s[r] = (key < a.Nk) ? s[r] : -1.0e30f;
            mt = fmaxf(mt, s[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        h16x8 pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(s[r] - m_new);
            psum += p;
            pf[r >> 3][r & 7] = (h16)p;
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        // O^T[d][query] += V^T[d][key] P^T[key][query]; contraction order of a 16-key step = this lane's key set (matches pf)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kofs = key0 + k2 * 16 + 4 * hh;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const h16* vrow = vt + (size_t)(d * 32 + l31) * a.ldv + kofs;
                h16x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
                if (kofs + 3 < a.Nk) lo = *reinterpret_cast<const h16x4*>(vrow);
                if (kofs + 11 < a.Nk) hi = *reinterpret_cast<const h16x4*>(vrow + 8);
                h16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[k2], o_acc[d], 0, 0, 0);
            }
        }
    }
    const float inv = 1.f / (l_run + __shfl_xor(l_run, 32));
    const int qrow = qbase + l31;
    if (qrow < a.Nq) {
        h16* op = a.o + (size_t)n * a.so + (size_t)qrow * a.ldo + w * DS;
#pragma unroll
        for (int d = 0; d < DB; ++d)
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
    static const int force_qb = getenv("LADI_ATTN_QB") ? atoi(getenv("LADI_ATTN_QB")) : 0;
    // two query blocks per wave once the 256-query tiles alone oversubscribe the 256 CUs and the K/V stream is long
    // (measured on the UNet shapes, n = 16: self L0 594 -> 623 TFLOP/s, self L1 452 -> 494; short K/V or few tiles: no gain)
    const long long tiles256 = (long long)(a.Nq / 256) * a.heads * a.n;
    const bool qb2 = force_qb ? force_qb == 2 : (tiles256 >= 400 && a.Nk >= 256);
    AttnArgs b = a;
    static const bool no_xcd = getenv("LADI_ATTN_NOXCD") != nullptr;   // A/B switch for the XCD-aware mapping
    b.xcd_map = no_xcd ? 0 : 1;
    auto go = [&](auto kern, int smem, int qpw) {
        struct Seen { const void* fp; unsigned long long devs; };      // the attribute is per (kernel, device)
        static thread_local Seen done[8] = {};
        const void* fp = reinterpret_cast<const void*>(kern);
        Seen* e = nullptr;
        for (auto& d : done) if (d.fp == fp || !d.fp) { e = &d; break; }
        if (!e) return -12;
        e->fp = fp;
        if (ladi_ensure_dyn_lds(fp, smem, e->devs)) return -12;
        b.qtiles = (a.Nq + qpw - 1) / qpw;
        hipLaunchKernelGGL(kern, dim3((unsigned)(b.qtiles * a.heads * a.n)), dim3(256), smem, st, b);
        return hipGetLastError() == hipSuccess ? 0 : -11;
    };
    // long self-attention: 2 query blocks per wave, 128-key stages, two workgroups per CU (64 KiB LDS, 226 VGPRs).  Everything else
    // (cross-attention on 77 keys, the 16x12 / 8x6 levels): 1 query block, 64-key stages, four workgroups per CU (32 KiB, 118 VGPRs);
    // measured n = 16: cross L0 170 -> 183 TFLOP/s, self L2 195 -> 222 against the 128-key form.
    if (qb2) return go(flash_attn64_kernel<2, 128, 2>, 65536, 256);
    return go(flash_attn64_kernel<1, 64, 4>, 32768, 128);
}

int ladi_launch_attn_generic(const AttnArgs& a, int head_dim, hipStream_t st) {
    if ((a.ldq & 7) || (a.ldk & 7) || (a.ldv & 7) || (a.ldo & 3) || a.Nk <= 0 || a.Nq <= 0 || a.causal) return -1;
    dim3 grid((unsigned)((a.Nq + 31) / 32), (unsigned)a.heads, (unsigned)a.n);
    switch (head_dim) {
        case 64: hipLaunchKernelGGL(attn_generic_kernel<64>, grid, dim3(64), 0, st, a); break;
        case 80: hipLaunchKernelGGL(attn_generic_kernel<80>, grid, dim3(64), 0, st, a); break;
        case 96: hipLaunchKernelGGL(attn_generic_kernel<96>, grid, dim3(64), 0, st, a); break;
        case 128: hipLaunchKernelGGL(attn_generic_kernel<128>, grid, dim3(64), 0, st, a); break;
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_attn_wide(const AttnArgs& a, int head_dim, hipStream_t st) {
    // Nk % 4 == 0 keeps the 8-byte V^T row segments whole (masked per segment); heads == 1
    if ((a.ldq & 7) || (a.ldk & 7) || (a.ldv & 3) || (a.ldo & 3) || a.Nk <= 0 || (a.Nk & 3) || a.Nq <= 0 || a.heads != 1 || a.causal) return -1;
    dim3 grid((unsigned)((a.Nq + 31) / 32), (unsigned)a.n);
    switch (head_dim) {
        case 128: hipLaunchKernelGGL(flash_attn_wide_kernel<128>, grid, dim3(256), 0, st, a); break;
        case 256: hipLaunchKernelGGL(flash_attn_wide_kernel<256>, grid, dim3(256), 0, st, a); break;
        case 512: hipLaunchKernelGGL(flash_attn_wide_kernel<512>, grid, dim3(256), 0, st, a); break;
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

int ladi_launch_attn_single_query(const h16* q, int ldq, const h16* k, int ldk, const h16* v, int ldv, h16* o, int ldo,
                                  int n, int heads, int d, int Nk, long long sk, long long sv, float scale, hipStream_t st) {
    if (d > 128 || Nk <= 0) return -1;
    hipLaunchKernelGGL(attn_single_query_kernel, dim3(heads, n), dim3(64), (size_t)Nk * sizeof(float), st, q, ldq, k, ldk, v,
                       ldv, o, ldo, d, Nk, sk, sv, scale);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}
