// Stand-alone prototype (round-5 candidate, VERDICT r03 item 4, whole block): LayerNorm -> to_q -> 77-key cross-attention -> to_out + bias +
// residual in ONE kernel for the 64x48 level of the UNet (C = 320, 5 heads of 64): neither Q nor the attention output is ever written.
// It is xattn_q.hip (read that file's header first) plus the output projection folded into the head loop:
//   out^T[320 x 32] += Wo[:, 64 h : 64 h + 64] O_h^T     for every head h,
// with O_h^T / l rounded to fp16 (the rounding point of the stand-alone attention kernel's output) used straight from its accumulator
// blocks as the B operand -- the third link of the accumulator -> B-operand chain.  The ten 32-channel accumulator blocks of the output
// (160 registers) live next to the 80-register pixel panel: one wave per SIMD, up to 512 registers per lane.  Wo is packed per head as
// [HEADS][320][68] halves (the host does it once at load time: row stride 68 = conflict-free 8-byte fragment reads, plain linear LDS-DMA).
// Stage ring per head: Wq rows (2 x 20 KB), then the head's Wo slice (2 x 21.25 KB); the attention of the head sits between them while
// the first Wo stage is already in flight.  Epilogue as linear_xs MODE 1: fp16(acc + bias), + residual, fp16, 64-byte row segments.
//
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/next/xattn_full.hip -o /tmp/xattn_full && /tmp/xattn_full
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int C = 320, HEADS = 5, D = 64, NKP = 96, KLD = 68, VLD = 100;
constexpr int KS = C / 16;                       // k16 steps of the Q projection
constexpr int STAGE = 32 * C * 2;                // one weight stage: 32 rows x 640 B
constexpr int W_DMA = STAGE / 16 / 256;          // 16-byte pieces per thread per stage (5)
constexpr int KTILE = NKP * KLD * 2;             // 13 056 B
constexpr int VTILE = D * VLD * 2;               // 12 800 B
constexpr int KVBUF = 16384;                     // both tiles land in whole 4 KB DMA rounds
constexpr int OLD = 68;                          // row stride of the packed Wo slices (halves)
constexpr int OSTAGE = 160 * OLD * 2;            // half of a head's Wo slice: 160 output channels x 136 B = 21 760 B
constexpr int O_DMA = 6;                         // ... lands in six 4 KB DMA rounds (the descriptor ends with the slice: the rest is zero)
constexpr int SLOT = O_DMA * 4096;               // ring slot: the larger of the two stage kinds
constexpr int PLD = 40;                          // halves per patch row (32 channels + 8 pad)
constexpr int PATCH = 32 * PLD * 2;
constexpr int NST = 4;                           // ring depth: ONE workgroup per CU, nobody else covers a stage's round trip (a stage is consumed
                                                 // in ~0.3 us): three stages (~70 KB) in flight; the K / V^T tiles are single-buffered
constexpr int NSTAGE = 4 * HEADS;
constexpr int SMEM = NST * SLOT + 2 * KVBUF + 4 * PATCH;

struct Args {
    const h16* x; const h16* ln_g; const h16* ln_b; float ln_eps;
    const h16* Wq;                               // [C][C] row-major (out, in), no bias (diffusers Attention.to_q)
    const h16* Kp; const h16* Vt;                // [n][HEADS][96][68] , [n][HEADS][64][100]
    const h16* Wo; const h16* bo;                // to_out: packed [HEADS][C][68] (Wo[q][64 h + d] at [h][q][d]) , bias [C]
    const h16* res;                              // residual [P][C] (the block input)
    h16* out;                                    // [P][C] = res + to_out(attention)
    int P, T, nk; float scale;
};

#define VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform; clamping down is conservative; the
// counter has 6 bits)
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n < 48 ? n : 48) {
        VM_CASE(1) VM_CASE(2) VM_CASE(3) VM_CASE(4) VM_CASE(5) VM_CASE(6) VM_CASE(7) VM_CASE(8) VM_CASE(9) VM_CASE(10) VM_CASE(11) VM_CASE(12)
        VM_CASE(13) VM_CASE(14) VM_CASE(15) VM_CASE(16) VM_CASE(17) VM_CASE(18) VM_CASE(19) VM_CASE(20) VM_CASE(21) VM_CASE(22) VM_CASE(23)
        VM_CASE(24) VM_CASE(25) VM_CASE(26) VM_CASE(27) VM_CASE(28) VM_CASE(29) VM_CASE(30) VM_CASE(31) VM_CASE(32) VM_CASE(33) VM_CASE(34)
        VM_CASE(35) VM_CASE(36) VM_CASE(37) VM_CASE(38) VM_CASE(39) VM_CASE(40) VM_CASE(41) VM_CASE(42) VM_CASE(43) VM_CASE(44) VM_CASE(45)
        VM_CASE(46) VM_CASE(47) VM_CASE(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int V> struct IntC { static constexpr int value = V; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IntC<I>{}); static_for<I + 1, N>(f); }
}

__global__ __launch_bounds__(256, 1) void xattn_full_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* kbuf = smem + NST * SLOT;              // [KVBUF]
    char* vbuf = kbuf + KVBUF;                   // [KVBUF]
    char* patch_base = vbuf + KVBUF;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int p0 = blockIdx.x * 128 + wave * 32;
    const int n = (blockIdx.x * 128) / a.T;      // a workgroup's 128 pixels lie in one sample (T % 128 == 0)

    // ---- weight DMA (linear_xs.hip): LDS position i = j*256 + tid (16-byte units) -> row i/40, physical chunk i%40, XOR swizzle on the source
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Wq), 0, (unsigned)(C * C * 2), 0x00020000);
    unsigned wrel[W_DMA];
#pragma unroll
    for (int j = 0; j < W_DMA; ++j) {
        const int i = j * 256 + tid;
        const int row = i / (C / 8), cph = i - row * (C / 8);
        const int c = cph ^ ((row >> 1) & 7);
        wrel[j] = (unsigned)((row * C + c * 8) * 2);
    }
    int vm_issued = 0;                           // running count of this wave's VMEM operations (wave-uniform)
    // stage (h, kind): kind 0 / 1 = rows [64 h + 32 kind, + 32) of Wq (swizzled, 5 pieces per thread); kind 2 / 3 = output channels
    // [160 (kind - 2), + 160) of head h's packed Wo slice (linear, 6 rounds)
    auto issue_stage = [&](int t, int slot) {    // stage t = (head t / 4, kind t % 4)
        const int h = t >> 2, kind = t & 3;
        char* dst = ring + slot * SLOT + wave * 1024;
        if (kind < 2) {
#pragma unroll
            for (int j = 0; j < W_DMA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + j * 4096), 16, wrel[j] + (unsigned)((2 * h + kind) * STAGE), 0, 0, 0);
            vm_issued += W_DMA;
        } else {
            const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<h16*>(a.Wo) + ((size_t)h * C + (kind - 2) * 160) * OLD, 0, (unsigned)OSTAGE, 0x00020000);
#pragma unroll
            for (int j = 0; j < O_DMA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rso, (lds_ptr_t)(dst + j * 4096), 16, (unsigned)((j * 256 + tid) * 16), 0, 0, 0);
            vm_issued += O_DMA;
        }
    };
    // ---- K / V^T tiles of head h: linear copies, 4 rounds of 256 x 16 B each; the descriptors end with the tile, the rest reads as zero
    auto issue_kv = [&](int h) {
        const size_t g = (size_t)n * HEADS + h;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Kp) + g * (NKP * KLD), 0, (unsigned)KTILE, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Vt) + g * (D * VLD), 0, (unsigned)VTILE, 0x00020000);
        char* kd = kbuf + wave * 1024;
        char* vd = vbuf + wave * 1024;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)(kd + r * 4096), 16, (unsigned)((r * 256 + tid) * 16), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(vd + r * 4096), 16, (unsigned)((r * 256 + tid) * 16), 0, 0, 0);
        }
        vm_issued += 8;
    };

    // ---- the wave's pixel panel as MFMA B fragments: lane = pixel l31, k half hh
    h16x8 xf[KS];
    {
        const h16* xp = a.x + (size_t)(p0 + l31) * C + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const h16x8*>(xp + ks * 16);
    }
    // marks: mk[i] = value of vm_issued right after stage (current + i) was issued; the ring keeps NST - 1 stages in flight
    int mk[NST - 1];
    issue_kv(0);
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) { issue_stage(i, i); mk[i] = vm_issued; }
    int rd_slot = 0, wr_slot = NST - 1;          // slot of the stage being multiplied / of the next stage to issue
    // ---- LayerNorm of the panel (same arithmetic and rounding point as layernorm_kernel / linear_xs PRE = 1)
    {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto pin = [](h16x8& v) {
            i32x4 t = __builtin_bit_cast(i32x4, v);
            asm volatile("" : "+v"(t));
            v = __builtin_bit_cast(h16x8, t);
        };
        const float invK = 1.f / (float)C;
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pin(xf[ks]);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t += (float)xf[ks][e];
            s += t;
            asm volatile("" : "+v"(s));
        }
        s += __shfl_xor(s, 32);
        const float mean = s * invK;
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pin(xf[ks]);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)xf[ks][e] - mean; t += d * d; }
            q += t;
            asm volatile("" : "+v"(q));
        }
        q += __shfl_xor(q, 32);
        const float rstd = rsqrtf(q * invK + a.ln_eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pin(xf[ks]);
            const h16x8 g = *reinterpret_cast<const h16x8*>(a.ln_g + ks * 16 + hh * 8);
            const h16x8 b = *reinterpret_cast<const h16x8*>(a.ln_b + ks * 16 + hh * 8);
            h16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)xf[ks][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            xf[ks] = o;
            pin(xf[ks]);
        }
    }

    // swizzled A-fragment addresses of a weight stage (linear_xs.hip)
    const int tsw = hh ^ ((l31 >> 1) & 7);
    int aoff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) aoff[m] = l31 * (C * 2) + ((tsw ^ (2 * m)) << 4);

    const float qscale = a.scale * 1.4426950408889634f;
    h16* patch = reinterpret_cast<h16*>(patch_base + wave * PATCH);

    f32x16 yacc[10];                             // out^T: ten 32-channel blocks x 32 pixels
#pragma unroll
    for (int ob = 0; ob < 10; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.f;

    for (int h = 0; h < HEADS; ++h) {
        f32x16 qacc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) qacc[b][r] = 0.f;
        h16x8 of[4];                             // O_h^T / l as B fragments (filled by the attention between stage kinds 1 and 2)
        static_for<0, 4>([&](auto Kc) {
            constexpr int kind = decltype(Kc)::value;
            const int s = 4 * h + kind;
            if (s == 0) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): panel, LayerNorm vectors, first K / V^T tile, the prologue's stages
            else wait_vm_n(vm_issued - mk[0]);                  // stage s has landed; the NST - 2 stages behind it may stay in flight
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int i = 0; i + 1 < NST - 1; ++i) mk[i] = mk[i + 1];
            // the K / V^T buffers are free once every wave is past the attention of head h (it sits in the kind-1 step, i.e. before this
            // barrier when kind == 2); the next head's tiles are issued AHEAD of stage s + NST - 1 = (h + 1, kind 1), whose wait -- at the
            // step that runs the attention of head h + 1 -- therefore covers them (VMEM retires in order), and that step's barrier
            // publishes every wave's part
            if (kind == 2 && h + 1 < HEADS) issue_kv(h + 1);
            if (s + NST - 1 < NSTAGE) { issue_stage(s + NST - 1, wr_slot); mk[NST - 2] = vm_issued; }    // into the slot stage s - 1 just left
            wr_slot = (wr_slot + 1 == NST) ? 0 : wr_slot + 1;
            const char* sW = ring + rd_slot * SLOT;
            rd_slot = (rd_slot + 1 == NST) ? 0 : rd_slot + 1;
            if constexpr (kind < 2) {
                // ---- Q_h^T block `kind` = Wq rows x X^T
#pragma unroll
                for (int k16 = 0; k16 < KS; ++k16) {
                    const h16x8 af = *reinterpret_cast<const h16x8*>(sW + aoff[k16 & 3] + (((2 * k16) & ~7) << 4));
                    qacc[kind] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, xf[k16], qacc[kind], 0, 0, 0);
                }
            } else {
                // ---- out^T blocks 5 (kind - 2) .. + 5  +=  Wo slice rows x O_h^T (k = d, in accumulator-row order)
                const h16* wo = reinterpret_cast<const h16*>(sW);
#pragma unroll
                for (int jb = 0; jb < 5; ++jb) {
#pragma unroll
                    for (int ks2 = 0; ks2 < 4; ++ks2) {
                        const h16* rowp = wo + (32 * jb + l31) * OLD + 32 * (ks2 >> 1) + 16 * (ks2 & 1) + 4 * hh;
                        const h16x4 lo = *reinterpret_cast<const h16x4*>(rowp);
                        const h16x4 hi = *reinterpret_cast<const h16x4*>(rowp + 8);
                        const h16x8 af = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        yacc[5 * (kind - 2) + jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, of[ks2], yacc[5 * (kind - 2) + jb], 0, 0, 0);
                    }
                }
            }
            if constexpr (kind == 1) {
                // ================= attention of head h (its K / V^T tiles were complete at this step's barrier) =================
                const h16* kt = reinterpret_cast<const h16*>(kbuf);
                const h16* vt = reinterpret_cast<const h16*>(vbuf);
                // ---- Q fragments: the accumulator blocks, rounded as the stand-alone path rounds them (fp16 Q, then fp16(Q * scale * log2 e))
                h16x8 qf[4];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const h16 q16 = (h16)qacc[b][4 * (2 * gp + (i >> 2)) + (i & 3)];
                            qf[b * 2 + gp][i] = (h16)((float)q16 * qscale);
                        }
                // ---- S^T = K_h Q_h: 3 key blocks x 4 k steps
                f32x16 sc[3];
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const h16* rowp = kt + (32 * kb + l31) * KLD + 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hh;
                        const h16x4 lo = *reinterpret_cast<const h16x4*>(rowp);
                        const h16x4 hi = *reinterpret_cast<const h16x4*>(rowp + 8);
                        const h16x8 af = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[ks], sc[kb], 0, 0, 0);
                    }
                }
                // ---- softmax over the keys of this lane's query column (rows 8g + 4hh + e of each block here, the rest in lane ^ 32)
                float mx = -3.0e38f;
#pragma unroll
                for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * kb + 8 * (r >> 2) + 4 * hh + (r & 3);
                        const float v = key < a.nk ? sc[kb][r] : -3.0e38f;
                        sc[kb][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float l = 0.f;
#pragma unroll
                for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(sc[kb][r] - mx);
                        sc[kb][r] = p;
                        l += p;
                    }
                l += __shfl_xor(l, 32);
                h16x8 pf[6];
#pragma unroll
                for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
                        for (int i = 0; i < 8; ++i) pf[kb * 2 + gp][i] = (h16)sc[kb][4 * (2 * gp + (i >> 2)) + (i & 3)];
                // ---- O_h^T = V_h^T P: 2 d blocks x 6 k steps
                f32x16 oa[2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) oa[db][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) {
                        const h16* rowp = vt + (32 * db + l31) * VLD + 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hh;
                        const h16x4 lo = *reinterpret_cast<const h16x4*>(rowp);
                        const h16x4 hi = *reinterpret_cast<const h16x4*>(rowp + 8);
                        const h16x8 af = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        oa[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, pf[ks], oa[db], 0, 0, 0);
                    }
                }
                // ---- O_h^T / l, fp16: B fragments of the output projection
                const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
                        for (int i = 0; i < 8; ++i) of[db * 2 + gp][i] = (h16)(oa[db][4 * (2 * gp + (i >> 2)) + (i & 3)] * inv);
            }
        });
    }

    // ---- epilogue (linear_xs MODE 1 rounding): fp16(acc + bias) + residual -> fp16, transposed through the wave's patch, 64-byte row segments
    {
        const int rb_row2 = lane >> 2, rb_chunk2 = lane & 3;
        static_for<0, 10>([&](auto Oc) {
            constexpr int ob = decltype(Oc)::value;
            h16x8 rr[2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
                rr[r] = *reinterpret_cast<const h16x8*>(a.res + (size_t)(p0 + rb_row2 + 16 * r) * C + ob * 32 + rb_chunk2 * 8);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const h16x4 b4 = *reinterpret_cast<const h16x4*>(a.bo + ob * 32 + 8 * g + 4 * hh);
                h16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (h16)(yacc[ob][4 * g + e] + (float)b4[e]);
                *reinterpret_cast<h16x4*>(patch + l31 * PLD + 8 * g + 4 * hh) = o;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = rb_row2 + 16 * r;
                const h16x8 v = *reinterpret_cast<const h16x8*>(patch + row * PLD + rb_chunk2 * 8);
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (h16)((float)v[e] + (float)rr[r][e]);
                *reinterpret_cast<h16x8*>(a.out + (size_t)(p0 + row) * C + ob * 32 + rb_chunk2 * 8) = o;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        });
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main() {
    const int n = 16, T = 3072, P = n * T, NK = 77;
    const float scale = 0.125f, eps = 1e-5f;
    unsigned seed = 12345u;
    std::vector<h16> x((size_t)P * C), g(C), b(C), Wq((size_t)C * C), Kp((size_t)n * HEADS * NKP * KLD, (h16)0.f), Vt((size_t)n * HEADS * D * VLD, (h16)0.f);
    std::vector<h16> Wo((size_t)C * C), Wop((size_t)HEADS * C * OLD, (h16)0.f), bo(C);
    std::vector<float> Kf((size_t)n * NK * C), Vf((size_t)n * NK * C);
    for (auto& v : x) v = (h16)(frand(seed) * 2.0f);
    for (int c = 0; c < C; ++c) { g[c] = (h16)(1.0f + 0.2f * frand(seed)); b[c] = (h16)(0.2f * frand(seed)); bo[c] = (h16)(0.3f * frand(seed)); }
    for (auto& v : Wq) v = (h16)(frand(seed) * 0.06f);
    for (auto& v : Wo) v = (h16)(frand(seed) * 0.08f);
    for (int h = 0; h < HEADS; ++h)
        for (int q = 0; q < C; ++q)
            for (int d = 0; d < D; ++d) Wop[((size_t)h * C + q) * OLD + d] = Wo[(size_t)q * C + h * D + d];
    for (size_t i = 0; i < Kf.size(); ++i) { Kf[i] = (float)(h16)(frand(seed)); Vf[i] = (float)(h16)(frand(seed)); }
    for (int s = 0; s < n; ++s)
        for (int h = 0; h < HEADS; ++h)
            for (int k = 0; k < NK; ++k)
                for (int d = 0; d < D; ++d) {
                    Kp[(((size_t)s * HEADS + h) * NKP + k) * KLD + d] = (h16)Kf[((size_t)s * NK + k) * C + h * D + d];
                    Vt[(((size_t)s * HEADS + h) * D + d) * VLD + k] = (h16)Vf[((size_t)s * NK + k) * C + h * D + d];
                }
    h16 *dx, *dg, *db, *dW, *dK, *dV, *dWo, *dbo, *dout;
    HIP_CHECK(hipMalloc(&dx, x.size() * 2)); HIP_CHECK(hipMalloc(&dg, C * 2)); HIP_CHECK(hipMalloc(&db, C * 2));
    HIP_CHECK(hipMalloc(&dW, Wq.size() * 2)); HIP_CHECK(hipMalloc(&dK, Kp.size() * 2)); HIP_CHECK(hipMalloc(&dV, Vt.size() * 2));
    HIP_CHECK(hipMalloc(&dWo, Wop.size() * 2)); HIP_CHECK(hipMalloc(&dbo, C * 2)); HIP_CHECK(hipMalloc(&dout, x.size() * 2));
    HIP_CHECK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dg, g.data(), C * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(db, b.data(), C * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dW, Wq.data(), Wq.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dK, Kp.data(), Kp.size() * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dV, Vt.data(), Vt.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dWo, Wop.data(), Wop.size() * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dbo, bo.data(), C * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dout, 0, x.size() * 2));
    Args a{dx, dg, db, eps, dW, dK, dV, dWo, dbo, dx, dout, P, T, NK, scale};
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_full_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    hipLaunchKernelGGL(xattn_full_kernel, dim3(P / 128), dim3(256), SMEM, 0, a);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<h16> out(x.size());
    HIP_CHECK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    // ---- fp32 reference on four workgroup tiles (first tile of samples 0 and 7, a middle tile, the last tile)
    const int tiles[4] = {0, 7 * (T / 128), 9 * (T / 128) + 11, P / 128 - 1};
    double max_err = 0.0, sum_sq = 0.0, ref_sq = 0.0;
    std::vector<float> xn(C), q(C), o(C), y(C);
    for (int t = 0; t < 4; ++t)
        for (int pp = 0; pp < 128; ++pp) {
            const int p = tiles[t] * 128 + pp, s = p / T;
            float mean = 0.f, var = 0.f;
            for (int c = 0; c < C; ++c) mean += (float)x[(size_t)p * C + c];
            mean /= C;
            for (int c = 0; c < C; ++c) { const float d = (float)x[(size_t)p * C + c] - mean; var += d * d; }
            const float rstd = 1.0f / std::sqrt(var / C + eps);
            for (int c = 0; c < C; ++c) xn[c] = (float)(h16)(((float)x[(size_t)p * C + c] - mean) * rstd * (float)g[c] + (float)b[c]);
            for (int j = 0; j < C; ++j) {
                float acc = 0.f;
                for (int c = 0; c < C; ++c) acc += (float)Wq[(size_t)j * C + c] * xn[c];
                q[j] = (float)(h16)acc;
            }
            for (int h = 0; h < HEADS; ++h) {
                float sc[NK], mx = -1e30f, l = 0.f;
                for (int k = 0; k < NK; ++k) {
                    float acc = 0.f;
                    for (int d = 0; d < D; ++d) acc += q[h * D + d] * Kf[((size_t)s * NK + k) * C + h * D + d];
                    sc[k] = acc * scale; mx = std::fmax(mx, sc[k]);
                }
                for (int k = 0; k < NK; ++k) { sc[k] = std::exp(sc[k] - mx); l += sc[k]; }
                for (int d = 0; d < D; ++d) {
                    float acc = 0.f;
                    for (int k = 0; k < NK; ++k) acc += sc[k] * Vf[((size_t)s * NK + k) * C + h * D + d];
                    o[h * D + d] = (float)(h16)(acc / l);
                }
            }
            for (int j = 0; j < C; ++j) {
                float acc = 0.f;
                for (int c = 0; c < C; ++c) acc += (float)Wo[(size_t)j * C + c] * o[c];
                y[j] = (float)(h16)(acc + (float)bo[j]) + (float)x[(size_t)p * C + j];
            }
            for (int c = 0; c < C; ++c) {
                const double e = (double)(float)out[(size_t)p * C + c] - (double)y[c];
                max_err = std::fmax(max_err, std::fabs(e)); sum_sq += e * e; ref_sq += (double)y[c] * y[c];
            }
        }
    printf("rel-L2 %.3e  max |err| %.3e  (fp16 storage: expect ~5e-4 / ~4e-3)\n", std::sqrt(sum_sq / ref_sq), max_err);
    // ---- timing
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(xattn_full_kernel, dim3(P / 128), dim3(256), SMEM, 0, a);
    HIP_CHECK(hipEventRecord(e0, 0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(xattn_full_kernel, dim3(P / 128), dim3(256), SMEM, 0, a);
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 4.0 * P * C * C + 4.0 * P * NK * C;
    printf("%.1f us per launch (%.0f TFLOP/s algorithmic; the three launches it replaces: ~35 + 28 + 27 us of to_q + attention + to_out)\n",
           1000.0 * ms / iters, flop / (ms / iters * 1e-3) / 1e12);
    std::vector<h16> out2(x.size());
    HIP_CHECK(hipMemcpy(out2.data(), dout, out2.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < out.size(); ++i) diff += (float)out[i] != (float)out2[i];
    printf("repeat launches bit-equal: %s\n", diff == 0 ? "yes" : "NO");
    return 0;
}
