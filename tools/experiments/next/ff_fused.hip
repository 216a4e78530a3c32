// Stand-alone prototype (round-5 candidate): the feed-forward of a transformer block of the 64x48 level -- LayerNorm -> GEGLU up-projection
// (320 -> 2 x 1280) -> down-projection (1280 -> 320) + bias + residual -- as ONE kernel.  Today: linear_xs GEGLU (~109 us) writes the
// [49152 x 1280] hidden tensor (126 MB), the down-projection (~57 us) reads it back; five blocks per forward.  Here the hidden tensor is
// never written: a wave keeps its 32-pixel panel in registers (linear_xs.hip), and for each block of 32 hidden channels
//     u = W1u X^T, g = W1g X^T  (2 x 20 MFMAs)  ->  h = fp16((u + bu) * gelu(g + bg))  ->  out^T[320 x 32] += W2[:, 32 hb : 32 hb + 32] h
// where h, straight from the accumulator block, is the B operand of the last product (the accumulator -> B-operand chain of
// xattn_q.hip / xattn_full.hip: the k order of an accumulator block is a row permutation that the A side absorbs by reading two 8-byte
// pieces per fragment).  The ten 32-channel output accumulator blocks (160 registers) live next to the 80-register panel: one wave per
// SIMD.  W1 / b1 in the library's GEGLU packing (32-row blocks alternate u | g, runtime_core.cpp load_geglu); W2 packed per hidden block
// as [40][320][36] halves (row stride 36: conflict-free 8-byte fragment reads, plain linear LDS-DMA).  Three ring stages per hidden block.
//
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/next/ff_fused.hip -o /tmp/ff_fused && /tmp/ff_fused
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

constexpr int C = 320, HID = 1280, NB = HID / 32;
constexpr int KS = C / 16;                       // k16 steps of the up-projection
constexpr int STAGE = 32 * C * 2;                // one up-projection stage: 32 rows x 640 B
constexpr int W_DMA = STAGE / 16 / 256;          // 16-byte pieces per thread per stage (5)
constexpr int DLD = 36;                          // row stride of a packed W2 slice (halves)
constexpr int DSTAGE = C * DLD * 2;              // 320 output channels x 72 B = 23 040 B
constexpr int D_DMA = 6;                         // ... lands in six 4 KB DMA rounds (the descriptor ends with the slice: the rest is zero)
constexpr int SLOT = D_DMA * 4096;
constexpr int PLD = 40;                          // halves per patch row (32 channels + 8 pad)
constexpr int PATCH = 32 * PLD * 2;
constexpr int BIAS1 = 2 * HID * 2;               // the up-projection's bias vector (5 KB)
constexpr int NST = 6;                           // ring depth: at ONE workgroup per CU nobody else covers a stage's round trip -- a stage is
                                                 // consumed in ~0.3 us (20 MFMAs), so five stages (~110 KB) have to be in flight
constexpr int NSTAGE = 3 * NB;
constexpr int SMEM = NST * SLOT + BIAS1 + 4 * PATCH;

struct Args {
    const h16* x; const h16* ln_g; const h16* ln_b; float ln_eps;
    const h16* W1; const h16* b1;                // GEGLU packing: [2 HID][C], [2 HID] (32-row blocks alternate u | g)
    const h16* W2; const h16* bo;                // [NB][C][36] (W2[q][32 hb + j] at [hb][q][j]) , bias [C]
    const h16* res;                              // residual [P][C] (the block input)
    h16* out;                                    // [P][C]
    int P;
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

// erf by Abramowitz & Stegun 7.1.26 and the exact GELU built on it: the library's (common.h)
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(fmaf(-p, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }

// PIPE = 1: the GEGLU arithmetic of hidden block hb - 1 (320 VALU instructions per wave) is placed next to the u / g MFMAs of block hb (a
// second pair of accumulator blocks), and the down-projection of a block runs one block late: at one wave per SIMD no other wave hides the
// VALU work, the wave has to overlap it with its own MFMAs.  Ring order: u0 g0 | u1 g1 W2(0) | u2 g2 W2(1) | ... | W2(NB-1).
template <int PIPE>
__global__ __launch_bounds__(256, 1) void ff_fused_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    h16* bias_s = reinterpret_cast<h16*>(smem + NST * SLOT);     // [2 HID]
    char* patch_base = smem + NST * SLOT + BIAS1;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int p0 = blockIdx.x * 128 + wave * 32;

    // ---- up-projection DMA (linear_xs.hip): LDS position i = j*256 + tid (16-byte units) -> row i/40, physical chunk i%40, XOR swizzle on the source
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.W1), 0, (unsigned)(2 * HID * C * 2), 0x00020000);
    unsigned wrel[W_DMA];
#pragma unroll
    for (int j = 0; j < W_DMA; ++j) {
        const int i = j * 256 + tid;
        const int row = i / (C / 8), cph = i - row * (C / 8);
        const int c = cph ^ ((row >> 1) & 7);
        wrel[j] = (unsigned)((row * C + c * 8) * 2);
    }
    int vm_issued = 0;                           // running count of this wave's VMEM operations (wave-uniform)
    // stage (hb, kind): kind 0 / 1 = the u / g rows of hidden block hb (32-row block 2 hb + kind of the packed W1; swizzled, 5 pieces per
    // thread); kind 2 = the packed W2 slice of hidden block hb (linear, 6 rounds)
    auto issue_stage = [&](int t, int slot) {    // PIPE 0: stage t = (hidden block t / 3, kind t % 3); PIPE 1: the order above
        int hb, kind;
        if (!PIPE) { hb = t / 3; kind = t - 3 * hb; }
        else if (t < 2) { hb = 0; kind = t; }
        else {
            const int grp = (t - 2) / 3, k = (t - 2) - 3 * grp;
            if (grp == NB - 1) { hb = NB - 1; kind = 2; }
            else if (k < 2) { hb = grp + 1; kind = k; }
            else { hb = grp; kind = 2; }
        }
        char* dst = ring + slot * SLOT + wave * 1024;
        if (kind < 2) {
#pragma unroll
            for (int j = 0; j < W_DMA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(dst + j * 4096), 16, wrel[j] + (unsigned)((2 * hb + kind) * STAGE), 0, 0, 0);
            vm_issued += W_DMA;
        } else {
            const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.W2) + (size_t)hb * C * DLD, 0, (unsigned)DSTAGE, 0x00020000);
#pragma unroll
            for (int j = 0; j < D_DMA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (lds_ptr_t)(dst + j * 4096), 16, (unsigned)((j * 256 + tid) * 16), 0, 0, 0);
            vm_issued += D_DMA;
        }
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

    // ---- the up-projection's bias -> LDS (its loads are drained by the vmcnt(0) in front of stage 0)
    for (int i = tid; i < 2 * HID; i += 256) bias_s[i] = a.b1[i];

    // swizzled A-fragment addresses of an up-projection stage (linear_xs.hip)
    const int tsw = hh ^ ((l31 >> 1) & 7);
    int aoff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) aoff[m] = l31 * (C * 2) + ((tsw ^ (2 * m)) << 4);

    h16* patch = reinterpret_cast<h16*>(patch_base + wave * PATCH);

    f32x16 yacc[10];                             // out^T: ten 32-channel blocks x 32 pixels
#pragma unroll
    for (int ob = 0; ob < 10; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.f;

    if constexpr (!PIPE) {
        for (int hb = 0; hb < NB; ++hb) {
            f32x16 ug[2];
    #pragma unroll
            for (int b = 0; b < 2; ++b)
    #pragma unroll
                for (int r = 0; r < 16; ++r) ug[b][r] = 0.f;
            h16x8 hf[2];                             // the hidden block as B fragments (filled after the g stage)
            static_for<0, 3>([&](auto Kc) {
                constexpr int kind = decltype(Kc)::value;
                const int s = 3 * hb + kind;
                if (s == 0) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): panel, LayerNorm vectors, bias, the prologue's stages
                else wait_vm_n(vm_issued - mk[0]);                  // stage s has landed; the NST - 2 stages behind it may stay in flight
                asm volatile("s_barrier" ::: "memory");
    #pragma unroll
                for (int i = 0; i + 1 < NST - 1; ++i) mk[i] = mk[i + 1];
                if (s + NST - 1 < NSTAGE) { issue_stage(s + NST - 1, wr_slot); mk[NST - 2] = vm_issued; }    // into the slot stage s - 1 just left
                wr_slot = (wr_slot + 1 == NST) ? 0 : wr_slot + 1;
                const char* sW = ring + rd_slot * SLOT;
                rd_slot = (rd_slot + 1 == NST) ? 0 : rd_slot + 1;
                if constexpr (kind < 2) {
    #pragma unroll
                    for (int k16 = 0; k16 < KS; ++k16) {
                        const h16x8 af = *reinterpret_cast<const h16x8*>(sW + aoff[k16 & 3] + (((2 * k16) & ~7) << 4));
                        ug[kind] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, xf[k16], ug[kind], 0, 0, 0);
                    }
                } else {
                    // ---- out^T += W2 slice x h (k = hidden channel inside the block, in accumulator-row order)
                    const h16* wd = reinterpret_cast<const h16*>(sW);
    #pragma unroll
                    for (int ob = 0; ob < 10; ++ob) {
    #pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const h16* rowp = wd + (32 * ob + l31) * DLD + 16 * gp + 4 * hh;
                            const h16x4 lo = *reinterpret_cast<const h16x4*>(rowp);
                            const h16x4 hi = *reinterpret_cast<const h16x4*>(rowp + 8);
                            const h16x8 af = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            yacc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, hf[gp], yacc[ob], 0, 0, 0);
                        }
                    }
                }
                if constexpr (kind == 1) {
                    // ---- h = fp16((u + bu) * gelu(g + bg)): the GEGLU epilogue of linear_xs MODE 2, kept in registers as B fragments
                    const h16* bu = bias_s + (2 * hb) * 32;
                    const h16* bg = bu + 32;
    #pragma unroll
                    for (int gp = 0; gp < 2; ++gp)
    #pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int r0 = 4 * (2 * gp + q);                  // first accumulator register of this group of four
                            const int row0 = 8 * (2 * gp + q) + 4 * hh;       // its row inside the 32-channel block
                            const h16x4 bu4 = *reinterpret_cast<const h16x4*>(bu + row0);
                            const h16x4 bg4 = *reinterpret_cast<const h16x4*>(bg + row0);
    #pragma unroll
                            for (int e = 0; e < 4; ++e)
                                hf[gp][4 * q + e] = (h16)((ug[0][r0 + e] + (float)bu4[e]) * gelu_f(ug[1][r0 + e] + (float)bg4[e]));
                        }
                }
            });
        }
    } else {
        static_assert(NB % 2 == 0, "the block loop is unrolled by two (two pairs of accumulator blocks with compile-time names)");
        int step = 0;
        auto begin = [&]() -> const char* {       // hand-over of the next ring stage (same protocol as the plain loop)
            if (step == 0) __builtin_amdgcn_s_waitcnt(0x0F70);
            else wait_vm_n(vm_issued - mk[0]);
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int i = 0; i + 1 < NST - 1; ++i) mk[i] = mk[i + 1];
            if (step + NST - 1 < NSTAGE) { issue_stage(step + NST - 1, wr_slot); mk[NST - 2] = vm_issued; }
            wr_slot = (wr_slot + 1 == NST) ? 0 : wr_slot + 1;
            const char* sW = ring + rd_slot * SLOT;
            rd_slot = (rd_slot + 1 == NST) ? 0 : rd_slot + 1;
            ++step;
            return sW;
        };
        auto up = [&](const char* sW, f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int k16 = 0; k16 < KS; ++k16) {
                const h16x8 af = *reinterpret_cast<const h16x8*>(sW + aoff[k16 & 3] + (((2 * k16) & ~7) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, xf[k16], acc, 0, 0, 0);
            }
        };
        auto geglu_half = [&](const f32x16& u, const f32x16& g, int hb, int gp, h16x8& dst) {
            const h16* bu = bias_s + (2 * hb) * 32;
            const h16* bg = bu + 32;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r0 = 4 * (2 * gp + q), row0 = 8 * (2 * gp + q) + 4 * hh;
                const h16x4 bu4 = *reinterpret_cast<const h16x4*>(bu + row0);
                const h16x4 bg4 = *reinterpret_cast<const h16x4*>(bg + row0);
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[4 * q + e] = (h16)((u[r0 + e] + (float)bu4[e]) * gelu_f(g[r0 + e] + (float)bg4[e]));
            }
        };
        auto down = [&](const char* sW, const h16x8 (&hf)[2]) {
            const h16* wd = reinterpret_cast<const h16*>(sW);
#pragma unroll
            for (int ob = 0; ob < 10; ++ob)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const h16* rowp = wd + (32 * ob + l31) * DLD + 16 * gp + 4 * hh;
                    const h16x4 lo = *reinterpret_cast<const h16x4*>(rowp);
                    const h16x4 hi = *reinterpret_cast<const h16x4*>(rowp + 8);
                    const h16x8 af = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    yacc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, hf[gp], yacc[ob], 0, 0, 0);
                }
        };
        f32x16 uA, gA, uB, gB;
        h16x8 hf[2];
        up(begin(), uA);
        up(begin(), gA);
        for (int hb = 1; hb < NB; hb += 2) {
            // block hb into (uB, gB) next to the GEGLU of block hb - 1 from (uA, gA)
            { const char* sW = begin(); up(sW, uB); geglu_half(uA, gA, hb - 1, 0, hf[0]); }
            { const char* sW = begin(); up(sW, gB); geglu_half(uA, gA, hb - 1, 1, hf[1]); }
            down(begin(), hf);
            if (hb + 1 < NB) {
                { const char* sW = begin(); up(sW, uA); geglu_half(uB, gB, hb, 0, hf[0]); }
                { const char* sW = begin(); up(sW, gA); geglu_half(uB, gB, hb, 1, hf[1]); }
                down(begin(), hf);
            }
        }
        geglu_half(uB, gB, NB - 1, 0, hf[0]);
        geglu_half(uB, gB, NB - 1, 1, hf[1]);
        down(begin(), hf);
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
    const int P = 16 * 3072;
    const float eps = 1e-5f;
    unsigned seed = 4242u;
    std::vector<h16> x((size_t)P * C), g(C), b(C), W1((size_t)2 * HID * C), b1(2 * HID), W1p((size_t)2 * HID * C), b1p(2 * HID);
    std::vector<h16> W2((size_t)C * HID), W2p((size_t)NB * C * DLD, (h16)0.f), bo(C);
    for (auto& v : x) v = (h16)(frand(seed) * 2.0f);
    for (int c = 0; c < C; ++c) { g[c] = (h16)(1.0f + 0.2f * frand(seed)); b[c] = (h16)(0.2f * frand(seed)); bo[c] = (h16)(0.3f * frand(seed)); }
    for (auto& v : W1) v = (h16)(frand(seed) * 0.06f);          // [2 HID][C]: rows 0..HID-1 = value half, HID.. = gate half (diffusers GEGLU.proj)
    for (auto& v : b1) v = (h16)(frand(seed) * 0.2f);
    for (auto& v : W2) v = (h16)(frand(seed) * 0.03f);          // [C][HID]
    for (int j = 0; j < HID; ++j) {                             // the library's packing (runtime_core.cpp load_geglu)
        const int blk = j / 32, i = j % 32, ru = blk * 64 + i, rg = blk * 64 + 32 + i;
        for (int c = 0; c < C; ++c) { W1p[(size_t)ru * C + c] = W1[(size_t)j * C + c]; W1p[(size_t)rg * C + c] = W1[(size_t)(HID + j) * C + c]; }
        b1p[ru] = b1[j]; b1p[rg] = b1[HID + j];
    }
    for (int hb = 0; hb < NB; ++hb)
        for (int q = 0; q < C; ++q)
            for (int j = 0; j < 32; ++j) W2p[((size_t)hb * C + q) * DLD + j] = W2[(size_t)q * HID + hb * 32 + j];
    h16 *dx, *dg, *db, *dW1, *db1, *dW2, *dbo, *dout;
    HIP_CHECK(hipMalloc(&dx, x.size() * 2)); HIP_CHECK(hipMalloc(&dg, C * 2)); HIP_CHECK(hipMalloc(&db, C * 2));
    HIP_CHECK(hipMalloc(&dW1, W1p.size() * 2)); HIP_CHECK(hipMalloc(&db1, b1p.size() * 2)); HIP_CHECK(hipMalloc(&dW2, W2p.size() * 2));
    HIP_CHECK(hipMalloc(&dbo, C * 2)); HIP_CHECK(hipMalloc(&dout, x.size() * 2));
    HIP_CHECK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dg, g.data(), C * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(db, b.data(), C * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dW1, W1p.data(), W1p.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(db1, b1p.data(), b1p.size() * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dW2, W2p.data(), W2p.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dbo, bo.data(), C * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dout, 0, x.size() * 2));
    Args a{dx, dg, db, eps, dW1, db1, dW2, dbo, dx, dout, P};
  for (int variant = 0; variant < 2; ++variant) {
    auto kfn = variant ? ff_fused_kernel<1> : ff_fused_kernel<0>;
    printf("---- %s\n", variant ? "PIPE = 1 (GEGLU of block hb - 1 next to the MFMAs of block hb)" : "PIPE = 0 (plain)");
    HIP_CHECK(hipMemset(dout, 0, x.size() * 2));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    hipLaunchKernelGGL(kfn, dim3(P / 128), dim3(256), SMEM, 0, a);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<h16> out(x.size());
    HIP_CHECK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    // ---- fp32 reference on two workgroup tiles (the first and the last)
    const int tiles[2] = {0, P / 128 - 1};
    double max_err = 0.0, sum_sq = 0.0, ref_sq = 0.0;
    std::vector<float> xn(C), hbuf(HID), y(C);
    for (int t = 0; t < 2; ++t)
        for (int pp = 0; pp < 128; ++pp) {
            const int p = tiles[t] * 128 + pp;
            float mean = 0.f, var = 0.f;
            for (int c = 0; c < C; ++c) mean += (float)x[(size_t)p * C + c];
            mean /= C;
            for (int c = 0; c < C; ++c) { const float d = (float)x[(size_t)p * C + c] - mean; var += d * d; }
            const float rstd = 1.0f / std::sqrt(var / C + eps);
            for (int c = 0; c < C; ++c) xn[c] = (float)(h16)(((float)x[(size_t)p * C + c] - mean) * rstd * (float)g[c] + (float)b[c]);
            for (int j = 0; j < HID; ++j) {
                float u = (float)b1[j], gg = (float)b1[HID + j];
                for (int c = 0; c < C; ++c) { u += (float)W1[(size_t)j * C + c] * xn[c]; gg += (float)W1[(size_t)(HID + j) * C + c] * xn[c]; }
                hbuf[j] = (float)(h16)(u * 0.5f * gg * (1.0f + std::erf(gg * 0.70710678f)));
            }
            for (int q = 0; q < C; ++q) {
                float acc = 0.f;
                for (int j = 0; j < HID; ++j) acc += (float)W2[(size_t)q * HID + j] * hbuf[j];
                y[q] = (float)(h16)(acc + (float)bo[q]) + (float)x[(size_t)p * C + q];
            }
            for (int c = 0; c < C; ++c) {
                const double e = (double)(float)out[(size_t)p * C + c] - (double)y[c];
                max_err = std::fmax(max_err, std::fabs(e)); sum_sq += e * e; ref_sq += (double)y[c] * y[c];
            }
        }
    printf("rel-L2 %.3e  max |err| %.3e  (fp16 storage: expect ~5e-4 / ~4e-3)\n", std::sqrt(sum_sq / ref_sq), max_err);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(P / 128), dim3(256), SMEM, 0, a);
    HIP_CHECK(hipEventRecord(e0, 0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kfn, dim3(P / 128), dim3(256), SMEM, 0, a);
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * P * C * (2.0 * HID) + 2.0 * P * HID * C;
    printf("%.1f us per launch (%.0f TFLOP/s algorithmic; the two launches it replaces: ~109 + 57 us)\n", 1000.0 * ms / iters, flop / (ms / iters * 1e-3) / 1e12);
    std::vector<h16> out2(x.size());
    HIP_CHECK(hipMemcpy(out2.data(), dout, out2.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < out.size(); ++i) diff += (float)out[i] != (float)out2[i];
    printf("repeat launches bit-equal: %s\n", diff == 0 ? "yes" : "NO");
  }
    return 0;
}
