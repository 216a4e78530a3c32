// Implicit-GEMM convolution / linear / batched GEMM on gfx950 MFMA (v_mfma_f32_32x32x16_f16).
//
// One kernel family covers every dense contraction of the LaDI-VTON hot path:
//   conv3x3 s1/s2 (with optional folded nearest-2x upsample and two-source channel concat),
//   conv1x1, nn.Linear, and the batched products of the VAE mid-block attention.
//
// Orientation: D[q][p] = sum_k W[q][k] * X[p][k]   (q = output channel, p = output pixel / token)
//   A operand (MFMA rows)  = weight tile  [BQ][64]  staged in LDS
//   B operand (MFMA cols)  = gathered activation tile [BP][64] staged in LDS (im2col done by the gather)
//   -> every lane ends up owning ONE pixel (col = lane&31) and groups of 4 CONSECUTIVE output channels
//      (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so NHWC epilogue loads/stores are 8-byte vectors and
//      per-pixel quantities (mask) are lane-local.
//
// Tiles: block = 256 threads = 4 waves (WQ x WP), each wave owns (TQ*32) x (TP*32) outputs.
// K loop: BK = 64 halves (128-byte rows) per step, register-staged global->LDS double buffering
// (one barrier per step).  LDS rows are XOR-swizzled in 16-byte chunks with ((row>>1)&7) so that both the
// 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups are bank-conflict free
// (MI355X_MICROARCH.md §LDS: ds_read_b128 bank = (addr/4)%64, 4x16-lane groups).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BK = 64;

__device__ __forceinline__ int swz(int row, int chunk) { return (row * BK) + (((chunk ^ ((row >> 1) & 7))) << 3); }

template <int WQ, int WP, int TQ, int TP>
__global__ __launch_bounds__(256) void igemm_kernel(const IGemmArgs a) {
    constexpr int BQ = WQ * TQ * 32, BP = WP * TP * 32;
    constexpr int RQ = BQ / 32, RP = BP / 32;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h16* smem = reinterpret_cast<h16*>(smem_raw);
    // layout: buf0: [W tile BQ*64][X tile BP*64]  buf1: same
    constexpr int BUF = (BQ + BP) * BK;

    const int tid = threadIdx.x;
    const int nq = (a.Q + BQ - 1) / BQ;
    const int tile = blockIdx.x;
    const int q0 = (tile % nq) * BQ;
    const int p0 = (tile / nq) * BP;
    const int z = blockIdx.z;

    const h16* __restrict__ src0 = a.src0 + (size_t)z * a.bs_src0;
    const h16* __restrict__ src1 = a.src1;
    const h16* __restrict__ Wp = a.W + (size_t)z * a.bs_w;

    const int c8 = tid & 7;
    const int r0 = tid >> 3;  // 0..31
    const int swz_chunk = ((c8 ^ ((r0 >> 1) & 7)) << 3);  // (r0 + 32*i)>>1 & 7 == (r0>>1)&7

    // ---- per-thread pixel-row decode (constant over the K loop)
    const int HoWo = a.Ho * a.Wo;
    const int Hlog = a.ups ? 2 * a.Hs : a.Hs;
    const int Wlog = a.ups ? 2 * a.Ws : a.Ws;
    int nb[RP], iy0[RP], ix0[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        int p = p0 + r0 + 32 * i;
        bool ok = p < a.P;
        int pp = ok ? p : 0;
        int n = pp / HoWo;
        int rem = pp - n * HoWo;
        int oy = rem / a.Wo;
        int ox = rem - oy * a.Wo;
        iy0[i] = ok ? (oy * a.stride - a.pad) : -100000;  // invalid rows fail the bounds test
        ix0[i] = ox * a.stride - a.pad;
        nb[i] = n * a.Hs * a.Ws;
    }

    const int Ct = a.C0 + a.C1;
    const int nk = a.K / BK;
    const int ldw = a.ldw ? a.ldw : a.K;

    uint4 xr[RP], wr[RQ];
    int tap = 0, cb = 0;  // running (tap, channel base) of the NEXT k-step to load

    auto gload = [&]() {
        int dy = 0, dx = 0;
        if (a.ksize == 3) { dy = tap / 3; dx = tap - dy * 3; }
        const h16* sp; int ld, c;
        if (cb < a.C0) { sp = src0; ld = a.ld0; c = cb; } else { sp = src1; ld = a.ld1; c = cb - a.C0; }
        const int k0 = tap * Ct + cb;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            int iy = iy0[i] + dy, ix = ix0[i] + dx;
            bool ok = ((unsigned)iy < (unsigned)Hlog) && ((unsigned)ix < (unsigned)Wlog);
            if (a.ups) { iy >>= 1; ix >>= 1; }
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                size_t off = (size_t)(nb[i] + iy * a.Ws + ix) * (size_t)ld + (size_t)(c + c8 * 8);
                v = *reinterpret_cast<const uint4*>(sp + off);
            }
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RQ; ++i) {
            int q = q0 + r0 + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < a.Q) v = *reinterpret_cast<const uint4*>(Wp + (size_t)q * ldw + k0 + c8 * 8);
            wr[i] = v;
        }
        cb += BK;
        if (cb >= Ct) { cb = 0; ++tap; }
    };
    auto lstore = [&](int buf) {
        h16* sW = smem + buf * BUF;
        h16* sX = sW + BQ * BK;
#pragma unroll
        for (int i = 0; i < RQ; ++i) *reinterpret_cast<uint4*>(sW + (r0 + 32 * i) * BK + swz_chunk) = wr[i];
#pragma unroll
        for (int i = 0; i < RP; ++i) *reinterpret_cast<uint4*>(sX + (r0 + 32 * i) * BK + swz_chunk) = xr[i];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wq = wave / WP, wp = wave % WP;
    const int l31 = lane & 31, hh = lane >> 5;

    f32x16 acc[TQ][TP];
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload();
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload();
        const h16* sW = smem + buf * BUF;
        const h16* sX = sW + BQ * BK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int chunk = kk * 2 + hh;
            h16x8 af[TQ], bf[TP];
#pragma unroll
            for (int i = 0; i < TQ; ++i) {
                int r = (wq * TQ + i) * 32 + l31;
                af[i] = *reinterpret_cast<const h16x8*>(sW + swz(r, chunk));
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                int r = (wp * TP + j) * 32 + l31;
                bf[j] = *reinterpret_cast<const h16x8*>(sX + swz(r, chunk));
            }
#pragma unroll
            for (int i = 0; i < TQ; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------------------------------
    // Epilogue: lane owns pixel (col) l31 of each pixel sub-tile and 4-channel groups of each q sub-tile.
    // ------------------------------------------------------------------------------------------
    const bool geglu = (a.act == LADI_ACT_GEGLU);
    const int Qout = geglu ? a.Q / 2 : a.Q;
    const float* rowadd = a.rowadd;
    if (rowadd && a.rowadd_idx) rowadd += (size_t)(*a.rowadd_idx) * a.rowadd_stride;
    const size_t zo = (size_t)z * a.bs_out;
    const size_t zr = (size_t)z * a.bs_res;
    const bool vec_ok = ((a.ldo & 3) == 0);
    const bool rvec0 = a.res0 && ((a.ldr0 & 3) == 0);
    const bool rvec1 = a.res1 && ((a.ldr1 & 3) == 0);

#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int p = p0 + (wp * TP + j) * 32 + l31;
        if (p >= a.P) continue;
        float mk = 1.f;
        if (a.mask) mk = 1.f - (float)a.mask[p];
        float pbias = 0.f;
        if (a.bias && a.bias_per_pixel) pbias = (float)a.bias[p];
#pragma unroll
        for (int i = 0; i < (geglu ? 1 : TQ); ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qw = q0 + (wq * TQ + i) * 32 + 8 * g + 4 * hh;  // W-row index of reg 4g
                int co;                                                   // output channel of reg 4g
                if (geglu) co = (q0 + wq * TQ * 32) / 2 + 8 * g + 4 * hh; else co = qw;
                if (co >= Qout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][4 * g + e];
                    if (a.bias_per_pixel) x += pbias;
                    else if (a.bias && (qw + e) < a.Q) x += (float)a.bias[qw + e];
                    if (geglu) {
                        float gg = acc[TQ > 1 ? 1 : 0][j][4 * g + e];
                        if (a.bias && (qw + 32 + e) < a.Q) gg += (float)a.bias[qw + 32 + e];
                        x = x * gelu_f(gg);
                    } else {
                        if (rowadd && (co + e) < Qout) x += rowadd[co + e];
                        if (a.act == LADI_ACT_SILU) x = silu_f(x);
                        else if (a.act == LADI_ACT_GELU) x = gelu_f(x);
                    }
                    v[e] = x * a.out_scale;
                }
                const bool full = (co + 3 < Qout);
                if (a.res0) {
                    const h16* rp = a.res0 + zr + (size_t)p * a.ldr0 + co;
                    if (full && rvec0) { h16x4 r = *reinterpret_cast<const h16x4*>(rp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)r[e]; }
                    else { for (int e = 0; e < 4; ++e) if (co + e < Qout) v[e] += (float)rp[e]; }
                }
                if (a.res1) {
                    const h16* rp = a.res1 + zr + (size_t)p * a.ldr1 + co;
                    if (full && rvec1) { h16x4 r = *reinterpret_cast<const h16x4*>(rp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)r[e]; }
                    else { for (int e = 0; e < 4; ++e) if (co + e < Qout) v[e] += (float)rp[e]; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= mk;
                if (a.out_f32) {
                    float* op = reinterpret_cast<float*>(a.out) + zo + (size_t)p * a.ldo + co;
                    if (full && vec_ok) { *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]); }
                    else { for (int e = 0; e < 4; ++e) if (co + e < Qout) op[e] = v[e]; }
                } else {
                    h16* op = reinterpret_cast<h16*>(a.out) + zo + (size_t)p * a.ldo + co;
                    if (full && vec_ok) {
                        h16x4 o; o[0] = (h16)v[0]; o[1] = (h16)v[1]; o[2] = (h16)v[2]; o[3] = (h16)v[3];
                        *reinterpret_cast<h16x4*>(op) = o;
                    } else { for (int e = 0; e < 4; ++e) if (co + e < Qout) op[e] = (h16)v[e]; }
                }
            }
        }
    }
}

template <int WQ, int WP, int TQ, int TP>
int launch_cfg(const IGemmArgs& a, int batch, hipStream_t st) {
    constexpr int BQ = WQ * TQ * 32, BP = WP * TP * 32;
    constexpr int SMEM = 2 * (BQ + BP) * BK * (int)sizeof(h16);
    static bool attr_set = false;
    auto kfn = igemm_kernel<WQ, WP, TQ, TP>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
            return -10;
        attr_set = true;
    }
    const int nq = (a.Q + BQ - 1) / BQ, np = (a.P + BP - 1) / BP;
    dim3 grid((unsigned)(nq * np), 1, (unsigned)batch);
    hipLaunchKernelGGL(kfn, grid, dim3(256), SMEM, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -11;
}

}  // namespace

#include <vector>
namespace {
struct ProfRec { hipEvent_t e0, e1; int cfg; double flops; };
bool g_prof = false;
std::vector<ProfRec> g_recs;
}  // namespace

// Tile-shape choice. cfg: 0 = auto, 1 = Q128xP128, 2 = Q64xP256, 3 = Q64xP64, 4 = Q128xP64
int ladi_launch_igemm(const IGemmArgs& a, int batch, int cfg, hipStream_t st) {
    if (a.ksize != 1 && a.ksize != 3) return -1;
    if ((a.C0 % BK) || (a.C1 % BK)) return -2;
    if (a.K != a.ksize * a.ksize * (a.C0 + a.C1)) return -3;
    if ((a.ld0 % 8) || (a.C1 && (a.ld1 % 8)) || (a.K % 8) || (a.ldw % 8)) return -4;
    if (a.P <= 0 || a.Q <= 0) return -5;
    const bool geglu = a.act == LADI_ACT_GEGLU;
    if (geglu && (a.Q % 64)) return -6;
    if (cfg == 0) {
        auto tiles = [&](int bq, int bp) { return (long long)((a.Q + bq - 1) / bq) * ((a.P + bp - 1) / bp) * batch; };
        auto waste = [&](int bq, int bp) {
            double padded = (double)((a.Q + bq - 1) / bq * bq) * ((a.P + bp - 1) / bp * bp);
            return padded / ((double)a.Q * a.P);
        };
        // prefer big tiles when they fill the chip (>= 2 blocks on each of 256 CUs) without padding waste
        if (waste(128, 128) < 1.05 && tiles(128, 128) >= 384) cfg = 1;
        else if (waste(64, 256) < 1.05 && tiles(64, 256) >= 384) cfg = 2;
        else if (waste(128, 64) < 1.10 && tiles(128, 64) >= 256) cfg = 4;
        else if (geglu) cfg = (waste(128, 64) <= waste(64, 256)) ? 4 : 2;
        else cfg = 3;
    }
    if (geglu && cfg == 3) cfg = 4;
    ProfRec rec;
    const bool prof = g_prof;
    if (prof) {
        if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) return -12;
        rec.cfg = cfg;
        rec.flops = 2.0 * (double)a.P * (double)a.Q * (double)a.K * (double)batch;
        (void)hipEventRecord(rec.e0, st);
    }
    int rc;
    switch (cfg) {
        case 1: rc = launch_cfg<2, 2, 2, 2>(a, batch, st); break;
        case 2: rc = launch_cfg<1, 4, 2, 2>(a, batch, st); break;
        case 3: rc = launch_cfg<2, 2, 1, 1>(a, batch, st); break;
        case 4: rc = launch_cfg<2, 2, 2, 1>(a, batch, st); break;
        default: rc = -7;
    }
    if (prof) { (void)hipEventRecord(rec.e1, st); g_recs.push_back(rec); }
    return rc;
}

// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream around every igemm launch
void ladi_igemm_profile_enable(int on) { g_prof = on != 0; }
// out[cfg*3 + {0,1,2}] = {total ms, algorithmic FLOP (2*P*Q*K), launches} for cfg 1..4 (index 0 = all); clears the records
int ladi_igemm_profile_collect(double* out15) {
    for (int i = 0; i < 15; ++i) out15[i] = 0.0;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return -2;
        out15[r.cfg * 3 + 0] += ms; out15[r.cfg * 3 + 1] += r.flops; out15[r.cfg * 3 + 2] += 1.0;
        out15[0] += ms; out15[1] += r.flops; out15[2] += 1.0;
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_recs.clear();
    return 0;
}
