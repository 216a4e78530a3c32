// Runtime core: error state, weight store, device pools, arena, weight repacking, op wrappers.
#include "runtime.h"
#include <stdexcept>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <cstdlib>

namespace ladi {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

const HostTensor& WeightStore::get(const std::string& k) const {
    auto it = m.find(k);
    if (it == m.end()) throw std::runtime_error("missing weight: " + k);
    return it->second;
}

// ------------------------------------------------------------------------------------------------
void* DevPool::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > left) {
        size_t chunk = std::max(bytes, (size_t)64 << 20);
        void* p = nullptr;
        HIP_OK(hipMalloc(&p, chunk));
        chunks.push_back(p);
        cur = reinterpret_cast<char*>(p);
        left = chunk;
        total += chunk;
    }
    void* r = cur;
    cur += bytes;
    left -= bytes;
    return r;
}
static inline uint16_t f2h_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
h16* DevPool::upload_h16(const std::vector<float>& v) {
    std::vector<uint16_t> tmp(v.size());
    for (size_t i = 0; i < v.size(); ++i) tmp[i] = f2h_bits(v[i]);
    void* d = alloc(tmp.size() * 2);
    HIP_OK(hipMemcpy(d, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    return reinterpret_cast<h16*>(d);
}
float* DevPool::upload_f32(const std::vector<float>& v) {
    void* d = alloc(v.size() * 4);
    HIP_OK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return reinterpret_cast<float*>(d);
}
DevPool::~DevPool() { for (void* p : chunks) (void)hipFree(p); }

void* Arena::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    size_t o = off;
    off += bytes;
    if (off > peak) peak = off;
    if (dry) return reinterpret_cast<void*>((uintptr_t)0x10000 + o);
    if (off > cap) throw std::runtime_error("activation arena overflow (planned " + std::to_string(cap) + " B)");
    return base + o;
}
void Arena::reserve(size_t bytes) {
    if (bytes <= cap) return;
    if (base) (void)hipFree(base);
    base = nullptr; cap = 0;
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, bytes));
    base = reinterpret_cast<char*>(p);
    cap = bytes;
}
Arena::~Arena() { if (base) (void)hipFree(base); }

float* Ctx::alloc_stats(size_t floats) {
    size_t o = stats_off;
    stats_off += floats;
    if (stats_off > stats_peak) stats_peak = stats_off;
    if (dry()) return reinterpret_cast<float*>((uintptr_t)0x10000 + o * 4);
    if (stats_off > stats_cap) throw std::runtime_error("GroupNorm stats arena overflow");
    return stats + o;
}
Act Ctx::new_act(int n, int h, int w, int cc, int ld) {
    Act a; a.n = n; a.h = h; a.w = w; a.c = cc; a.ld = ld ? ld : cc;
    a.p = alloc_h16(a.pixels() * (size_t)a.ld);
    return a;
}
void Ctx::check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + " failed rc=" + std::to_string(rc));
}

// ------------------------------------------------------------------------------------------------
// weight repacking
// ------------------------------------------------------------------------------------------------
static int pad64(int c) { return (c + 63) / 64 * 64; }

// [cout][cin][kh][kw] (or [cout][cin]) fp32 -> [cout][kh*kw][cin_pad] fp32
static void repack_conv(const HostTensor& w, int& cout, int& cin, int& k, int cin_pad, std::vector<float>& out) {
    if (w.shape.size() == 4) { cout = (int)w.shape[0]; cin = (int)w.shape[1]; k = (int)w.shape[2]; }
    else if (w.shape.size() == 2) { cout = (int)w.shape[0]; cin = (int)w.shape[1]; k = 1; }
    else throw std::runtime_error("unsupported weight rank");
    if (cin_pad < cin) throw std::runtime_error("cin_pad < cin");
    const int taps = k * k;
    out.assign((size_t)cout * taps * cin_pad, 0.f);
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i)
            for (int t = 0; t < taps; ++t)
                out[((size_t)o * taps + t) * cin_pad + i] = w.data[((size_t)o * cin + i) * taps + t];
}

DConv load_conv(DevPool& pool, const WeightStore& ws, const std::string& prefix, int cin_expected) {
    const HostTensor& w = ws.get(prefix + ".weight");
    DConv d;
    int cin_raw = (int)w.shape[1];
    if (cin_expected >= 0 && cin_raw != cin_expected) throw std::runtime_error(prefix + ": unexpected in_channels");
    std::vector<float> r;
    repack_conv(w, d.cout, d.cin, d.k, pad64(cin_raw), r);
    d.cin_pad = pad64(cin_raw);
    d.w = pool.upload_h16(r);
    if (ws.has(prefix + ".bias")) d.b = pool.upload_h16(ws.get(prefix + ".bias").data);
    return d;
}

DConv load_linear_cat(DevPool& pool, const WeightStore& ws, const std::vector<std::string>& prefixes, bool bias) {
    DConv d; d.k = 1;
    std::vector<float> wcat, bcat;
    for (auto& p : prefixes) {
        const HostTensor& w = ws.get(p + ".weight");
        if (w.shape.size() != 2) throw std::runtime_error(p + ": linear weight must be 2-D");
        int cin = (int)w.shape[1], cp = pad64(cin);
        if (d.cin && d.cin != cin) throw std::runtime_error("linear cat: in_features mismatch");
        d.cin = cin; d.cin_pad = cp;
        const int rows = (int)w.shape[0];
        size_t base = wcat.size();
        wcat.resize(base + (size_t)rows * cp, 0.f);
        for (int r = 0; r < rows; ++r) std::memcpy(&wcat[base + (size_t)r * cp], &w.data[(size_t)r * cin], (size_t)cin * 4);
        d.cout += rows;
        if (bias) { const HostTensor& b = ws.get(p + ".bias"); bcat.insert(bcat.end(), b.data.begin(), b.data.end()); }
    }
    d.w = pool.upload_h16(wcat);
    if (bias) d.b = pool.upload_h16(bcat);
    return d;
}

// GEGLU projection [8C][C]: rows interleaved in blocks of 32: [u block | g block] (igemm LADI_ACT_GEGLU)
DConv load_geglu(DevPool& pool, const WeightStore& ws, const std::string& prefix) {
    const HostTensor& w = ws.get(prefix + ".weight");
    const HostTensor& b = ws.get(prefix + ".bias");
    const int rows = (int)w.shape[0], cin = (int)w.shape[1], half = rows / 2;
    if (half % 32) throw std::runtime_error("GEGLU inner dim must be a multiple of 32");
    const int cp = pad64(cin);
    std::vector<float> wr((size_t)rows * cp, 0.f), br(rows);
    for (int j = 0; j < half; ++j) {
        const int blk = j / 32, i = j % 32;
        const int ru = blk * 64 + i, rg = blk * 64 + 32 + i;
        std::memcpy(&wr[(size_t)ru * cp], &w.data[(size_t)j * cin], (size_t)cin * 4);
        std::memcpy(&wr[(size_t)rg * cp], &w.data[(size_t)(half + j) * cin], (size_t)cin * 4);
        br[ru] = b.data[j];
        br[rg] = b.data[half + j];
    }
    DConv d; d.k = 1; d.cin = cin; d.cin_pad = cp; d.cout = rows;
    d.w = pool.upload_h16(wr);
    d.b = pool.upload_h16(br);
    return d;
}

DNorm load_norm(DevPool& pool, const WeightStore& ws, const std::string& prefix) {
    DNorm n;
    const HostTensor& g = ws.get(prefix + ".weight");
    n.c = (int)g.numel();
    n.g = pool.upload_h16(g.data);
    n.b = pool.upload_h16(ws.get(prefix + ".bias").data);
    return n;
}

// ------------------------------------------------------------------------------------------------
// op wrappers
// ------------------------------------------------------------------------------------------------
static float* alloc_part(Ctx& c, const Act& a) {   // worst case: rows of 32 pixels
    return c.alloc_f32(((a.pixels() + 31) / 32) * (size_t)a.c * 2);
}

Act new_act_with_stats(Ctx& c, int n, int h, int w, int cc) {
    Act a = c.new_act(n, h, w, cc);
    a.st_part = alloc_part(c, a);
    return a;
}

void launch_conv_into(Ctx& c, IGemmArgs& a, Act& out, int cfg) {
    a.out = out.p; a.ldo = out.ld;
    a.stats = (out.st_part && out.ld == out.c) ? out.st_part : nullptr;
    int px = 0;
    // split-K slab from this handle's planned arena (stack discipline: it lives until the enclosing block releases its mark), so a
    // captured graph only ever references memory covered by the graph key (arena base)
    const size_t wsb = ladi_igemm_splitk_ws_bytes(a, 1);
    float* ws = wsb ? c.alloc_f32(wsb / sizeof(float)) : nullptr;
    if (!c.dry()) c.check(ladi_launch_igemm(a, 1, cfg, c.st, &px, ws, wsb, c.sk_cnt), "igemm");
    out.st_px = px;
    if (!c.dry() && px == 0) out.st_part = nullptr;
}

Act conv2d(Ctx& c, const DConv& cv, const Act& x, const Act* x2, const ConvOpt& o) {
    const int pad = o.pad >= 0 ? o.pad : cv.k / 2;
    const int Hlog = o.ups ? 2 * x.h : x.h, Wlog = o.ups ? 2 * x.w : x.w;
    int Ho, Wo;
    if (o.stride == 1) { Ho = Hlog; Wo = Wlog; }
    else { Ho = Hlog / 2; Wo = Wlog / 2; }
    const int C0 = x.c, C1 = x2 ? x2->c : 0;
    if (C0 + C1 != cv.cin_pad) throw std::runtime_error("conv2d: channel mismatch (" + std::to_string(C0 + C1) + " vs " + std::to_string(cv.cin_pad) + ")");
    const bool geglu = o.act == LADI_ACT_GEGLU;
    const int cout = geglu ? cv.cout / 2 : cv.cout;
    Act out = c.new_act(x.n, Ho, Wo, cout, o.out_ld);
    if (o.stats && !geglu) out.st_part = alloc_part(c, out);
    IGemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.src0 = x.p; a.C0 = C0; a.ld0 = x.ld;
    if (x2) { a.src1 = x2->p; a.C1 = C1; a.ld1 = x2->ld; }
    a.Hs = x.h; a.Ws = x.w; a.Ho = Ho; a.Wo = Wo; a.P = x.n * Ho * Wo;
    a.ksize = cv.k; a.stride = o.stride; a.pad = pad; a.ups = o.ups;
    a.W = cv.w; a.Q = cv.cout; a.K = cv.K(); a.ldw = 0;
    a.bias = cv.b; a.rowadd = o.rowadd; a.rowadd_idx = o.rowadd_idx; a.rowadd_stride = o.rowadd_stride;
    a.act = o.act; a.out_scale = o.out_scale; a.bias_mul = o.bias_mul;
    if (o.res0) { a.res0 = o.res0->p; a.ldr0 = o.res0->ld; }
    if (o.res1) { a.res1 = o.res1->p; a.ldr1 = o.res1->ld; }
    a.mask = o.mask;
    a.out_f32 = 0;
    if (o.gn_ss) {
        if (x2 || o.ln) throw std::runtime_error("conv2d: GroupNorm affine needs a single, un-normalised source");
        a.gn_ss = o.gn_ss; a.gn_hw = x.h * x.w;
    }
    if (o.ln) {
        if (x2 || o.ln->c != C0) throw std::runtime_error("conv2d: LayerNorm input must be a single source of matching width");
        a.ln_gamma = o.ln->g; a.ln_beta = o.ln->b; a.ln_eps = o.ln_eps;
        // fused into the X-stationary kernel or run as its own kernel into this scratch: the tuner decides per shape, so the
        // scratch is reserved unconditionally (the planning pass and the real pass allocate identically)
        a.ln_scratch = c.new_act(x.n, x.h, x.w, C0).p;
    }
    launch_conv_into(c, a, out, o.cfg);
    return out;
}

// GroupNorm over the virtual concat (x | x2): partial statistics come from the producers' epilogues when available
namespace {
struct GnParts { const float* part[2] = {nullptr, nullptr}; int rps[2] = {0, 0}; };
GnParts gn_parts(Ctx& c, const Act& x, const Act* x2, bool direct_ok) {
    const int HW = x.h * x.w;
    const Act* srcs[2] = {&x, x2};
    GnParts g;
    for (int i = 0; i < 2; ++i) {
        const Act* s = srcs[i];
        if (!s) continue;
        // the fallback buffer is reserved unconditionally so that the planning pass and the real pass allocate identically
        const int rows = ladi_gn_partial_rows(s->n, HW, s->c);
        float* p = c.alloc_f32((size_t)s->n * rows * s->c * 2);
        if (s->st_part && s->st_px > 0) {
            g.part[i] = s->st_part;
            g.rps[i] = HW / s->st_px;
        } else if (direct_ok && ladi_gn_norm_direct(HW)) {
            g.part[i] = nullptr; g.rps[i] = 0;      // tiny samples: the one-pass kernel takes the statistics from the data (no gn_partial launch)
        } else {
            if (!c.dry()) c.check(ladi_launch_gn_partial(s->p, s->c, s->ld, s->n, HW, p, c.st), "gn_partial");
            g.part[i] = p; g.rps[i] = rows;
        }
    }
    return g;
}
}  // namespace

float* gn_scale_shift(Ctx& c, const DNorm& nm, const Act& x, const Act* x2, int groups, float eps) {
    const int C0 = x.c, C1 = x2 ? x2->c : 0;
    if (C0 + C1 != nm.c) throw std::runtime_error("group_norm: channel mismatch");
    const int HW = x.h * x.w;
    float* ss = c.alloc_f32((size_t)x.n * (C0 + C1) * 2);
    const GnParts g = gn_parts(c, x, x2, false);
    if (!c.dry())
        c.check(ladi_launch_gn_finalize(g.part[0], C0, g.rps[0], g.part[1], C1, g.rps[1], x.n, HW, groups, nm.g, nm.b, eps, ss, c.st, c.bad), "gn_finalize");
    return ss;
}

Act group_norm(Ctx& c, const DNorm& nm, const Act& x, const Act* x2, int groups, float eps, int silu, const Act* add) {
    const int C0 = x.c, C1 = x2 ? x2->c : 0;
    if (C0 + C1 != nm.c) throw std::runtime_error("group_norm: channel mismatch");
    const int HW = x.h * x.w;
    Act out = c.new_act(x.n, x.h, x.w, C0 + C1);
    // (the scale / shift table is reserved whichever form runs: the arena plan does not depend on LADI_GN_ONEPASS)
    float* ss = c.alloc_f32((size_t)x.n * (C0 + C1) * 2);
    // (direct statistics only when the one-pass form will take the launch: group size and the other source's rows decide that)
    const int gs = (C0 + C1) / groups;
    const bool onepass_shape = ladi_gn_norm_eligible(C0, 1, C1, C1 ? 1 : 0, groups, HW) && gs * groups == C0 + C1;
    GnParts g = gn_parts(c, x, x2, onepass_shape);
    // VAE-sized tensors (thousands of partial rows per sample): fold the rows first (one small coalesced kernel per source), then the one-pass
    // kernel -- instead of gn_finalize's strided walk over 50 MB of rows (52.6 us x 62 per step in round 5).  The scratch is reserved whichever
    // form runs (planning pass == real pass).
    {
        const Act* srcs[2] = {&x, x2};
        for (int i = 0; i < 2; ++i) {
            if (!srcs[i]) continue;
            float* folded = c.alloc_f32((size_t)x.n * ladi_gn_reduce_rows() * srcs[i]->c * 2);
            if (onepass_shape && g.part[i] && ladi_gn_reduce_eligible(srcs[i]->c, g.rps[i])) {
                if (!c.dry()) c.check(ladi_launch_gn_reduce(g.part[i], srcs[i]->c, g.rps[i], x.n, folded, c.st), "gn_reduce");
                g.part[i] = folded; g.rps[i] = ladi_gn_reduce_rows();
            }
        }
    }
    if (c.dry()) return out;
    if (ladi_gn_norm_eligible(C0, g.rps[0], C1, g.rps[1], groups, HW)) {
        // few partial rows per sample (every UNet level): each block finalises its own 64-channel chunk -- one launch instead of two
        c.check(ladi_launch_gn_norm(x.p, C0, x.ld, g.part[0], g.rps[0], x2 ? x2->p : nullptr, C1, x2 ? x2->ld : 0, g.part[1], g.rps[1], x.n, HW, groups,
                                    nm.g, nm.b, eps, silu, add ? add->p : nullptr, out.p, c.st, c.bad), "gn_norm");
        return out;
    }
    c.check(ladi_launch_gn_finalize(g.part[0], C0, g.rps[0], g.part[1], C1, g.rps[1], x.n, HW, groups, nm.g, nm.b, eps, ss, c.st, c.bad), "gn_finalize");
    c.check(ladi_launch_gn_apply(x.p, C0, x.ld, x2 ? x2->p : nullptr, C1, x2 ? x2->ld : 0, x.n, HW, ss, silu, add ? add->p : nullptr, out.p,
                                 c.st), "gn_apply");
    return out;
}

bool gn_fusable(const Act& x, int cout) {
    static const bool on = [] { const char* e = getenv("LADI_GN_FUSE"); return !(e && e[0] == '0'); }();
    const int HW = x.h * x.w;
    return on && (x.c == 320 || x.c == 640) && x.ld == x.c && (HW % 128) == 0 && (cout % 32) == 0;
}

Act layer_norm(Ctx& c, const DNorm& nm, const Act& x, float eps) {
    Act out = c.new_act(x.n, x.h, x.w, x.c);
    if (c.dry()) return out;
    c.check(ladi_launch_layernorm(x.p, x.ld, nm.g, nm.b, eps, (int)x.pixels(), x.c, out.p, out.ld, c.st), "layernorm");
    return out;
}

// ------------------------------------------------------------------------------------------------
// scheduler tables
// ------------------------------------------------------------------------------------------------
void default_alphas_cumprod(std::vector<float>& ac) {
    // scaled_linear: betas = linspace(sqrt(0.00085), sqrt(0.012), 1000)^2 ; fp32 arithmetic like torch
    const int N = 1000;
    ac.resize(N);
    const float s = std::sqrt(0.00085f), e = std::sqrt(0.012f);
    const float step = (e - s) / (float)(N - 1);
    float prod = 1.f;
    for (int i = 0; i < N; ++i) {
        float b = (i < N / 2) ? (s + step * (float)i) : (e - step * (float)(N - 1 - i));
        b = b * b;
        prod *= (1.f - b);
        ac[i] = prod;
    }
}

// integral over [lo, hi] of the Lagrange basis polynomial of node s[j] among s[0..order) (degree <= 3): 4-point Gauss-Legendre, exact
static double lms_basis_integral(const double* s, int order, int j, double lo, double hi) {
    static const double gx[4] = {-0.8611363115940526, -0.3399810435848563, 0.3399810435848563, 0.8611363115940526};
    static const double gw[4] = {0.3478548451374538, 0.6521451548625461, 0.6521451548625461, 0.3478548451374538};
    const double c = 0.5 * (hi + lo), h = 0.5 * (hi - lo);
    double acc = 0.0;
    for (int q = 0; q < 4; ++q) {
        const double tau = c + h * gx[q];
        double prod = 1.0;
        for (int k = 0; k < order; ++k) if (k != j) prod *= (tau - s[k]) / (s[j] - s[k]);
        acc += gw[q] * prod;
    }
    return acc * h;
}

void build_step_table(int kind, int steps, const float* ac, int cloth_zero_from, std::vector<double>& timesteps,
                      std::vector<StepTable>& table, SchedInfo* info) {
    const int T = 1000;
    if (steps < 2 || steps > T) throw std::runtime_error("num_inference_steps out of range [2, 1000]");   // ts[steps - 2] below
    if (kind < 0 || kind > 2) throw std::runtime_error("scheduler kind must be 0 (DDIM), 1 (PNDM) or 2 (LMSDiscrete)");
    const int ratio = T / steps;
    const double final_ac = ac[0];  // set_alpha_to_one = False
    timesteps.clear(); table.clear();
    if (info) *info = SchedInfo();
    if (kind == 2) {
        // diffusers 0.14 LMSDiscreteScheduler.set_timesteps / step (order 4, epsilon prediction): timesteps = linspace(0, T-1, n)[::-1];
        // sigma = interp(t, arange(T), sqrt((1 - a) / a)) held in fp32, trailing 0; derivative d_i = (x - (x - sigma_i eps)) / sigma_i = eps
        std::vector<float> sig((size_t)steps + 1, 0.f);
        for (int i = 0; i < steps; ++i) {
            // numpy.linspace: arange(n) * ((stop - start) / (n - 1)) + start, last element set to stop exactly
            const int k = steps - 1 - i;
            const double t = k == steps - 1 ? (double)(T - 1) : (double)k * ((double)(T - 1) / (double)(steps - 1));
            timesteps.push_back(t);
            const int lo = std::min((int)t, T - 1), hi = std::min(lo + 1, T - 1);
            const double slo = (double)(float)std::sqrt((1.0 - (double)ac[lo]) / (double)ac[lo]);
            const double shi = (double)(float)std::sqrt((1.0 - (double)ac[hi]) / (double)ac[hi]);
            sig[i] = (float)(slo + (shi - slo) * (t - (double)lo));
        }
        float smax = 0.f;
        for (float s : sig) smax = std::max(smax, s);
        std::vector<float> coeffs((size_t)steps * 4, 0.f);
        for (int i = 0; i < steps; ++i) {
            StepTable e; std::memset(&e, 0, sizeof(e));
            const int order = std::min(i + 1, 4);
            double nodes[4];
            for (int k = 0; k < order; ++k) nodes[k] = (double)sig[i - k];
            for (int j = 0; j < order; ++j) {
                e.w[j] = (float)lms_basis_integral(nodes, order, j, (double)sig[i], (double)sig[i + 1]);
                coeffs[(size_t)i * 4 + j] = e.w[j];
            }
            e.c_x = 1.f; e.c_e = 1.f;
            const int slot = i & 3, s1 = (i - 1) & 3, s2 = (i - 2) & 3, s3 = (i - 3) & 3;
            e.push = 1 | (slot << 4) | (s1 << 8) | (s2 << 10) | (s3 << 12);
            e.in_scale_next = i + 1 < steps ? (float)(1.0 / std::sqrt((double)sig[i + 1] * (double)sig[i + 1] + 1.0)) : 1.f;
            table.push_back(e);
        }
        if (info) {
            info->init_noise_sigma = smax;
            info->in_scale0 = (float)(1.0 / std::sqrt((double)sig[0] * (double)sig[0] + 1.0));
            info->sigmas = sig; info->lms_coeffs = coeffs;
        }
    } else if (kind == 0) {
        for (int i = steps - 1; i >= 0; --i) timesteps.push_back(i * ratio + 1);
        for (int i = 0; i < steps; ++i) {
            const int t = (int)timesteps[i], tp = t - ratio;
            const double a_t = ac[t], a_p = tp >= 0 ? (double)ac[tp] : final_ac;
            StepTable e; std::memset(&e, 0, sizeof(e));
            e.c_x = (float)std::sqrt(a_p / a_t);
            e.c_e = (float)(std::sqrt(1.0 - a_p) - std::sqrt(a_p) * std::sqrt(1.0 - a_t) / std::sqrt(a_t));
            e.w[0] = 1.f;
            table.push_back(e);
        }
    } else {
        std::vector<int> ts;
        for (int i = 0; i < steps; ++i) ts.push_back(i * ratio + 1);
        // concat(_ts[:-1], _ts[-2:-1], _ts[-1:])[::-1]
        std::vector<int> seq(ts.begin(), ts.end() - 1);
        seq.push_back(ts[steps - 2]);
        seq.push_back(ts[steps - 1]);
        for (int i = (int)seq.size() - 1; i >= 0; --i) timesteps.push_back(seq[i]);
        int npush = 0;  // number of pushes so far
        for (int i = 0; i < (int)timesteps.size(); ++i) {
            int t = (int)timesteps[i], tp = t - ratio;
            StepTable e; std::memset(&e, 0, sizeof(e));
            const int counter = i;
            int hist = npush;  // entries available before this evaluation
            if (counter != 1) {
                // push eps_now; ets[-1] = now, ets[-2] = previous pushes...
                const int slot = npush & 3;
                const int s1 = (npush - 1) & 3, s2 = (npush - 2) & 3, s3 = (npush - 3) & 3;
                e.push = 1 | (slot << 4) | (s1 << 8) | (s2 << 10) | (s3 << 12);
                const int len = std::min(hist + 1, 4);
                if (len == 1) { e.w[0] = 1.f; e.save_cur = (counter == 0); }
                else if (len == 2) { e.w[0] = 1.5f; e.w[1] = -0.5f; }
                else if (len == 3) { e.w[0] = 23.f / 12.f; e.w[1] = -16.f / 12.f; e.w[2] = 5.f / 12.f; }
                else { e.w[0] = 55.f / 24.f; e.w[1] = -59.f / 24.f; e.w[2] = 37.f / 24.f; e.w[3] = -9.f / 24.f; }
                ++npush;
            } else {
                tp = t; t = t + ratio;
                const int s1 = (npush - 1) & 3;
                e.push = (s1 << 8);
                e.mode = 1; e.w[0] = 0.5f; e.w[1] = 0.5f;
            }
            const double a_t = ac[t], a_p = tp >= 0 ? (double)ac[tp] : final_ac;
            const double b_t = 1.0 - a_t, b_p = 1.0 - a_p;
            e.c_x = (float)std::sqrt(a_p / a_t);
            const double denom = a_t * std::sqrt(b_p) + std::sqrt(a_t * b_t * a_p);
            e.c_e = (float)(-(a_p - a_t) / denom);
            table.push_back(e);
        }
    }
    if (kind != 2) for (auto& e : table) e.in_scale_next = 1.f;
    // `if i >= num_inference_steps - cloth_conditioning_steps: cloth = 0` (tryon_pipe.py:718-719), evaluated at the
    // START of evaluation i -> mark entry i-1 so that the step kernel zeroes the cloth channels for evaluation i.
    for (int i = 1; i < (int)table.size(); ++i)
        if (i >= cloth_zero_from) table[i - 1].zero_cloth_next = 1;
}

}  // namespace ladi
