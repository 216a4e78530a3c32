// Host-side launch prototypes of every HIP kernel in libladi_native (internal header).
// All activations are NHWC fp16 unless noted; every launcher is asynchronous on `st`, allocates
// nothing and is therefore hipGraph-capturable. Return 0 on success, negative on argument errors.
#pragma once
#include "common.h"

// ---- igemm.hip
// a.stats (optional): per-channel partial statistics of the output, [ceil(P / px)][Q][2] floats (sum, sumsq) where px =
// *stats_row_px pixels per row (a multiple of 32 chosen with the tile shape; 0 = not produced, use ladi_launch_gn_partial)
// ws / ws_bytes: caller-owned split-K slab (fp32 partial tiles) of at least ladi_igemm_splitk_ws_bytes(a, batch) bytes; null -> a
// process-wide grow-only fallback buffer (never freed, so pointers baked into captured graphs stay valid)
// sk_cnt: caller-owned arrival counters of the in-launch split-K combine (>= 1024 ints, zero-initialised once; every launch leaves them
// zeroed) -- one buffer per stream that may run split-K launches concurrently; null -> a process-wide buffer (single-stream callers)
int ladi_launch_igemm(const IGemmArgs& a, int batch, int cfg, hipStream_t st, int* stats_row_px = nullptr, float* ws = nullptr,
                      size_t ws_bytes = 0, int* sk_cnt = nullptr);
// worst-case split-K slab for this problem over every admissible split configuration (0: split-K can never be chosen); depends on
// the problem shape only, so a planning pass and the real pass allocate identically
size_t ladi_igemm_splitk_ws_bytes(const IGemmArgs& a, int batch);

// ---- igemm8.hip: phase-staggered 8-wave large-tile kernel, wave tile (tq*32 channels) x (tp*32 pixels); a.splitk / a.tile_map as set by
// ladi_launch_igemm, which is the only caller
int ladi_launch_igemm8(const IGemmArgs& a, int tq, int tp, int batch, hipStream_t st);

// ---- igemm_lc.hip: loader / consumer kernel, consumer wave tile (tq*32 channels) x (tp*32 pixels) on a 2 x 2 consumer grid, nst-deep ring
int ladi_launch_igemm_lc(const IGemmArgs& a, int tq, int tp, int nst, int batch, hipStream_t st);

// ---- igemm_halo.hip: halo-resident 3x3 convolution, workgroup tile (64 tq) x (128 tp), nxb halo buffers
bool ladi_igemm_halo_eligible(const IGemmArgs& a, int batch);
bool ladi_igemm_halo2d_eligible(const IGemmArgs& a, int batch, int th);
bool ladi_igemm_halo_ups_eligible(const IGemmArgs& a, int batch, int bp);   // folded nearest-2x upsample + 3x3 (round 6)   // 2-D blocked form (nxb 20 / 21): W % 32 == 0, H % th == 0
int ladi_launch_igemm_halo(const IGemmArgs& a, int tq, int tp, int nxb, int batch, hipStream_t st);

// ---- linear_xs.hip: X-stationary kernel for 1x1 layers with K = 320 / 640 (reached through ladi_launch_igemm cfg 23..27)
// nst: weight-ring depth (3: two workgroups per CU; 2: three workgroups per CU, K = 320 plain / GEGLU projections only)
bool ladi_linear_xs_eligible(const IGemmArgs& a, int batch, int pb, int qs, int nst = 3);
int ladi_launch_linear_xs(const IGemmArgs& a, int pb, int qs, hipStream_t st, int nst = 3);
void ladi_linear_xs_symbol(const IGemmArgs& a, int pb, int nst, char* out, int n);

// ---- xf_fused.hip: fused transformer sub-blocks of the C = 320 level (5 heads of 64); see the file header for the operand packings
struct XAttnBlockArgs {
    const h16* x; const h16* ln_g; const h16* ln_b; float ln_eps;     // block input [P][320] (dense rows), LayerNorm (norm2)
    const h16* Wq;                               // attn2.to_q [320][320] row-major, no bias
    const h16* Kp; const h16* Vt;                // packed context tiles of the FIRST sample of x: [n][5][96][68], [n][5][64][100]
    const h16* Wo; const h16* bo;                // attn2.to_out.0 packed [5][320][68], bias [320]
    const h16* res;                              // residual [P][320] (normally == x)
    h16* out;                                    // [P][320]
    int P, T, nk; float scale;                   // pixels, pixels per sample (multiples of 128), context length (<= 96), 1 / sqrt(64)
};
struct FFBlockArgs {
    const h16* x; const h16* ln_g; const h16* ln_b; float ln_eps;     // block input [P][320], LayerNorm (norm3)
    const h16* W1; const h16* b1;                // ff.net.0.proj in the GEGLU packing [2560][320], [2560] (load_geglu)
    const h16* W2; const h16* bo;                // ff.net.2 packed [40][320][36], bias [320]
    const h16* res;                              // residual [P][320] (normally == x)
    h16* out;
    int P;
};
bool ladi_xf_fused_eligible(int C, int heads, int T, int L);
size_t ladi_xf_kp_elems(int n);
size_t ladi_xf_vt_elems(int n);
size_t ladi_xf_wo_packed_elems();
size_t ladi_xf_w2_packed_elems();
int ladi_launch_pack_kv_tiles(const h16* kv, int n, int L, int C, h16* kp, h16* vt, hipStream_t st);   // kv [n][L][2C]
int ladi_launch_pack_wo(const h16* Wo, h16* out, hipStream_t st);
int ladi_launch_pack_w2(const h16* W2, h16* out, hipStream_t st);
int ladi_launch_xattn_block(const XAttnBlockArgs& a, hipStream_t st);
int ladi_launch_ff_block(const FFBlockArgs& a, hipStream_t st);        // LADI_FF_PIPE=0: the plain loop (A/B)

// ---- norm.hip
// GroupNorm in three stages (all atomics-free): per-channel partial statistics rows [rows][C][2] (written by the producing
// igemm's epilogue, or by ladi_launch_gn_partial), finalize -> scale_shift[n][C0+C1][2], apply.
int ladi_gn_partial_rows(int n, int HW, int C);   // rows per sample ladi_launch_gn_partial writes
int ladi_launch_gn_partial(const h16* src, int C, int ld, int n, int HW, float* part, hipStream_t st);
int ladi_launch_gn_finalize(const float* part0, int C0, int rps0, const float* part1, int C1, int rps1, int n, int HW, int groups,
                            const h16* gamma, const h16* beta, float eps, float* scale_shift, hipStream_t st, int* bad = nullptr);
// `bad` (optional, device): set to 1 when a group's statistics are not finite -- an fp16 overflow upstream (VAE range guard)
// y = act(x * scale + shift) (+ add) over the virtual concat (src0[C0] | src1[C1]): out [n][HW][C0+C1] dense
int ladi_launch_gn_apply(const h16* src0, int C0, int ld0, const h16* src1, int C1, int ld1, int n, int HW,
                         const float* scale_shift, int silu, const h16* add, h16* out, hipStream_t st);
// one-pass form (round 5): every block finalises the groups of its own 64-channel chunk from the partial rows (few rows per sample only:
// ladi_gn_norm_eligible) and applies -- no gn_finalize launch, no scale / shift table
// (rps = 0 with a null part pointer: that source has no partial rows and the kernel sums the data itself -- samples of <= 64 pixels only,
// ladi_gn_norm_direct)
// round 6: fold the many partial rows of a VAE-sized tensor ([n][rps][C][2]) into ladi_gn_reduce_rows() rows per sample ([n][rows][C][2]),
// after which ladi_launch_gn_norm takes it (rps > the one-pass kernel's row limit only)
int ladi_gn_reduce_rows();
bool ladi_gn_reduce_eligible(int C, int rps);
int ladi_launch_gn_reduce(const float* part, int C, int rps, int n, float* out, hipStream_t st);
bool ladi_gn_norm_direct(int HW);
bool ladi_gn_norm_eligible(int C0, int rps0, int C1, int rps1, int groups, int HW);
int ladi_launch_gn_norm(const h16* src0, int C0, int ld0, const float* part0, int rps0, const h16* src1, int C1, int ld1, const float* part1,
                        int rps1, int n, int HW, int groups, const h16* gamma, const h16* beta, float eps, int silu, const h16* add, h16* out,
                        hipStream_t st, int* bad = nullptr);
int ladi_launch_layernorm(const h16* x, int ldx, const h16* gamma, const h16* beta, float eps, int rows, int C, h16* out,
                          int ldo, hipStream_t st);
// P[r][:] = softmax(scale * S[r][:]) ; S fp32 [rows][cols], P fp16
int ladi_launch_softmax_rows(const float* S, int rows, int cols, float scale, h16* P, hipStream_t st);

// ---- attention.hip
struct AttnArgs {
    const h16* q; const h16* k; const h16* v; h16* o;
    int ldq, ldk, ldv, ldo;                 // row strides (elements)
    long long sq, sk, sv, so;               // per-sample strides (elements)
    int n, heads, Nq, Nk;                   // head h lives at column offset h*64 of each row
    float scale;
    int causal = 0;                         // 1: query i sees keys <= i (Nq == Nk)
    int qtiles = 0, xcd_map = 0;            // set by the launcher: query tiles per (sample, head); XCD-aware workgroup order
};
int ladi_launch_flash_attn64(const AttnArgs& a, hipStream_t st);
// heads of dimension 64 / 80 / 96 / 128 at column offset h*head_dim (CLIP ViT-H vision tower: 80); non-causal
int ladi_launch_attn_generic(const AttnArgs& a, int head_dim, hipStream_t st);
// ONE wide head (head_dim 128 / 256 / 512: the VAE AttentionBlock), flash-style; a.v = V^T [head_dim][Nk] (row stride a.ldv),
// a.heads must be 1, Nk % 4 == 0
int ladi_launch_attn_wide(const AttnArgs& a, int head_dim, hipStream_t st);
// single query per (sample, head): q [n][ldq], k/v [n][Nk][ld], generic head dim d <= 128
int ladi_launch_attn_single_query(const h16* q, int ldq, const h16* k, int ldk, const h16* v, int ldv, h16* o, int ldo,
                                  int n, int heads, int d, int Nk, long long sk, long long sv, float scale, hipStream_t st);

// ---- elementwise.hip
// out[m][nn] = act(sum_k x[m][k] W[nn][k] + b[nn]) for small M (weight-streaming GEMV-like); x/out fp16 or fp32
// optional residual res[m][nn] (fp16, row stride ldr) is added after the activation
int ladi_launch_small_linear(const void* x, int x_f32, int ldx, const h16* W, const h16* bias, const h16* res, int ldr, int M,
                             int N, int K, int act, int pre_silu, void* out, int out_f32, int ldo, hipStream_t st);
int ladi_launch_nchw_to_nhwc(const void* src, int src_f32, int n, int C, int H, int W, h16* dst, int ld, hipStream_t st);
int ladi_launch_nhwc_to_nchw(const h16* src, int ld, int n, int C, int H, int W, void* dst, int dst_f32, hipStream_t st);
// sinusoidal timestep embedding (flip_sin_to_cos=True, freq_shift=0): out[i][dim] fp32 for timesteps t[i]
int ladi_launch_timestep_embedding(const float* t, int count, int dim, float* out, hipStream_t st);

struct StepTable {            // one entry per scheduler evaluation, device resident (see sched.h)
    float c_x;                // multiplies current sample
    float c_e;                // multiplies the (possibly multistep-combined) epsilon
    float w[4];               // PLMS combination weights over [eps_now, ets[-1], ets[-2], ets[-3]]
    int   mode;               // 0 plain (use w), 1 = PLMS 2nd evaluation (eps'=(eps+ets[-1])/2, sample=cur_sample, no push)
    int   push;               // bit0: push eps_now into the ring; bits 4-5: slot to push into;
                              // bits 8-9 / 10-11 / 12-13: ring slots holding ets[-1] / ets[-2] / ets[-3] (before the push)
    int   save_cur;           // 1: save current sample as cur_sample before the update
    int   zero_cloth_next;    // 1: the NEXT evaluation must see zero cloth latents
    float in_scale_next;      // scheduler.scale_model_input of the NEXT evaluation: the UNet sees latents * in_scale_next (1 for DDIM / PNDM,
                              // 1 / sqrt(sigma^2 + 1) for LMS); the fp32 latents themselves stay unscaled
};
struct StepArgs {
    const h16* eps; int ld_eps;     // UNet output NHWC [2B or B][hw][ld_eps], channels 0..3 valid
    int B, hw, cfg;                 // cfg: 1 = rows [0,B) uncond, [B,2B) cond
    float guidance;
    float* latents;                 // fp32 [B][hw][4] in/out
    float* cur_sample;              // fp32 [B][hw][4] (PLMS)
    float* ets;                     // fp32 [4][B][hw][4] ring (PLMS); slot = (count) & 3
    const StepTable* table; int* step_idx;   // device step counter (read, then incremented by the kernel)
    h16* unet_in; int ld_in;        // next UNet input [2B or B][hw][ld_in]; channels 0..3 rewritten
    int cloth_ch0;                  // first cloth channel (27) ; zeroed when table says so (4 channels)
    // optional per-evaluation trace (parity tests): guided noise prediction and updated latents of evaluation i are written to
    // trace_*[i][B][hw][4] (fp32) for i < trace_cap; null = off
    float* trace_eps; float* trace_lat; int trace_cap;
};
int ladi_launch_sched_step(const StepArgs& a, hipStream_t st);

// static part of the 31-channel UNet input: mask, masked-image latents, pose, cloth; uncond half zero pose/cloth
int ladi_launch_assemble_static(h16* unet_in, int ld_in, int B, int hw, int cfg, const float* latents, const h16* mask_lat,
                                const float* masked_lat, const h16* pose, int pose_ch, const float* cloth_lat, int has_cloth,
                                float lat_scale, hipStream_t st);
// posterior sample: lat[b][hw][4] = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scaling ; moments NHWC [..][ldm] (8 ch),
// noise fp32 NCHW [B][4][h][w]
int ladi_launch_posterior_sample(const h16* moments, int ldm, const float* noise_nchw, int B, int hw, float scaling, float* lat,
                                 hipStream_t st);
// mask binarise (>=0.5 -> 1) at full res + masked image: img NHWC (ld) *= (mask<0.5); also writes binarised mask fp16 [B][H*W]
int ladi_launch_prepare_mask(const void* image_nchw, int img_f32, const void* mask_nchw, int mask_f32, int B, int H, int W,
                             h16* masked_img, int ld, h16* mask_bin, hipStream_t st);
// nearest downsample by integer factor s (top-left pixel): dst[b][h/s][w/s]
int ladi_launch_mask_down(const h16* src, int B, int H, int W, int s, h16* dst, hipStream_t st);
// bilinear /8 (align_corners=False == mean of centre 2x2) of pose [B][C][H][W] (NCHW) -> NHWC fp16 [B][h*w][C]
int ladi_launch_pose_down8(const void* pose_nchw, int f32, int B, int C, int H, int W, h16* dst, hipStream_t st);
// image = clamp(x/2+0.5, 0, 1) : src NHWC fp16 (ld) 3 valid channels -> fp32 [B][H][W][3], or (dst_u8) uint8 = round(image * 255)
int ladi_launch_image_post(const h16* src, int ld, int n_pix, void* dst, int dst_u8, hipStream_t st);
// features[i] *= (1-mask) standalone (mask_features for the module-by-module shim path)
int ladi_launch_mask_mul(h16* feat, int C, int n_pix, const h16* mask, hipStream_t st);
// one wave spins for wall_ticks ticks of the 100 MHz counter; out2 (device) = {shader cycles, wall ticks}
int ladi_launch_clock_probe(unsigned long long wall_ticks, unsigned long long* out2, hipStream_t st);
int ladi_launch_fill_f32(float* p, size_t n, float v, hipStream_t st);
int ladi_launch_scale_h16(const h16* src, int lds_, h16* dst, int ldd, size_t n_pix, int C, float s, hipStream_t st);
// CLIP text embeddings + pseudo-word splice: ids [B][T] (device), first [B] = position of the sentence's first '$' or -1,
// wemb fp16 [B][nv][H] or null; out [B][T][H] = (token | pseudo-word) embedding + position embedding
int ladi_launch_text_embed(const int* ids, const int* first, int nv, const h16* tok, const h16* pos, const h16* wemb, int B, int T,
                           int H, int vocab, h16* out, hipStream_t st);
// per sentence: first[b] = position of the first `vstar` id (or -1; -1 too when use_words == 0), eot[b] = b*T + argmax(ids[b]) (first maximum)
int ladi_launch_text_meta(const int* ids, int B, int T, int vstar, int use_words, int* first, int* eot, hipStream_t st);
// TPS matching network helpers: per-channel affine (BatchNorm after ReLU), per-pixel L2 normalisation over channels, TPS grid
int ladi_launch_channel_affine(const h16* x, int ldx, size_t n_pix, int C, const float* scale, const float* shift, h16* y, int ldy, hipStream_t st);
int ladi_launch_l2norm_rows(const h16* x, int ldx, int rows, int C, h16* y, int ldy, hipStream_t st);
// coor [B][N][2], inv [(N+3)^2], ctrl [N][2] (all fp32, device) -> grid [B][H][W][2] fp32
int ladi_launch_tps_grid(const float* coor, const float* inv, const float* ctrl, int N, int B, int H, int W, float* grid, hipStream_t st);
// refinement UNet helpers (NHWC fp16, C % 8 == 0): 2x2 max pooling; bilinear x2 upsampling with align_corners=True
int ladi_launch_maxpool2(const h16* src, int lds_, int n, int H, int W, int C, h16* dst, int ldd, hipStream_t st);
int ladi_launch_upsample2x_bilinear_ac(const h16* src, int lds_, int n, int H, int W, int C, h16* dst, int ldd, hipStream_t st);
// glue of the warping module (src/inference.py:242-260), NCHW planes, fp32 or fp16 in / out:
// torchvision resize(BILINEAR, antialias=True) == aten _upsample_bilinear2d_aa (align_corners=False)
struct ResizeEpi { int on, C; float pre_mul, pre_add; float sub[4], div[4]; float quant; };   // value epilogue, see elementwise.hip (quant > 0: floor(v * quant) / quant after the clamp)
int ladi_launch_resize_bilinear_aa(const void* src, int in_f32, int planes, int H, int W, void* dst, int out_f32, int Ho, int Wo,
                                   hipStream_t st, const ResizeEpi* epi = nullptr);
// F.grid_sample(x, grid, bilinear, padding_mode="border", align_corners=False); grid fp32 [B][Ho][Wo][2] (x, y)
int ladi_launch_grid_sample_border(const void* src, int in_f32, int B, int C, int H, int W, const float* grid, int Ho, int Wo, void* dst,
                                   int out_f32, hipStream_t st);
// ViT patch rows for the patch-embedding GEMM: out [B][1 + (S/ps)^2][KP] fp16, row 0 and the padding columns zero
int ladi_launch_patchify(const void* px, int in_f32, int B, int S, int ps, int KP, h16* out, hipStream_t st);
int ladi_launch_gather_rows(const h16* src, const int* rows, int n, int H, h16* dst, hipStream_t st);
// decoder input: post_quant_conv(lat / scaling_factor) -> NHWC fp16 padded to ld; pq = device [16 w | 4 b] or null (identity)
int ladi_launch_post_quant(const float* lat, const float* pq, float inv_sf, int n, h16* dst, int ld, hipStream_t st);
int ladi_launch_lat_nchw_to_pix(const float* src, int B, int hw, float scale, float* dst, hipStream_t st);
int ladi_launch_lat_pix_to_nchw(const float* src, int B, int hw, float* dst, hipStream_t st);

// ---- f32path.hip: fp32 kernels of the warping module (fp32 NHWC activations, fp32 weights, v_mfma_f32_32x32x2_f32)
struct ConvF32Args {
    const float* src0; const float* src1;   // up to two NHWC fp32 sources concatenated along channels
    int C0, C1, ld0, ld1;                   // channels (multiples of 8) and row strides (multiples of 4) of each source
    int Hs, Ws, Ho, Wo, P;                  // source / output spatial size per sample, output pixels = n * Ho * Wo
    int ksize, stride, pad;
    const float* W; int Q, K, ldw;          // [Q][ksize*ksize*(C0+C1)] tap-major / channel-minor (ldw = 0 -> K)
    long long bs_src0, bs_w, bs_out;        // batching over grid.z (element strides)
    const float* bias; int act;             // fp32 bias [Q] or null; LADI_ACT_NONE / RELU / TANH / SILU
    float* out; int ldo;
};
int ladi_launch_conv_f32(const ConvF32Args& a, int batch, hipStream_t st);
int ladi_launch_nchw_to_nhwc_f32(const void* src, int in_f32, int n, int C, int H, int W, float* dst, int ld, hipStream_t st);
int ladi_launch_nhwc_to_nchw_f32(const float* src, int ld, int n, int C, int H, int W, void* dst, int out_f32, hipStream_t st);
int ladi_launch_channel_affine_f32(float* x, int ld, size_t n_pix, int C, const float* scale, const float* shift, hipStream_t st);
int ladi_launch_l2norm_rows_f32(float* x, int ld, int rows, int C, hipStream_t st);
int ladi_launch_gather_rows_f32(const float* src, const int* rows, int n, int H, float* dst, hipStream_t st);
int ladi_launch_maxpool2_f32(const float* src, int lds_, int n, int H, int W, int C, float* dst, int ldd, hipStream_t st);
int ladi_launch_upsample2x_bilinear_ac_f32(const float* src, int lds_, int n, int H, int W, int C, float* dst, int ldd, hipStream_t st);
int ladi_launch_linear_f32(const float* x, int ldx, const float* W, const float* b, int M, int N, int K, int act, float* out, int ldo, hipStream_t st);

// ---- igemm per-launch timing hooks (HIP events on the launch stream); see igemm.hip
void ladi_igemm_profile_enable(int on);
void ladi_igemm_autotune(int on);   // measured tile-shape selection on first use of a problem shape (default on)
void ladi_igemm_splitk_two_pass(int on);   // 1: split-K launches use the separate reduce pass (A/B switch; default: in-launch combine)
int ladi_igemm_tuned_count();
int ladi_igemm_num_cfgs();
const char* ladi_igemm_cfg_symbol(int cfg);
int ladi_igemm_profile_collect(double* out, int n_out);
// the same records grouped by kernel SYMBOL (exact rocprofv3 name): lines "symbol\tms\tflop\tlaunches\n" into buf (truncated at n); does
// NOT clear the records (call before ladi_igemm_profile_collect)
int ladi_igemm_profile_symbols(char* buf, int n);
