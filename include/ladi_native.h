/*
 * libladi_native — C ABI of the MI355X-native LaDI-VTON denoising hot path.
 *
 * The reference (miccunifi/ladi-vton) has no FFI / operator registry: its seam is the Python nn.Module duck type that
 * StableDiffusionTryOnePipeline is constructed from (src/vto_pipelines/tryon_pipe.py:56-68,129-137; built at
 * src/inference.py:212-220).  Each group of entry points below replaces exactly one of those modules; the citation
 * next to it is the reference interface it stands in for.  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - Every function returns 0 on success and a negative code on failure; ladi_last_error() returns a message
 *     (thread local).  Nothing throws across the boundary.
 *   - Pointers named *_dev are DEVICE pointers owned by the caller (e.g. torch allocations); the library never frees
 *     them.  `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).  All work is
 *     enqueued asynchronously on that stream; no entry point synchronises except where stated.
 *   - dtype codes: 0 = float32, 1 = float16.
 *   - Handles are not thread-safe (one pipeline per process per GPU, as in the reference).
 *   - The library has NO CPU fallback: without a gfx950 device every compute entry point fails.
 */
#ifndef LADI_NATIVE_H
#define LADI_NATIVE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LADI_F32 0
#define LADI_F16 1

const char* ladi_last_error(void);
int ladi_version(void);
/* number of visible HIP devices (0 when there is no GPU); never fails */
int ladi_device_count(void);

/* ---------------------------------------------------------------------------------------------------------------
 * Weight store: host-side staging of a diffusers-format state_dict (replaces nn.Module.load_state_dict,
 * hubconf.py:28,39,55; key names per SURVEY.md App. A.6).  Data is copied; the caller may free its buffer.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct ladi_weights ladi_weights;
ladi_weights* ladi_weights_create(void);
int ladi_weights_add(ladi_weights* ws, const char* key, const void* host_data, int dtype, int ndim, const int64_t* shape);
int ladi_weights_count(const ladi_weights* ws);
void ladi_weights_destroy(ladi_weights* ws);

/* ---------------------------------------------------------------------------------------------------------------
 * UNet — replaces diffusers UNet2DConditionModel as built by hubconf.py:31-39 (in_channels = 31) and called at
 * tryon_pipe.py:732: unet(latent_model_input, t, encoder_hidden_states=prompt_embeds).sample
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int in_channels, out_channels;
    int block_out_channels[4];
    int num_heads[4];            /* diffusers "attention_head_dim" used as head COUNT; head dim must be 64 */
    int layers_per_block;
    int cross_attention_dim;
    int norm_num_groups;
    float norm_eps;
} ladi_unet_config;
typedef struct ladi_unet ladi_unet;
ladi_unet* ladi_unet_create(const ladi_unet_config* cfg, const ladi_weights* ws);
void ladi_unet_destroy(ladi_unet* u);
/* encoder_hidden_states [n, L, cross_attention_dim] fp16 dense; precomputes the cross-attention K/V of all blocks */
int ladi_unet_set_context(ladi_unet* u, const void* ehs_dev, int n, int L, void* stream);
/* sample [n, in_channels, h, w] NCHW (dtype), scalar timestep, out [n, out_channels, h, w] NCHW (out_dtype) */
int ladi_unet_forward(ladi_unet* u, const void* sample_dev, int dtype, int n, int h, int w, float timestep, void* out_dev,
                      int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * VAE — replaces src/models/AutoencoderKL.py AutoencoderKL.encode (:145-157) / .decode (:174-188) with the EMASC
 * wiring of src/models/vae.py Encoder.forward (:99-119) / Decoder.forward (:183-212).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int in_channels, out_channels, latent_channels;
    int block_out_channels[4];
    int layers_per_block;
    int norm_num_groups;
    float scaling_factor;
} ladi_vae_config;
typedef struct ladi_vae ladi_vae;
ladi_vae* ladi_vae_create(const ladi_vae_config* cfg, const ladi_weights* ws);
void ladi_vae_destroy(ladi_vae* v);
/* x [B,3,H,W] NCHW (dtype) -> moments [B, 2*latent, H/8, W/8] NCHW fp32 (quant_conv applied) and, when feats_dev is
 * non-NULL, the five intermediate features idx1..5 written NHWC fp16 into caller buffers
 * ([B,H,W,c0] [B,H,W,c0] [B,H/2,W/2,c0] [B,H/4,W/4,c1] [B,H/8,W/8,c2], c = block_out_channels). */
int ladi_vae_encode(ladi_vae* v, const void* x_dev, int dtype, int B, int H, int W, float* moments_dev, void* const* feats_dev,
                    void* stream);
/* z [B,latent,h,w] NCHW fp32 (already divided by scaling_factor, as vae.decode receives it), skips_dev = NULL or five
 * NHWC fp16 EMASC outputs (idx1..5 order) -> sample [B,3,8h,8w] NCHW (out_dtype), NOT post-processed. */
int ladi_vae_decode(ladi_vae* v, const float* z_dev, int B, int h, int w, const void* const* skips_dev, void* sample_dev,
                    int out_dtype, void* stream);
/* fp16-range guard of the decoder (SD VAE activations can leave the fp16 range with real checkpoints; the reference
 * runs src/models/AutoencoderKL.py:159-188 in whatever dtype the caller picked).  The decoder's residual stream is stored multiplied by
 * 2^-shift (GroupNorm is scale invariant, every branch into the stream is scaled in its producer's epilogue): the same function with
 * 2^shift more head-room.  shift = -1 (default): automatic -- decode at shift 0 and, if a GroupNorm saw non-finite statistics, again at 4,
 * then 8 (an error after that); shift >= 0: fixed.  ladi_vae_last_range_shift reports what the last decode used. */
int ladi_vae_set_range_shift(ladi_vae* v, int shift);
int ladi_vae_last_range_shift(const ladi_vae* v);

/* ---------------------------------------------------------------------------------------------------------------
 * EMASC — replaces src/models/emasc.py EMASC.forward (:37-40); mask_features (src/utils/data_utils.py:4-16) can be
 * fused by passing the binarised full-resolution mask.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { int n; int in_channels[8]; int out_channels[8]; } ladi_emasc_config;
typedef struct ladi_emasc ladi_emasc;
ladi_emasc* ladi_emasc_create(const ladi_emasc_config* cfg, const ladi_weights* ws);
void ladi_emasc_destroy(ladi_emasc* e);
/* feats_dev[i]: NHWC fp16 [B, hs[i], wss[i], in_channels[i]] ; outs_dev[i]: NHWC fp16 [B, hs[i], wss[i], out_channels[i]].
 * mask_dev: NULL, or [B, Hm, Wm] fp16 binary mask at full resolution (Hm = hs[0]): out *= (1 - nearest(mask)). */
int ladi_emasc_forward(ladi_emasc* e, const void* const* feats_dev, const int* hs, const int* wss, int B, const void* mask_dev,
                       int Hm, int Wm, void* const* outs_dev, void* stream);
/* stand-alone mask_features on one NHWC fp16 feature map (in place): feat *= (1 - nearest(mask -> h x w)) */
int ladi_mask_features(void* feat_dev, int B, int h, int w, int C, const void* mask_dev, int Hm, int Wm, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Inversion adapter — replaces src/models/inversion_adapter.py InversionAdapter.forward (:22-28), dims hubconf.py:17-24
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { int hidden, heads, mlp_dim, head_hidden, out_dim; float layer_norm_eps; } ladi_adapter_config;
typedef struct ladi_adapter ladi_adapter;
ladi_adapter* ladi_adapter_create(const ladi_adapter_config* cfg, const ladi_weights* ws);
void ladi_adapter_destroy(ladi_adapter* a);
/* x [B, T, hidden] fp16 dense -> out [B, out_dim] fp16 */
int ladi_adapter_forward(ladi_adapter* a, const void* x_dev, int B, int T, void* out_dev, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * CLIP text encoder with pseudo-word splice — replaces src/utils/encode_text_word_embedding.py encode_text_word_embedding
 * (:6-72; call site src/inference.py:291-295) and the transformers CLIPTextModel it drives (SD2 text encoder: 23 layers, 1024-d,
 * 16 heads, gelu MLP 4096, 77 positions, causal).  SURVEY.md §8(f) rank 1: the producer of `prompt_embeds`.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { int vocab_size, hidden, heads, mlp_dim, layers, max_positions, vstar_token_id; float layer_norm_eps; } ladi_text_config;
typedef struct ladi_text_encoder ladi_text_encoder;
/* weights: transformers-4.27 key layout (text_model.embeddings.*, text_model.encoder.layers.N.*, text_model.final_layer_norm.*);
 * the flattened layout without the text_model. prefix is accepted too */
ladi_text_encoder* ladi_text_encoder_create(const ladi_text_config* cfg, const ladi_weights* ws);
void ladi_text_encoder_destroy(ladi_text_encoder* t);
/* input_ids_host: [B][T] int32 in HOST memory (tokenizer output); word_embeddings_dev: fp16 [B][num_vstar][hidden] or NULL (no splice).
 * In every sentence containing vstar_token_id ('$' = 259) the num_vstar positions starting at its FIRST occurrence are replaced by that
 * sentence's pseudo-word embeddings (:12-35); slots running past T are an error (the reference raises IndexError).
 * out_hidden_dev: fp16 [B][T][hidden] = final_layer_norm(encoder output) (:56-57); out_pooled_dev: fp16 [B][hidden] = row at
 * argmax(input_ids) (:62-65), or NULL. */
int ladi_text_encoder_forward(ladi_text_encoder* t, const int* input_ids_host, int B, int T, const void* word_embeddings_dev,
                              int num_vstar, void* out_hidden_dev, void* out_pooled_dev, void* stream);
/* the same with input_ids already in DEVICE memory (the reference moves them there before the call, src/inference.py:291): the first '$'
 * and the end-of-text row are found by a kernel and nothing crosses the host, so the call never synchronises.  No host-side validation in
 * this form: ids are clamped to the vocabulary by the lookup and slots running past T are cut instead of raising. */
int ladi_text_encoder_forward_dev(ladi_text_encoder* t, const int* input_ids_dev, int B, int T, const void* word_embeddings_dev,
                                  int num_vstar, void* out_hidden_dev, void* out_pooled_dev, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * CLIP ViT-H/14 vision encoder — replaces the transformers CLIPVisionModelWithProjection forward the reference calls at
 * src/inference.py:269-273 (`vision_encoder(pixel_values).last_hidden_state`, fed to the inversion adapter at :276).
 * SURVEY.md §8(f) rank 2.  32 layers, 1280-d, 16 heads of 80, gelu MLP 5120, 224x224 / patch 14 -> 257 tokens.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { int hidden, heads, mlp_dim, layers, image_size, patch_size; float layer_norm_eps; } ladi_vision_config;
typedef struct ladi_vision_encoder ladi_vision_encoder;
/* weights: transformers-4.27 key layout (vision_model.embeddings.*, vision_model.pre_layrnorm.*, vision_model.encoder.layers.N.*,
 * vision_model.post_layernorm.*); the flattened layout without the vision_model. prefix is accepted too; visual_projection unused */
ladi_vision_encoder* ladi_vision_encoder_create(const ladi_vision_config* cfg, const ladi_weights* ws);
void ladi_vision_encoder_destroy(ladi_vision_encoder* v);
/* pixel_values_dev: [B,3,S,S] (dtype: 0 fp32, 1 fp16), already CLIP-normalised (the CLIPProcessor stays on the caller's side);
 * out_hidden_dev: fp16 [B][1 + (S/ps)^2][hidden] = last_hidden_state (encoder output, NO post_layernorm);
 * out_pooled_dev: fp16 [B][hidden] = post_layernorm(last_hidden_state[:, 0]) (pooler_output) or NULL */
int ladi_vision_encoder_forward(ladi_vision_encoder* v, const void* pixel_values_dev, int dtype, int B, void* out_hidden_dev,
                                void* out_pooled_dev, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Refinement UNet of the warping module — replaces src/models/UNet.py UNetVanilla.forward (:23-34; parts in src/models/unet_parts.py)
 * as instantiated by hubconf.py:57 (24 -> 3 channels, bilinear=True) and called at src/inference.py:264.  SURVEY.md §8(f) rank 3
 * (first half; the TPS matching network is not native yet).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { int in_channels, out_channels, base_channels; float bn_eps; } ladi_refine_config;
typedef struct ladi_refine ladi_refine;
/* weights: the 'refinement' state_dict of the released warping checkpoint (hubconf.py:62): inc.double_conv.*, downN.maxpool_conv.1.*,
 * upN.conv.double_conv.*, outc.conv.*; BatchNorm running statistics are folded into the convolutions (inference mode) */
ladi_refine* ladi_refine_create(const ladi_refine_config* cfg, const ladi_weights* ws);
void ladi_refine_destroy(ladi_refine* r);
/* x_dev: [B, in_channels, H, W] NCHW (dtype 0 fp32 / 1 fp16), H and W multiples of 16; out_dev: [B, out_channels, H, W] NCHW (out_dtype) */
int ladi_refine_forward(ladi_refine* r, const void* x_dev, int dtype, int B, int H, int W, void* out_dev, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * TPS geometric-matching network of the warping module — replaces the inference data flow of src/models/ConvNet_TPS.py
 * ConvNet_TPS.forward (:315-337) as instantiated by hubconf.py:56 (256x192, input_nc 21, n_layer 3) and called at
 * src/inference.py:253 (`low_grid, theta, ... = tps(low_cloth, agnostic)`).  SURVEY.md §8(f) rank 3 (second half).
 * The training-only regulariser outputs (rx, ry, cx, cy, rg, cg; :201-224) are not produced.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { int height, width, input_nc, n_layers, grid_size, ngf; float bn_eps; } ladi_tps_config;
typedef struct ladi_tps ladi_tps;
/* weights: the 'tps' state_dict of the released warping checkpoint (hubconf.py:61); gridGen.* buffers are recomputed, not read */
ladi_tps* ladi_tps_create(const ladi_tps_config* cfg, const ladi_weights* ws);
void ladi_tps_destroy(ladi_tps* t);
/* input_a_dev [B,3,H,W], input_b_dev [B,input_nc,H,W] NCHW (dtype 0 fp32 / 1 fp16); grid_dev: fp32 [B,H,W,2] sampling grid in [-1,1]
 * (x, y) as F.grid_sample expects; coor_dev: fp32 [B, grid_size^2, 2] source control points (theta) or NULL */
int ladi_tps_forward(ladi_tps* t, const void* input_a_dev, const void* input_b_dev, int dtype, int B, float* grid_dev, float* coor_dev,
                     void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Scheduler — replaces diffusers DDIMScheduler / PNDMScheduler (skip_prk_steps) / LMSDiscreteScheduler set_timesteps +
 * scale_model_input + step (tryon_pipe.py:62,424,650-651,722,740).  kind: 0 = DDIM, 1 = PNDM, 2 = LMSDiscrete.
 * ------------------------------------------------------------------------------------------------------------- */
/* host helper: writes the N (DDIM) or N+1 (PNDM) timesteps; returns their count, or negative on error */
int ladi_sched_timesteps(int kind, int num_inference_steps, int* timesteps_out, int cap);
/* host helper, LMSDiscrete (order 4): timesteps_out[N] (fractional, float64 as diffusers holds them), sigmas_out[N + 1] (trailing 0;
 * init_noise_sigma = sigmas_out[0]), coeffs_out[N][4] = linear-multistep weights of evaluation i over its derivatives
 * [d_i, d_{i-1}, d_{i-2}, d_{i-3}] (zero beyond the order min(i + 1, 4)).  alphas_cumprod_host null = default.  Any output may be null. */
int ladi_sched_lms(int num_inference_steps, const float* alphas_cumprod_host, double* timesteps_out, float* sigmas_out, float* coeffs_out);
/* host helper: default alphas_cumprod (scaled_linear 0.00085..0.012, 1000 steps), out[1000] */
int ladi_sched_alphas_cumprod(float* out);

/* ---------------------------------------------------------------------------------------------------------------
 * Whole-loop fast path — StableDiffusionTryOnePipeline.__call__ steps 4-11 (tryon_pipe.py:630-753) in one call.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int batch, height, width;
    int in_dtype;                      /* dtype of image / mask_image / pose_map / warped_cloth */
    const void* image_dev;             /* [B,3,H,W] in [-1,1] */
    const void* mask_image_dev;        /* [B,1,H,W] in [0,1] (binarised at 0.5 like prepare_mask_and_masked_image) */
    const void* pose_map_dev;          /* [B,P,H,W] */
    const void* warped_cloth_dev;      /* [B,3,H,W] or NULL (cloth_input_type == 'none') */
    int pose_channels;
    const void* prompt_embeds_dev;           /* [B,L,D] fp16 */
    const void* negative_prompt_embeds_dev;  /* [B,L,D] fp16 (required when guidance_scale > 1) */
    int L;
    const float* noise_cloth_dev;      /* fp32 [B,4,h,w]: the three generator draws in pipeline order */
    const float* noise_latents_dev;
    const float* noise_masked_dev;
    int num_inference_steps;
    float guidance_scale;
    int scheduler;                     /* 0 DDIM, 1 PNDM, 2 LMSDiscrete */
    int cloth_zero_from_eval;          /* first evaluation index i that sees zero cloth latents: the smallest i with
                                        * i >= num_inference_steps - (1 - cloth_cond_rate) * num_inference_steps, evaluated by the CALLER in
                                        * float64 exactly like tryon_pipe.py:654,718 (a float32 rate crossing the ABI shifts the cut-off by
                                        * one step for rates such as 0.2 / 0.4 / 0.6 / 0.8); >= the evaluation count: never */
    int no_pose;
    int use_graph;                     /* capture the denoising step into a hipGraph and replay it */
    const float* alphas_cumprod_host;  /* optional [1000] override */
} ladi_tryon_inputs;
typedef struct ladi_tryon ladi_tryon;
/* emasc may be NULL (pipeline without EMASC); handles stay owned by the caller */
ladi_tryon* ladi_tryon_create(ladi_unet* unet, ladi_vae* vae, ladi_emasc* emasc);
void ladi_tryon_destroy(ladi_tryon* t);
/* images_dev: fp32 [B,H,W,3] in [0,1] (decode_latents layout, tryon_pipe.py:356-358); latents_dev: optional fp32 [B,4,h,w] */
int ladi_tryon_run(ladi_tryon* t, const ladi_tryon_inputs* in, float* images_dev, float* latents_dev, void* stream);
/* the same call with the batch in the dtype numpy_to_pil produces (tryon_pipe.py:357-360): uint8 [B,H,W,3] = round(image * 255),
 * round-half-to-even like numpy -- what the RCCL all-gather of the sharded path and the JPEG encoder consume */
int ladi_tryon_run_u8(ladi_tryon* t, const ladi_tryon_inputs* in, unsigned char* images_dev, float* latents_dev, void* stream);
/* per-evaluation trace for parity tests (tryon_pipe.py:732-740 intermediate values): subsequent runs write the guided noise prediction
 * and the updated latents of evaluation i to *_trace_dev[i] (fp32 [B, h*w, 4] each) for i < cap_evals; NULL pointers switch it off.
 * Buffers are caller-owned and must outlive the runs. */
int ladi_tryon_set_trace(ladi_tryon* t, float* eps_trace_dev, float* latents_trace_dev, int cap_evals);
/* stage times (ms) of the last run: [0] preprocess + VAE encodes + EMASC, [1] denoising loop, [2] decode. Sync first. */
int ladi_tryon_stage_ms(ladi_tryon* t, float* out3);
/* fp16-range guard of the decode, examined WITHOUT a host round trip inside ladi_tryon_run (round 6): a run decodes once and queues the guard's flag
 * behind its last kernel.  Returns 0 if the last run's decode stayed inside the fp16 range (or nothing is pending), 1 if it did not -- that run's
 * images are invalid; with ladi_vae_set_range_shift(-1) (automatic, the default) the shift has been raised (0 -> 4 -> 8) and re-submitting the
 * batch gives the result -- and < 0 on error.  Waits for the queued flag copy (i.e. for the run to finish).  A caller that never asks is told by
 * the NEXT ladi_tryon_run, which then fails with -101 instead of running. */
int ladi_tryon_poll_overflow(ladi_tryon* t);
/* sample-group lanes of the denoising loop: the UNet forward of the 2B (CFG) or B samples runs as `lanes` independent forwards on as
 * many HIP streams inside one hipGraph (csrc/runtime.h UNetLanes).  0 = default (environment LADI_UNET_LANES, else 1); a count that
 * does not divide the sample count falls back to the default rule.  Results do not depend on it beyond fp16 rounding of other tile
 * selections.  ladi_tryon_lanes() returns the count the last run used. */
int ladi_tryon_set_lanes(ladi_tryon* t, int lanes);
int ladi_tryon_lanes(ladi_tryon* t);
/* run ONLY `iters` UNet forwards (n samples of h x w latents, context already set) bracketed by HIP events on `stream`
 * and return the average milliseconds per forward (synchronises). Used by bench.py for the roofline figure. */
int ladi_unet_time_forward(ladi_unet* u, int n, int h, int w, int iters, float* avg_ms, void* stream);
/* the same measurement of the forward as the denoising loop runs it: `lanes` independent sample groups on as many HIP streams (0 =
 * the loop's own choice, LADI_UNET_LANES or 1; must divide n), replayed from one hipGraph with `lanes` parallel branches when
 * use_graph != 0.  Runs on an internal stream fenced against `stream`; synchronises. */
int ladi_unet_time_forward_lanes(ladi_unet* u, int n, int h, int w, int iters, int lanes, int use_graph, float* avg_ms, void* stream);

/* per-launch HIP-event timing of the implicit-GEMM kernel family (the dominant kernel): enable, run any entry point,
 * then collect: out[cfg*3 + {0,1,2}] = {total ms, algorithmic FLOP = 2*P*Q*K, launches} for tile configuration cfg = 1..ladi_igemm_cfg_count()
 * (igemm tile shapes, split-K variants, the X-stationary linear kernel; table in csrc/igemm.hip), index 0 = all; entries beyond
 * n_out are dropped. collect() synchronises and clears the records. */
/* measured tile-shape selection (default on): the first launch of a new problem shape outside a stream capture times the
 * admissible tile configurations and caches the fastest; off = static cost model */
void ladi_igemm_set_autotune(int on);
/* split-K launches combine their K slices inside the launch (last-arriving slice of a tile runs the fused epilogue; csrc/igemm_common.h);
 * on = 1 switches to the separate reduce pass of rounds 1-3 (A/B measurements, parity tests of one form against the other) */
void ladi_igemm_set_splitk_two_pass(int on);
void ladi_profile_igemm_enable(int on);
int ladi_profile_igemm_collect(double* out, int n_out);
/* the same records grouped by the exact kernel symbol rocprofv3 reports (the X-stationary kernel has several): text lines
 * "symbol\tms\tflop\tlaunches\n" into buf (at most n bytes, NUL-terminated); returns the untruncated length.  Call BEFORE collect(). */
int ladi_profile_igemm_symbols(char* buf, int n);
/* number of tile configurations (valid ids 1..count) and the kernel symbol configuration `cfg` launches, as rocprofv3 names it
 * (split-K variants share the symbol of their base tile); "" for an unknown id.  The string is owned by the library. */
int ladi_igemm_cfg_count(void);
const char* ladi_igemm_cfg_symbol_name(int cfg);

/* ---------------------------------------------------------------------------------------------------------------
 * Op-level entry points (kernel parity tests; NHWC fp16 device tensors)
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* src0; const void* src1; int C0, C1, ld0, ld1, Hs, Ws, Ho, Wo, P, ksize, stride, pad, ups;
    const void* W; int Q, K, ldw; long long bs_src0, bs_w, bs_out, bs_res;
    const void* bias; int bias_per_pixel; const float* rowadd; const int* rowadd_idx; int rowadd_stride; int act; float out_scale;
    const void* res0; const void* res1; int ldr0, ldr1; const void* mask; void* out; int ldo; int out_f32;
    float* stats; int stats_groups; int splitk, tile_map /* both ignored: set by the launcher */;
    const void* ln_gamma; const void* ln_beta; float ln_eps; float bias_mul /* multiplier of bias, 0 = 1 */; void* ln_scratch;
    float* sk_ws; int* sk_cnt;   /* both ignored: set by the launcher (in-launch split-K combine) */
    const float* gn_ss; int gn_hw; /* optional GroupNorm affine of the pixel operand: [n][C0][2] floats (scale, shift), pixels per sample
                                      (multiple of 32); K = 320 / 640 projections without residual / GEGLU only */
    /* optional LayerNorm of the pixel operand (single source, 1x1): fused into the X-stationary linear kernel where the tuner finds that
       faster, else run as its own kernel into ln_scratch ([P][C0] fp16, caller-provided; null = only the fused form is admissible) */
} ladi_igemm_desc;
int ladi_op_igemm(const ladi_igemm_desc* d, int batch, int tile_cfg, void* stream);
int ladi_op_group_norm(const void* src0, int C0, const void* src1, int C1, int n, int HW, int groups, const void* gamma,
                       const void* beta, float eps, int silu, const void* add, void* out, float* stats_scratch, void* stream);
int ladi_op_layer_norm(const void* x, const void* gamma, const void* beta, float eps, int rows, int C, void* out, void* stream);
/* Fused sub-blocks of diffusers' BasicTransformerBlock on the 320-channel level (5 heads of 64), fp16 operands in torch layout:
 *   xattn: out = x + to_out(softmax(to_q(LayerNorm(x)) K^T / 8) V) with kv = [n][L][640] rows (K | V of the context, L <= 96), x / out [n*T][320],
 *          T % 128 == 0 (norm2 / attn2 of the block: tryon_pipe.py:732 -> UNet2DConditionModel -> Transformer2DModel);
 *   ff:    out = x + W2 ((u + bu) * gelu(g + bg)) + bo with (u | g) = W1 LayerNorm(x), W1 / b1 in the GEGLU packing of ladi_op_igemm (32-row blocks
 *          alternate value | gate rows), w2 = [320][1280], P % 128 == 0 (norm3 / ff).  One launch each (ladi_vton_amd/csrc/xf_fused.hip). */
int ladi_op_xattn_block(const void* x, const void* ln_gamma, const void* ln_beta, float eps, const void* wq, const void* kv, int L, const void* wo,
                        const void* bo, int n, int T, void* out, void* stream);
int ladi_op_ff_block(const void* x, const void* ln_gamma, const void* ln_beta, float eps, const void* w1_geglu, const void* b1_geglu, const void* w2,
                     const void* bo, int P, void* out, void* stream);
int ladi_op_attention(const void* q, const void* k, const void* v, void* o, int ldq, int ldk, int ldv, int ldo, long long sq,
                      long long sk, long long sv, long long so, int n, int heads, int Nq, int Nk, float scale, void* stream);
/* same with causal = 1: query i attends to keys <= i (the CLIP text encoder's mask; Nq == Nk) */
int ladi_op_attention_causal(const void* q, const void* k, const void* v, void* o, int ldq, int ldk, int ldv, int ldo, long long sq,
                             long long sk, long long sv, long long so, int n, int heads, int Nq, int Nk, float scale, int causal,
                             void* stream);
/* heads of dimension head_dim in {64, 80, 96, 128} at column offset h*head_dim (ViT-H: 80) */
int ladi_op_attention_generic(const void* q, const void* k, const void* v, void* o, int ldq, int ldk, int ldv, int ldo, long long sq,
                              long long sk, long long sv, long long so, int n, int heads, int head_dim, int Nq, int Nk, float scale,
                              void* stream);
/* ONE wide head (head_dim 128 / 256 / 512) — the VAE mid-block AttentionBlock (diffusers 0.14; src/models/vae.py:66-75 builds it through
 * UNetMidBlock2D): flash-style, scores never materialised.  q, k: [n][N][ld]; vt = V TRANSPOSED [n][head_dim][ldvt >= Nk]; Nk % 4 == 0 */
int ladi_op_attention_wide(const void* q, const void* k, const void* vt, void* o, int ldq, int ldk, int ldvt, int ldo, long long sq,
                           long long sk, long long svt, long long so, int n, int head_dim, int Nq, int Nk, float scale, void* stream);
/* glue between the TPS network and the refinement UNet (src/inference.py:242-260), NCHW planes, dtype 0 fp32 / 1 fp16 in and out:
 * torchvision.transforms.functional.resize(x, size, BILINEAR, antialias=True) over `planes` = B*C images of H x W */
int ladi_op_resize_bilinear_aa(const void* src, int dtype, int planes, int H, int W, void* dst, int out_dtype, int Ho, int Wo, void* stream);
/* F.grid_sample(x [B,C,H,W], grid fp32 [B,Ho,Wo,2], mode bilinear, padding_mode "border", align_corners False) -> [B,C,Ho,Wo] */
int ladi_op_grid_sample_border(const void* src, int dtype, int B, int C, int H, int W, const float* grid, int Ho, int Wo, void* dst,
                               int out_dtype, void* stream);
/* shader-clock probe for the bench's roofline bookkeeping: one wave spins for `wall_ticks_100mhz` ticks of the constant 100 MHz counter
 * (at most 10^8 = 1 s) on `stream` -- a side stream, next to the work being measured -- and writes {shader cycles, wall ticks} to
 * out2_dev (2 x uint64, device): cycles / ticks * 100 = the MHz the chip sustained under that load */
int ladi_clock_probe(unsigned long long wall_ticks_100mhz, unsigned long long* out2_dev, void* stream);
/* CLIP image pre-processing of the in-shop cloth in one pass (src/inference.py:268-272): v = resize((x + 1) / 2, (size, size), antialias)
 * .clamp(0, 1), then the processor's 8-bit round trip v = floor(v * 255) / 255, then (v - mean[c]) / std[c]; src [B,3,H,W] in [-1,1]
 * (dtype 0 fp32 / 1 fp16), dst fp16 [B,3,size,size]; mean3 / std3 host.
 * The round trip is what CLIPImageProcessor of the pinned transformers 4.27.3 (environment.yml:92) does to a FLOAT32 image on its way
 * through to_pil_image -- i.e. the reference's default run (inference.py:186: weight_dtype = float32 unless --mixed_precision is given).
 * It is applied for either input dtype: with --mixed_precision fp16 the reference hands the processor a float16 array, which 4.27.3's
 * to_pil_image does not recognise as floating point (its isinstance test lists float / np.float32 / np.float64 only) and truncates
 * without the x255 rescale -- a reference-side accident this library does not reproduce; 4.27.3 is not installable in the build image
 * (5.x is), so neither branch could be run against the real processor here. */
int ladi_op_clip_preprocess(const void* src, int dtype, int B, int H, int W, int size, const float* mean3, const float* std3, void* dst_f16,
                            void* stream);
/* NHWC fp16 helpers of the refinement UNet: 2x2 max pooling, bilinear x2 upsampling with align_corners=True (C % 8 == 0) */
int ladi_op_maxpool2(const void* src, int n, int H, int W, int C, void* dst, void* stream);
int ladi_op_upsample2x_bilinear(const void* src, int n, int H, int W, int C, void* dst, void* stream);
int ladi_op_softmax_rows(const float* S, int rows, int cols, float scale, void* P, void* stream);
int ladi_op_small_linear(const void* x, int x_f32, int ldx, const void* W, const void* bias, const void* res, int ldr, int M, int N,
                         int K, int act, int pre_silu, void* out, int out_f32, int ldo, void* stream);
int ladi_op_nchw_to_nhwc(const void* src, int dtype, int n, int C, int H, int W, void* dst, int ld, void* stream);
int ladi_op_nhwc_to_nchw(const void* src, int ld, int n, int C, int H, int W, void* dst, int dtype, void* stream);
/* one scheduler evaluation on device tensors (fused CFG + DDIM/PLMS update), for scheduler parity tests:
 * runs evaluations [0, evals) feeding eps_seq[i] ([2B or B][hw][4] fp16 NHWC per evaluation); latents fp32 [B][hw][4] in/out */
int ladi_op_sched_run(int kind, int steps, const float* alphas_cumprod_host, const void* eps_seq_dev, int evals, int B, int hw,
                      int cfg, float guidance, float* latents_dev, void* stream);
/* pipeline pre-processing kernels, one entry point per kernel so each can be checked on its own:
 * prepare_mask_and_masked_image (diffusers tensor branch; tryon_pipe.py:630): mask binarised at 0.5 -> mask_bin_dev fp16 [B,H,W];
 *   masked_image_dev NHWC fp16 [B,H,W,ld] (3 valid channels, the rest zero) = image * (mask < 0.5) */
int ladi_op_prepare_mask(const void* image_dev, const void* mask_dev, int dtype, int B, int H, int W, void* masked_image_dev, int ld,
                         void* mask_bin_dev, void* stream);
/* F.interpolate(mask, size=(H/s, W/s)) (nearest; prepare_mask_latents tryon_pipe.py:424-427, mask_features data_utils.py:11) on fp16 [B,H,W] */
int ladi_op_mask_down(const void* mask_dev, int B, int H, int W, int s, void* out_dev, void* stream);
/* F.interpolate(pose_map, size=(H/8, W/8), mode="bilinear") (tryon_pipe.py:632-634): NCHW in (dtype) -> NHWC fp16 [B, H/8 * W/8, C] */
int ladi_op_pose_down8(const void* pose_dev, int dtype, int B, int C, int H, int W, void* out_dev, void* stream);
/* scaling_factor * DiagonalGaussianDistribution(moments).sample() (vae.py:329-348; tryon_pipe.py:640,647) with the generator draw passed in:
 * moments NHWC fp16 [B, hw, ldm] (8 channels: mean | logvar), noise fp32 NCHW [B,4,h,w] -> latents fp32 [B, hw, 4] */
int ladi_op_posterior_sample(const void* moments_dev, int ldm, const float* noise_dev, int B, int hw, float scaling, float* lat_dev,
                             void* stream);
/* 31-channel UNet input assembly (tryon_pipe.py:702-729): [latents(4) | mask(1) | masked-image latents(4) | pose(P) | cloth latents(4)],
 * CFG batch order [uncond(B); cond(B)] with zero pose / cloth in the uncond half; unet_in NHWC fp16 [(cfg ? 2B : B), hw, ld] */
int ladi_op_assemble_input(void* unet_in_dev, int ld, int B, int hw, int cfg, const float* latents_dev, const void* mask_lat_dev,
                           const float* masked_lat_dev, const void* pose_dev, int pose_channels, const float* cloth_lat_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
