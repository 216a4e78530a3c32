// Native try-on pipeline: steps 4b-11 of StableDiffusionTryOnePipeline.__call__
// (src/vto_pipelines/tryon_pipe.py:630-753) with the denoising step hipGraph-captured.
#include "runtime.h"
#include <stdexcept>
#include <cstring>
#include <cmath>

namespace ladi {

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

TryOn::~TryOn() {
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    if (stats) (void)hipFree(stats);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

int TryOn::stage_ms(float out[3]) {
    if (!ev_valid) return -1;
    for (int i = 0; i < 3; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) return -2;
        out[i] = ms;
    }
    return 0;
}

static unsigned long long mix(unsigned long long h, unsigned long long v) {
    h ^= v + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
    return h;
}

int TryOn::run(const TryOnInputs& in, void* images_out, int images_u8, float* latents_out, hipStream_t user_st) {
    if (!unet || !vae) { set_error("tryon: unet and vae are required"); return -1; }
    hipStream_t st = user_st;
    try {
        if (!own_stream) {
            HIP_OK(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
        }
        HIP_OK(hipEventRecord(ev_in, user_st));
        HIP_OK(hipStreamWaitEvent(own_stream, ev_in, 0));
        st = own_stream;
    } catch (const std::exception& e) { set_error(std::string("tryon: ") + e.what()); return -100; }
    try {
        if (vae->poll_overflow()) {      // the PREVIOUS run's decode left the fp16 range: its images are invalid and nobody asked (ladi_tryon_poll_overflow)
            set_error("tryon: the previous run's VAE decode overflowed the fp16 range (non-finite GroupNorm statistics) at range shift " + std::to_string(vae->last_shift) +
                      "; its images are invalid -- re-submit that batch (the automatic shift is now " + std::to_string(vae->guard_shift()) + ")");
            (void)hipEventRecord(ev_out, st); (void)hipStreamWaitEvent(user_st, ev_out, 0);
            return -101;
        }
    } catch (const std::exception& e) { set_error(std::string("tryon: ") + e.what()); return -100; }
    const int B = in.batch, H = in.height, W = in.width;
    if (H % 8 || W % 8) { set_error("height and width must be divisible by 8"); return -2; }
    const int h = H / 8, w = W / 8, hw = h * w;
    const int cfgf = in.guidance > 1.0f ? 1 : 0;
    const int n = cfgf ? 2 * B : B;
    const bool has_cloth = in.warped_cloth != nullptr;
    const int pose_ch = in.pose_channels;
    const int in_ch = 9 + pose_ch + (has_cloth ? 4 : 0);
    if (in_ch != unet->cfg.in_channels) { set_error("tryon: UNet in_channels does not match 9 + pose + cloth channels"); return -3; }
    const int L = in.L, D = unet->cfg.cross_dim;
    if (cfgf && !in.negative_prompt_embeds) { set_error("tryon: negative_prompt_embeds required when guidance_scale > 1"); return -4; }

    // ---- scheduler tables (host)
    std::vector<float> ac;
    if (in.alphas_cumprod) ac.assign(in.alphas_cumprod, in.alphas_cumprod + 1000); else default_alphas_cumprod(ac);
    std::vector<double> timesteps; std::vector<StepTable> table; SchedInfo sinfo;
    build_step_table(in.scheduler, in.steps, ac.data(), in.cloth_zero_from, timesteps, table, &sinfo);
    const int evals = (int)timesteps.size();
    const bool cloth_zero_from_start = has_cloth && in.cloth_zero_from <= 0;
    last_evals = evals;

    if (!d_step) d_step = reinterpret_cast<int*>(pool.alloc(256));
    if (!sk_cnt) { sk_cnt = reinterpret_cast<int*>(pool.alloc(1024 * sizeof(int))); HIP_OK(hipMemset(sk_cnt, 0, 1024 * sizeof(int))); }
    if (evals > table_cap) { d_table = reinterpret_cast<StepTable*>(pool.alloc((size_t)evals * sizeof(StepTable))); table_cap = evals; }
    if (!ev[0]) for (auto& e : ev) HIP_OK(hipEventCreate(&e));

    try {
        lanes.configure(n, lanes_override > 0 && (n % lanes_override) == 0 ? lanes_override : 0);
        for (int pass = 0; pass < 2; ++pass) {
            arena.dry = (pass == 0);
            arena.off = 0;
            Ctx c; c.st = st; c.ar = &arena; c.stats = stats; c.stats_cap = stats_cap; c.sk_cnt = sk_cnt;
            if (pass == 1) {
                HIP_OK(hipMemcpyAsync(d_table, table.data(), (size_t)evals * sizeof(StepTable), hipMemcpyHostToDevice, st));
                HIP_OK(hipMemsetAsync(d_step, 0, 2 * sizeof(int), st));    // evaluation index + the step kernel's arrival ticket
                std::vector<float> tsf(timesteps.begin(), timesteps.end());
                if (unet->compute_temb(tsf.data(), evals, st)) return -5;
                HIP_OK(hipEventRecord(ev[0], st));
            }
            // ---------------- persistent buffers for this call
            h16* ehs = c.alloc_h16((size_t)n * L * D);
            h16* mask_bin = c.alloc_h16((size_t)B * H * W);
            h16* mask2 = c.alloc_h16((size_t)B * (H / 2) * (W / 2));
            h16* mask4 = c.alloc_h16((size_t)B * (H / 4) * (W / 4));
            h16* mask8 = c.alloc_h16((size_t)B * hw);
            h16* pose_lat = c.alloc_h16((size_t)B * hw * pose_ch);
            float* cloth_lat = c.alloc_f32((size_t)B * hw * 4);
            float* masked_lat = c.alloc_f32((size_t)B * hw * 4);
            float* latents = c.alloc_f32((size_t)B * hw * 4);
            float* cur_sample = c.alloc_f32((size_t)B * hw * 4);
            float* ets = c.alloc_f32((size_t)4 * B * hw * 4);
            Act unet_in = c.new_act(n, h, w, 64);
            Act skips[5];
            const bool use_emasc = emasc != nullptr;
            if (use_emasc) {
                const int sh[5] = {H, H, H / 2, H / 4, H / 8}, sw[5] = {W, W, W / 2, W / 4, W / 8};
                for (int i = 0; i < 5; ++i) skips[i] = c.new_act(B, sh[i], sw[i], emasc->cfg.out_ch[i]);
            }
            const size_t mk_persist = arena.mark();

            if (!c.dry()) {
                if (stats_cap) HIP_OK(hipMemsetAsync(stats, 0, stats_cap * sizeof(float), st));
                // prompt embeddings [uncond ; cond] (tryon_pipe.py:620-628)
                const size_t pe = (size_t)B * L * D * sizeof(h16);
                if (cfgf) {
                    HIP_OK(hipMemcpyAsync(ehs, in.negative_prompt_embeds, pe, hipMemcpyDeviceToDevice, st));
                    HIP_OK(hipMemcpyAsync(ehs + (size_t)B * L * D, in.prompt_embeds, pe, hipMemcpyDeviceToDevice, st));
                } else HIP_OK(hipMemcpyAsync(ehs, in.prompt_embeds, pe, hipMemcpyDeviceToDevice, st));
                if (unet->set_context(ehs, n, L, st)) return -6;
            }
            // ---------------- 4. mask / masked image / pose (tryon_pipe.py:630-636)
            Act masked_img = c.new_act(B, H, W, 64);
            if (!c.dry()) {
                c.check(ladi_launch_prepare_mask(in.image, in.in_f32, in.mask_image, in.in_f32, B, H, W, masked_img.p, 64, mask_bin, st), "prepare_mask");
                c.check(ladi_launch_mask_down(mask_bin, B, H, W, 2, mask2, st), "mask_down");
                c.check(ladi_launch_mask_down(mask_bin, B, H, W, 4, mask4, st), "mask_down");
                c.check(ladi_launch_mask_down(mask_bin, B, H, W, 8, mask8, st), "mask_down");
                if (in.no_pose) HIP_OK(hipMemsetAsync(pose_lat, 0, (size_t)B * hw * pose_ch * sizeof(h16), st));
                else c.check(ladi_launch_pose_down8(in.pose_map, in.in_f32, B, pose_ch, H, W, pose_lat, st), "pose_down8");
            }
            // ---------------- 4b. cloth latents (RNG draw #1)
            if (has_cloth) {
                const size_t mk = arena.mark();
                Act cloth = c.new_act(B, H, W, 64);
                if (!c.dry()) c.check(ladi_launch_nchw_to_nhwc(in.warped_cloth, in.in_f32, B, 3, H, W, cloth.p, 64, st), "nchw_to_nhwc");
                Act feats[5];
                c.stats_off = 0;
                Act mom = vae->encode(c, cloth, feats);
                if (!c.dry()) c.check(ladi_launch_posterior_sample(mom.p, mom.ld, in.noise_cloth, B, hw, vae->cfg.scaling_factor, cloth_lat, st), "posterior");
                arena.release(mk);
            }
            // ---------------- 6. initial latents (RNG draw #2) * init_noise_sigma (1 for DDIM / PNDM; tryon_pipe.py:424)
            if (!c.dry()) c.check(ladi_launch_lat_nchw_to_pix(in.noise_latents, B, hw, sinfo.init_noise_sigma, latents, st), "latents");
            // ---------------- 7. masked-image latents (RNG draw #3) + EMASC skips
            {
                Act feats[5];
                if (!c.dry()) if (stats_cap) HIP_OK(hipMemsetAsync(stats, 0, stats_cap * sizeof(float), st));
                c.stats_off = 0;
                Act mom = vae->encode(c, masked_img, feats);
                if (!c.dry()) c.check(ladi_launch_posterior_sample(mom.p, mom.ld, in.noise_masked, B, hw, vae->cfg.scaling_factor, masked_lat, st), "posterior");
                if (use_emasc) {
                    const h16* masks[5] = {mask_bin, mask_bin, mask2, mask4, mask8};
                    emasc->forward(c, feats, masks, skips, true);
                }
            }
            arena.release(mk_persist);
            // ---------------- 7a. static UNet input channels
            if (!c.dry()) {
                c.check(ladi_launch_assemble_static(unet_in.p, 64, B, hw, cfgf, latents, mask8, masked_lat, pose_lat, pose_ch,
                                                    (has_cloth && !cloth_zero_from_start) ? cloth_lat : nullptr, has_cloth ? 1 : 0,
                                                    sinfo.in_scale0, st), "assemble");
                HIP_OK(hipEventRecord(ev[1], st));
            }
            // ---------------- 9. denoising loop
            StepArgs sa; std::memset(&sa, 0, sizeof(sa));
            sa.B = B; sa.hw = hw; sa.cfg = cfgf; sa.guidance = in.guidance; sa.latents = latents; sa.cur_sample = cur_sample; sa.ets = ets;
            sa.table = d_table; sa.step_idx = d_step; sa.unet_in = unet_in.p; sa.ld_in = 64; sa.cloth_ch0 = 9 + pose_ch;
            sa.trace_eps = trace_eps; sa.trace_lat = trace_lat; sa.trace_cap = trace_cap;
            // the UNet forward runs as lanes.G independent sample groups on as many streams (runtime.h UNetLanes); the lanes own their
            // arenas, the shared noise prediction lives in this one
            const int eps_ld = (unet->cfg.out_channels + 3) / 4 * 4;
            Act eps = c.new_act(n, h, w, unet->cfg.out_channels, eps_ld);
            const size_t mk_loop = arena.mark();
            auto one_step = [&](bool concurrent) {
                arena.release(mk_loop);
                lanes.forward(*unet, st, c.dry(), concurrent, unet_in, eps, unet->temb_table, d_step);
                if (!c.dry()) { sa.eps = eps.p; sa.ld_eps = eps.ld; c.check(ladi_launch_sched_step(sa, st), "sched_step"); }
            };
            if (c.dry()) one_step(false);
            else if (!in.use_graph || evals < 3) { for (int i = 0; i < evals; ++i) one_step(i > 0); }
            else {
                one_step(false);  // eager first evaluation, lanes one after the other (one-time function attribute setup, per-shape tile measurement)
                unsigned long long key = 0x1234;
                key = mix(key, (unsigned long long)(uintptr_t)arena.base); key = mix(key, (unsigned long long)B * 1000003ULL + H * 4099ULL + W);
                key = mix(key, (unsigned long long)cfgf); key = mix(key, (unsigned long long)L);
                unsigned gb; std::memcpy(&gb, &in.guidance, 4); key = mix(key, gb);
                key = mix(key, (unsigned long long)(uintptr_t)unet->temb_table); key = mix(key, (unsigned long long)(uintptr_t)unet->mid_xf.kv_cache);
                key = mix(key, (unsigned long long)(uintptr_t)d_table); key = mix(key, (unsigned long long)(uintptr_t)stats);
                key = mix(key, (unsigned long long)pose_ch * 7 + has_cloth);
                key = mix(key, (unsigned long long)(uintptr_t)trace_eps); key = mix(key, (unsigned long long)(uintptr_t)trace_lat);
                key = mix(key, (unsigned long long)trace_cap);
                key = mix(key, lanes.key());
                if (!gexec || key != graph_key) {
                    if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
                    if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
                    HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
                    try { one_step(true); } catch (...) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(st, &g); if (g) (void)hipGraphDestroy(g); throw; }
                    HIP_OK(hipStreamEndCapture(st, &graph));
                    HIP_OK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
                    graph_key = key;
                }
                for (int i = 1; i < evals; ++i) HIP_OK(hipGraphLaunch(gexec, st));
            }
            arena.release(mk_loop);
            if (!c.dry()) HIP_OK(hipEventRecord(ev[2], st));
            // ---------------- 11. decode (tryon_pipe.py:349-359)
            {
                Act z = c.new_act(B, h, w, 64);
                if (!c.dry()) {
                    if (stats_cap) HIP_OK(hipMemsetAsync(stats, 0, stats_cap * sizeof(float), st));
                    c.check(ladi_launch_post_quant(latents, vae->d_pq, 1.0f / vae->cfg.scaling_factor, B * hw, z.p, 64, st), "post_quant");
                }
                // decode under the fp16-range guard (runtime.h VAE::range_shift): the planning pass reserves the guarded form's arena (a
                // superset: scaled copies of the skips); the real pass re-runs the decode with more head-room only if a GroupNorm of the
                // decoder saw non-finite statistics
                const size_t mk_dec = arena.mark();
                if (c.dry()) {
                    c.stats_off = 0;
                    (void)vae->decode(c, z, use_emasc ? skips : nullptr, vae->range_shift < 0 ? 4 : vae->range_shift);
                } else {
                    // ONE decode at the guard's current shift; the flag is examined without a host round trip (runtime.h VAE::post_overflow_check /
                    // poll_overflow: at the entry of the next run, or by ladi_tryon_poll_overflow)
                    const int sh = vae->guard_shift();
                    arena.release(mk_dec);
                    c.stats_off = 0;
                    if (stats_cap) HIP_OK(hipMemsetAsync(stats, 0, stats_cap * sizeof(float), st));
                    Act img = vae->decode(c, z, use_emasc ? skips : nullptr, sh);
                    c.check(ladi_launch_image_post(img.p, img.ld, B * H * W, images_out, images_u8, st), "image_post");
                    vae->last_shift = sh;
                    vae->post_overflow_check(st);
                    if (latents_out) c.check(ladi_launch_lat_pix_to_nchw(latents, B, hw, latents_out, st), "latents_out");
                    HIP_OK(hipEventRecord(ev[3], st));
                    ev_valid = true;
                }
            }
            if (pass == 0) {
                lanes.commit_plan();
                arena.reserve(arena.peak + 4096);
                if (c.stats_peak > stats_cap) {
                    if (stats) (void)hipFree(stats);
                    stats = nullptr;
                    HIP_OK(hipMalloc(reinterpret_cast<void**>(&stats), c.stats_peak * sizeof(float)));
                    stats_cap = c.stats_peak;
                }
            }
        }
        HIP_OK(hipEventRecord(ev_out, st));
        HIP_OK(hipStreamWaitEvent(user_st, ev_out, 0));
    } catch (const std::exception& e) {
        set_error(std::string("tryon: ") + e.what());
        (void)hipEventRecord(ev_out, st);
        (void)hipStreamWaitEvent(user_st, ev_out, 0);
        return -100;
    }
    return 0;
}

}  // namespace ladi
