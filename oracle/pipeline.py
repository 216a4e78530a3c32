"""CPU fp32 restatement of the schedulers (diffusers 0.14.0 DDIMScheduler / PNDMScheduler) and of
StableDiffusionTryOnePipeline.__call__ (src/vto_pipelines/tryon_pipe.py:494-765), plus the deterministic
synthetic inputs of SURVEY.md §8d.  Test infrastructure only (see oracle/__init__.py)."""
import math

import torch
import torch.nn.functional as F

from . import models as M


# ---------------------------------------------------------------------------------------------------------------
# schedulers
# ---------------------------------------------------------------------------------------------------------------
def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class DDIM:
    """diffusers DDIMScheduler(steps_offset=1, set_alpha_to_one=False, clip_sample=False, epsilon), eta = 0"""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.ac = alphas_cumprod()
        self.final_ac = self.ac[0]

    def set_timesteps(self, n):
        self.n = n
        self.ratio = 1000 // n
        self.timesteps = [int(i * self.ratio) + 1 for i in range(n)][::-1]

    def step(self, eps, t, x, eta=0.0, noise=None):
        """DDIM eq. (12); eta > 0 adds sigma_t * noise with sigma_t = eta * sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev))"""
        tp = t - self.ratio
        a_t = self.ac[t]
        a_p = self.ac[tp] if tp >= 0 else self.final_ac
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        std = eta * ((1 - a_p) / (1 - a_t) * (1 - a_t / a_p)) ** 0.5
        out = a_p ** 0.5 * x0 + (1 - a_p - std ** 2) ** 0.5 * eps
        return out + std * noise if eta > 0 else out


class PNDM:
    """diffusers PNDMScheduler(skip_prk_steps=True, steps_offset=1, set_alpha_to_one=False): PLMS only"""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.ac = alphas_cumprod()
        self.final_ac = self.ac[0]

    def set_timesteps(self, n):
        self.n = n
        self.ratio = 1000 // n
        ts = [int(i * self.ratio) + 1 for i in range(n)]
        seq = ts[:-1] + ts[-2:-1] + ts[-1:]
        self.timesteps = seq[::-1]
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def _prev(self, x, t, tp, e):
        a_t = self.ac[t]
        a_p = self.ac[tp] if tp >= 0 else self.final_ac
        b_t, b_p = 1 - a_t, 1 - a_p
        coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return coeff * x - (a_p - a_t) * e / denom

    def step(self, eps, t, x):
        tp = t - self.ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            tp = t
            t = t + self.ratio
        if len(self.ets) == 1 and self.counter == 0:
            e = eps
            self.cur_sample = x
        elif len(self.ets) == 1 and self.counter == 1:
            e = (eps + self.ets[-1]) / 2
            x = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            e = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            e = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            e = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        out = self._prev(x, t, tp, e)
        self.counter += 1
        return out


class LMS:
    """diffusers 0.14.0 LMSDiscreteScheduler(beta_schedule="scaled_linear", epsilon prediction) -- the third scheduler type
    StableDiffusionTryOnePipeline accepts (src/vto_pipelines/tryon_pipe.py:62).  diffusers is not installed here, so this restates
    the published algorithm (Katherine Crowson's k-diffusion linear multistep sampler as diffusers ships it):
      set_timesteps : timesteps = linspace(0, T-1, n)[::-1] (fractional, float64); sigma(t) = interp of sqrt((1-a)/a) (float32),
                      a trailing sigma 0; init_noise_sigma = max sigma
      scale_model_input : sample / sqrt(sigma_i^2 + 1)
      step          : derivative d_i = (sample - pred_x0) / sigma_i with pred_x0 = sample - sigma_i * eps; order = min(i + 1, 4);
                      prev = sample + sum_j c_ij * d_{i-j},  c_ij = integral over [sigma_i, sigma_{i+1}] of the Lagrange basis
                      polynomial of node sigma_{i-j} among {sigma_i .. sigma_{i-order+1}} (scipy.integrate.quad, epsrel 1e-4)
    Known-answer anchor (tests/test_cpu.py): init_noise_sigma = 14.6146 for the SD beta schedule."""
    order = 1

    def __init__(self):
        self.ac = alphas_cumprod()

    def set_timesteps(self, n):
        import numpy as np
        self.n = n
        ts = np.linspace(0, 999, n, dtype=float)[::-1].copy()
        sig = (((1 - self.ac) / self.ac) ** 0.5).numpy()                      # float32, as diffusers holds it
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = [float(t) for t in ts]
        self.init_noise_sigma = float(self.sigmas.max())
        self.derivatives = []
        self._i = 0

    def scale_model_input(self, x, t):
        i = self.timesteps.index(float(t))
        return x / ((float(self.sigmas[i]) ** 2 + 1) ** 0.5)

    def coefficient(self, order, i, j):
        from scipy import integrate
        s = [float(v) for v in self.sigmas]

        def basis(tau):
            prod = 1.0
            for k in range(order):
                if k != j:
                    prod *= (tau - s[i - k]) / (s[i - j] - s[i - k])
            return prod

        return integrate.quad(basis, s[i], s[i + 1], epsrel=1e-4)[0]

    def step(self, eps, t, x, order=4):
        i = self.timesteps.index(float(t))
        sigma = float(self.sigmas[i])
        x0 = x - sigma * eps
        self.derivatives.append((x - x0) / sigma)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [self.coefficient(order, i, j) for j in range(order)]
        return x + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))


def _identity_scale(self, x, t):
    return x


DDIM.scale_model_input = _identity_scale
PNDM.scale_model_input = _identity_scale


def make_scheduler(kind):
    if kind in (2, "lms"):
        return LMS()
    return DDIM() if kind in (0, "ddim") else PNDM()


# ---------------------------------------------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------------------------------------------
def synthetic_inputs(B, H, W, L=77, D=1024, pose_channels=18, seed=1234):
    def gen(s):
        return torch.Generator().manual_seed(seed + s)

    def smooth(s):
        low = torch.rand((B, 3, H // 8, W // 8), generator=gen(s)) * 2 - 1
        return F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False).clamp(-1, 1)

    image, cloth = smooth(0), smooth(1)
    mask = torch.zeros(B, 1, H, W)
    mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
    g = gen(2)
    ys = torch.arange(H, dtype=torch.float32)[:, None]
    xs = torch.arange(W, dtype=torch.float32)[None, :]
    pose = torch.zeros(B, pose_channels, H, W)
    for b in range(B):
        for c in range(pose_channels):
            if c % 9 == 7:
                continue  # missing joints -> all-zero channel
            cy = float(torch.rand((), generator=g)) * H
            cx = float(torch.rand((), generator=g)) * W
            pose[b, c] = torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (9.0 ** 2))
    prompt = torch.randn((B, L, D), generator=gen(3))
    neg = torch.randn((1, L, D), generator=gen(4)).expand(B, L, D).contiguous()
    gn = torch.Generator().manual_seed(seed)
    h, w = H // 8, W // 8
    noise = [torch.randn((B, 4, h, w), generator=gn) for _ in range(3)]  # cloth posterior, init latents, masked posterior
    return dict(image=image, mask_image=mask, pose_map=pose, warped_cloth=cloth, prompt_embeds=prompt,
                negative_prompt_embeds=neg, noise_cloth=noise[0], noise_latents=noise[1], noise_masked=noise[2])


def fp16_round(t):
    return t.half().float()


# ---------------------------------------------------------------------------------------------------------------
# the pipeline (tryon_pipe.py:494-765); RNG draws are injected (SURVEY.md §8d "Noise")
# ---------------------------------------------------------------------------------------------------------------
def tryon_pipeline(unet_sd, unet_cfg, vae_sd, vae_cfg, emasc_sd, inp, num_inference_steps=50, guidance_scale=7.5, scheduler="ddim",
                   cloth_cond_rate=1.0, no_pose=False, int_layers=(1, 2, 3, 4, 5), trace=None, unet_fn=None):
    """Returns images [B,H,W,3] float in [0,1] (decode_latents layout) and final latents."""
    image, mask_image = inp["image"].clone(), inp["mask_image"].clone()
    pose_map, cloth = inp["pose_map"], inp["warped_cloth"]
    B = image.shape[0]
    do_cfg = guidance_scale > 1.0
    sf = vae_cfg["scaling_factor"]
    # 3. prompt embeddings [negative ; positive]  (:620-628)
    pe = inp["prompt_embeds"]
    if do_cfg:
        pe = torch.cat([inp["negative_prompt_embeds"], pe])
    # 4. prepare_mask_and_masked_image (diffusers): mask binarised IN PLACE, masked = image * (mask < 0.5)
    mask_image[mask_image < 0.5] = 0
    mask_image[mask_image >= 0.5] = 1
    mask = mask_image
    masked_image = image.float() * (mask < 0.5)
    pose = F.interpolate(pose_map, size=(pose_map.shape[2] // 8, pose_map.shape[3] // 8), mode="bilinear")  # :632-634
    if no_pose:
        pose = torch.zeros_like(pose)
    # 4b. cloth latents (RNG draw #1)  (:639-647)
    cloth_latents = None
    if cloth is not None:
        mom, _ = M.vae_encode(vae_sd, vae_cfg, cloth)
        cloth_latents = sf * M.posterior_sample(mom, inp["noise_cloth"])
    # 5. timesteps
    sch = make_scheduler(scheduler)
    sch.set_timesteps(num_inference_steps)
    cloth_conditioning_steps = (1 - cloth_cond_rate) * num_inference_steps  # :654
    # 6. latents (RNG draw #2)
    latents = inp["noise_latents"] * sch.init_noise_sigma
    # 7. mask latents (RNG draw #3) + EMASC (:670-685)
    h, w = image.shape[2] // 8, image.shape[3] // 8
    mask_lat = F.interpolate(mask, size=(h, w))
    mom, feats = M.vae_encode(vae_sd, vae_cfg, masked_image)
    masked_lat = sf * M.posterior_sample(mom, inp["noise_masked"])
    inter = None
    if emasc_sd is not None:
        inter = [feats[i] for i in int_layers]
        inter = M.emasc_forward(emasc_sd, inter)
        inter = M.mask_features(inter, mask_image)
    if do_cfg:
        mask_lat = torch.cat([mask_lat] * 2)
        masked_lat_in = torch.cat([masked_lat] * 2)
        pose = torch.cat([torch.zeros_like(pose), pose])  # :702
        if cloth_latents is not None:
            cloth_latents = torch.cat([torch.zeros_like(cloth_latents), cloth_latents])
    else:
        masked_lat_in = masked_lat
    if unet_fn is None:
        def unet_fn(x, t, e):
            return M.unet_forward(unet_sd, unet_cfg, x, t, e)
    if trace is not None:
        trace.update(cloth_latents=cloth_latents, masked_latents=masked_lat, skips=inter, noise_pred=[], latents=[])
    # 9. loop (:713-747)
    for i, t in enumerate(sch.timesteps):
        x = torch.cat([latents] * 2) if do_cfg else latents
        if cloth_latents is not None and i >= (num_inference_steps - cloth_conditioning_steps):
            cloth_latents = torch.zeros_like(cloth_latents)
        x = sch.scale_model_input(x, t)                     # tryon_pipe.py:722 (identity for DDIM / PNDM)
        parts = [x, mask_lat, masked_lat_in, pose]
        if cloth_latents is not None:
            parts.append(cloth_latents)
        x = torch.cat(parts, dim=1)
        eps = unet_fn(x, t, pe)
        if do_cfg:
            eu, et = eps.chunk(2)
            eps = eu + guidance_scale * (et - eu)
        latents = sch.step(eps, t, latents)
        if trace is not None:
            trace["noise_pred"].append(eps)
            trace["latents"].append(latents)
    # 11. decode_latents (:349-359)
    z = latents / sf
    img = M.vae_decode(vae_sd, vae_cfg, z, list(inter) if inter is not None else None, list(int_layers) if inter is not None else None)
    img = (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()
    return img, latents


def psnr(a, b, peak=None):
    a, b = a.double(), b.double()
    if peak is None:
        peak = float(b.abs().max())
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)
