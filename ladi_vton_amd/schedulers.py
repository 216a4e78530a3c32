"""Host-side mirrors of the three schedulers the reference pipeline accepts (tryon_pipe.py:62: diffusers 0.14.0 DDIMScheduler — what
src/inference.py:123-124 instantiates —, PNDMScheduler with skip_prk_steps, which the SD2-inpainting scheduler_config.json
describes, and LMSDiscreteScheduler).  They expose the attributes tryon_pipe.py touches
(:74,88,331-346,424,650-651,711,722,740).

The fused native loop (ladi_tryon_run) does not call .step(): it consumes the same tables on the device.  .step() here serves
the module-by-module drop-in path and operates on small [B,4,h,w] tensors.
"""
from types import SimpleNamespace

import torch

from . import _lib

DDIM, PNDM, LMS = 0, 1, 2


def _alphas_cumprod():
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _SchedulerBase:
    order = 1
    init_noise_sigma = 1.0
    kind = None

    def __init__(self):
        self.alphas_cumprod = _alphas_cumprod()
        self.final_alpha_cumprod = self.alphas_cumprod[0]  # set_alpha_to_one = False
        self.config = SimpleNamespace(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                      steps_offset=1, skip_prk_steps=True, set_alpha_to_one=False, clip_sample=False,
                                      prediction_type="epsilon")
        self.timesteps = None
        self.num_inference_steps = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _native_timesteps(self, n):
        import ctypes
        buf = (ctypes.c_int * (n + 2))()
        cnt = _lib.load().ladi_sched_timesteps(self.kind, n, buf, n + 2)
        if cnt < 0:
            raise _lib.NativeError("ladi_sched_timesteps: " + _lib.last_error())
        return list(buf[:cnt])

    def _a(self, t):
        return self.alphas_cumprod[t] if t >= 0 else self.final_alpha_cumprod


class DDIMScheduler(_SchedulerBase):
    kind = DDIM

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.ratio = 1000 // num_inference_steps
        ts = self._native_timesteps(num_inference_steps)
        self.timesteps = torch.tensor(ts, dtype=torch.int64, device=device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None, **kw):
        """DDIM eq. (12) as diffusers 0.14 writes it: sigma_t = eta * sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)); the stochastic
        term (eta > 0) draws ONE batch-shaped normal tensor from `generator` per step (randn_tensor), or uses `variance_noise`"""
        t = int(timestep)
        a_t, a_p = float(self._a(t)), float(self._a(t - self.ratio))
        x, e = sample.float(), model_output.float()
        x0 = (x - (1 - a_t) ** 0.5 * e) / a_t ** 0.5
        var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
        std = float(eta) * max(var, 0.0) ** 0.5
        prev = a_p ** 0.5 * x0 + max(1 - a_p - std * std, 0.0) ** 0.5 * e
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output=True is not implemented (the try-on pipeline never sets it: tryon_pipe.py:740)")
        if eta > 0:
            if variance_noise is None:
                # diffusers 0.14 randn_tensor(model_output.shape, generator, device, dtype=model_output.dtype): drawn on the generator's
                # device; a LIST of generators draws one [1, ...] tensor per sample from that sample's generator
                shape, dt = tuple(model_output.shape), model_output.dtype
                if isinstance(generator, (list, tuple)):
                    if len(generator) != shape[0]:
                        raise ValueError("You have passed a list of generators of length %d, but requested an effective batch size of %d."
                                         % (len(generator), shape[0]))
                    variance_noise = torch.cat([torch.randn((1,) + shape[1:], generator=g_, device=g_.device, dtype=dt).to(sample.device)
                                                for g_ in generator], dim=0)
                else:
                    gdev = generator.device if generator is not None else sample.device
                    variance_noise = torch.randn(shape, generator=generator, device=gdev, dtype=dt).to(sample.device)
            prev = prev + std * variance_noise.float()
        return SimpleNamespace(prev_sample=prev.to(sample.dtype))


class PNDMScheduler(_SchedulerBase):
    kind = PNDM

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.ratio = 1000 // num_inference_steps
        ts = self._native_timesteps(num_inference_steps)
        self.timesteps = torch.tensor(ts, dtype=torch.int64, device=device)
        self.ets, self.counter, self.cur_sample = [], 0, None

    def step(self, model_output, timestep, sample, **kw):
        t = int(timestep)
        tp = t - self.ratio
        e_now, x = model_output.float(), sample.float()
        if self.counter != 1:
            self.ets = self.ets[-3:] + [e_now]
        else:
            tp, t = t, t + self.ratio
        n = len(self.ets)
        if n == 1 and self.counter == 0:
            e, self.cur_sample = e_now, x
        elif n == 1 and self.counter == 1:
            e, x, self.cur_sample = (e_now + self.ets[-1]) / 2, self.cur_sample, None
        elif n == 2:
            e = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif n == 3:
            e = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            e = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        a_t, a_p = float(self._a(t)), float(self._a(tp))
        denom = a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5
        prev = (a_p / a_t) ** 0.5 * x - (a_p - a_t) * e / denom
        self.counter += 1
        return SimpleNamespace(prev_sample=prev.to(sample.dtype))


class LMSDiscreteScheduler(_SchedulerBase):
    """diffusers 0.14 LMSDiscreteScheduler (order-4 linear multistep in the sigma parameterisation, epsilon prediction).  The
    fractional timesteps, the sigmas and the multistep weights all come from the native table builder (ladi_sched_lms), i.e. they
    are the very numbers the fused device loop uses."""
    kind = LMS

    def __init__(self):
        super().__init__()
        self.sigmas = None
        self.init_noise_sigma = float(((1 - self.alphas_cumprod) / self.alphas_cumprod).sqrt().max())   # as diffusers before set_timesteps

    def set_timesteps(self, num_inference_steps, device=None):
        import ctypes
        n = int(num_inference_steps)
        ts = (ctypes.c_double * n)()
        sg = (ctypes.c_float * (n + 1))()
        cf = (ctypes.c_float * (4 * n))()
        ac = self.alphas_cumprod.to("cpu", torch.float32).contiguous()      # the same table the fused loop receives
        if _lib.load().ladi_sched_lms(n, ctypes.c_void_p(ac.data_ptr()), ts, sg, cf) < 0:
            raise _lib.NativeError("ladi_sched_lms: " + _lib.last_error())
        self.num_inference_steps = n
        self.timesteps = torch.tensor(list(ts), dtype=torch.float64, device=device)
        self.sigmas = torch.tensor(list(sg), dtype=torch.float32)
        self._coeffs = [list(cf[4 * i:4 * i + 4]) for i in range(n)]
        self._ts = list(ts)
        self.init_noise_sigma = float(self.sigmas.max())
        self.derivatives = []

    def _index(self, timestep):
        t = float(timestep)
        return min(range(len(self._ts)), key=lambda i: abs(self._ts[i] - t))

    def scale_model_input(self, sample, timestep=None):
        sigma = float(self.sigmas[self._index(timestep)])
        return sample / ((sigma * sigma + 1.0) ** 0.5)

    def step(self, model_output, timestep, sample, order=4, **kw):
        if order != 4:
            raise NotImplementedError("LMSDiscreteScheduler.step: only the default order = 4 is supported")
        i = self._index(timestep)
        sigma = float(self.sigmas[i])
        x, e = sample.float(), model_output.float()
        x0 = x - sigma * e
        self.derivatives = (self.derivatives + [(x - x0) / sigma])[-4:]
        prev = x
        for c, d in zip(self._coeffs[i][:min(i + 1, 4)], reversed(self.derivatives)):
            prev = prev + c * d
        return SimpleNamespace(prev_sample=prev.to(sample.dtype))
