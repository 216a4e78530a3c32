"""StableDiffusionTryOnePipeline on the native modules — same constructor and __call__ signature as the reference
(src/vto_pipelines/tryon_pipe.py:56-68, 494-520), so src/inference.py:212-220,298-311 can use it unchanged.

Two execution modes with identical semantics:
  * fused (default): steps 4-11 run inside libladi_native (ladi_tryon_run), the denoising step hipGraph-captured;
  * modular (fused=False): the same stages driven from Python through the drop-in module shims, one C-ABI call per module.
"""
import ctypes
import inspect
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from ._lib import TryOnInputs, check, dtype_code, ptr, stream_ptr
from .modules import NativeEMASC, NativeUNet, NativeVAE, mask_features
from .schedulers import DDIMScheduler, LMSDiscreteScheduler, PNDMScheduler


def numpy_to_pil(images):
    from PIL import Image
    if images.ndim == 3:
        images = images[None, ...]
    images = (images * 255).round().astype("uint8")
    return [Image.fromarray(im) for im in images]


class StableDiffusionTryOnePipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker=None, feature_extractor=None,
                 requires_safety_checker=False, emasc=None, emasc_int_layers=None):
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.emasc, self.emasc_int_layers = emasc, emasc_int_layers
        if getattr(scheduler.config, "steps_offset", 1) != 1:
            scheduler.config.steps_offset = 1          # tryon_pipe.py:74-86
        if getattr(scheduler.config, "skip_prk_steps", True) is False:
            scheduler.config.skip_prk_steps = True     # tryon_pipe.py:88-100
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self._tryon = None
        self.last_stage_ms = None
        self.trace_evals = 0          # > 0: the next fused runs record per-evaluation noise_pred / latents into self.last_trace
        self.lanes = None             # sample-group lanes of the fused loop's UNet forward (None: library default, LADI_UNET_LANES or 1)

    def to(self, *a, **k):
        return self

    @property
    def _execution_device(self):
        return torch.device("cuda", torch.cuda.current_device())

    # --- tryon_pipe.py:362-407 -------------------------------------------------------------------------------
    def check_inputs(self, prompt, height, width, callback_steps, negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape, got "
                             f"{prompt_embeds.shape} != {negative_prompt_embeds.shape}.")

    # --- tryon_pipe.py:184-317 (text-encoder branch requires a real text_encoder/tokenizer) -------------------
    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_cfg, negative_prompt=None, prompt_embeds=None,
                       negative_prompt_embeds=None):
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("`prompt` given but the pipeline has no text_encoder/tokenizer; pass `prompt_embeds`.")
            ids = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                                 return_tensors="pt").input_ids
            prompt_embeds = self.text_encoder(ids.to(device))[0]
        prompt_embeds = prompt_embeds.to(device=device)
        B, L, D = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, L, D)
        if do_cfg and negative_prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("classifier-free guidance needs `negative_prompt_embeds` when no text_encoder is attached.")
            toks = [""] * B if negative_prompt is None else ([negative_prompt] * B if isinstance(negative_prompt, str) else negative_prompt)
            ids = self.tokenizer(toks, padding="max_length", max_length=L, truncation=True, return_tensors="pt").input_ids
            negative_prompt_embeds = self.text_encoder(ids.to(device))[0]
        if do_cfg:
            neg = negative_prompt_embeds.to(device=device, dtype=prompt_embeds.dtype)
            neg = neg.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, L, D)
            return prompt_embeds, neg
        return prompt_embeds, None

    @staticmethod
    def _validate_images(image, mask_image):
        # diffusers prepare_mask_and_masked_image tensor branch
        if not isinstance(image, torch.Tensor) or not isinstance(mask_image, torch.Tensor):
            raise TypeError("`image` and `mask_image` must be torch tensors")
        if image.ndim != 4 or mask_image.ndim != 4:
            raise ValueError("`image` and `mask_image` must be 4-D batches")
        if image.shape[-2:] != mask_image.shape[-2:] or image.shape[0] != mask_image.shape[0]:
            raise ValueError("Image and Mask must have the same spatial dimensions and batch size")
        if image.min() < -1 or image.max() > 1:
            raise ValueError("Image should be in [-1, 1] range")
        if mask_image.min() < 0 or mask_image.max() > 1:
            raise ValueError("Mask should be in [0, 1] range")

    def _draw(self, shape, generator, dtype, device):
        """diffusers randn_tensor: one draw of the whole batch, or -- for a LIST of generators (tryon_pipe.py:443-455,463-470) -- one
        [1, ...] draw per sample from that sample's generator"""
        if isinstance(generator, (list, tuple)):
            if len(generator) != shape[0]:
                raise ValueError("You have passed a list of generators of length %d, but requested an effective batch size of %d."
                                 % (len(generator), shape[0]))
            parts = [torch.randn((1,) + tuple(shape[1:]), generator=g, device=g.device, dtype=dtype).to(device) for g in generator]
            return torch.cat(parts, dim=0).to(torch.float32)
        gdev = generator.device if generator is not None else device
        return torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device=device, dtype=torch.float32)

    def decode_latents(self, latents, intermediate_features=None):
        z = latents / self.vae.config.scaling_factor
        if intermediate_features:
            image = self.vae.decode(z, intermediate_features=intermediate_features, int_layers=self.emasc_int_layers).sample
        else:
            image = self.vae.decode(z).sample
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.cpu().permute(0, 2, 3, 1).float().numpy()

    @torch.no_grad()
    def __call__(self, image, mask_image, pose_map, warped_cloth, prompt=None, height=None, width=None, num_inference_steps=50,
                 guidance_scale=7.5, negative_prompt=None, num_images_per_prompt=1, eta=0.0, prompt_embeds=None,
                 negative_prompt_embeds=None, generator=None, latents=None, output_type="pil", return_dict=True, callback=None,
                 callback_steps=1, cloth_cond_rate=1.0, no_pose=False, cloth_input_type="warped", fused=True, noise=None,
                 use_graph=True):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if image is None:
            raise ValueError("`image` input cannot be undefined.")
        if mask_image is None:
            raise ValueError("`mask_image` input cannot be undefined.")
        if cloth_input_type not in ("warped", "none"):
            raise ValueError(f"Invalid cloth_input_type {cloth_input_type}")
        # num_images_per_prompt = k repeats every prompt k times (tryon_pipe.py:259-260,309-310) and sizes the latents for B * k samples
        # (:659); the reference concatenates image-derived tensors (pose map, warped cloth) unrepeated (:724-729), so the only inputs it
        # can run are image batches that already hold B * k samples -- the same rule applies here (shape checks below)
        if not isinstance(num_images_per_prompt, int) or num_images_per_prompt < 1:
            raise ValueError("num_images_per_prompt must be a positive integer")
        device = self._execution_device
        do_cfg = guidance_scale > 1.0
        pe, neg = self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        B = pe.shape[0]
        self._validate_images(image, mask_image)
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        # RNG draws in pipeline order (cloth posterior, initial latents, masked-image posterior)
        if noise is None:
            n_cloth = self._draw((B, 4, h, w), generator, pe.dtype, device) if cloth_input_type == "warped" else None
            n_lat = latents.to(device=device, dtype=torch.float32) if latents is not None else self._draw((B, 4, h, w), generator, pe.dtype, device)
            n_mask = self._draw((B, 4, h, w), generator, pe.dtype, device)
        else:
            n_cloth, n_lat, n_mask = [t.to(device=device, dtype=torch.float32).contiguous() if t is not None else None for t in noise]
        native = isinstance(self.unet, NativeUNet) and isinstance(self.vae, NativeVAE) and (self.emasc is None or isinstance(self.emasc, NativeEMASC))
        can_fuse = (fused and native and callback is None and isinstance(self.scheduler, (DDIMScheduler, PNDMScheduler, LMSDiscreteScheduler)) and eta == 0.0
                    and (not self.emasc or list(self.emasc_int_layers or []) == [1, 2, 3, 4, 5]))
        if can_fuse:
            images = self._run_fused(image, mask_image, pose_map, warped_cloth if cloth_input_type == "warped" else None, pe, neg,
                                     n_cloth, n_lat, n_mask, height, width, num_inference_steps, guidance_scale, cloth_cond_rate,
                                     no_pose, use_graph)
            # prepare_mask_and_masked_image binarises the caller's mask in place; keep that side effect
            mask_image[mask_image < 0.5] = 0
            mask_image[mask_image >= 0.5] = 1
        else:
            images = self._run_modular(image, mask_image, pose_map, warped_cloth if cloth_input_type == "warped" else None, pe, neg,
                                       n_cloth, n_lat, n_mask, height, width, num_inference_steps, guidance_scale, cloth_cond_rate,
                                       no_pose, eta, generator, callback, callback_steps)
        if output_type == "pil":
            images = numpy_to_pil(images)
        if not return_dict:
            return (images, None)
        return SimpleNamespace(images=images, nsfw_content_detected=None)

    # -------------------------------------------------------------------------------------------------------
    def _run_fused(self, image, mask_image, pose_map, cloth, pe, neg, n_cloth, n_lat, n_mask, H, W, steps, guidance, ccr, no_pose,
                   use_graph, return_device=False, out_uint8=False, lanes=None):
        """return_device: hand back the device tensor (no host copy); out_uint8: the batch as uint8 [B,H,W,3] = numpy_to_pil's
        (images * 255).round() computed by the decode epilogue (ladi_tryon_run_u8); lanes: sample-group lanes of the UNet forward"""
        lib = _lib.load()
        if self._tryon is None:
            self._tryon = lib.ladi_tryon_create(self.unet.h, self.vae.h, self.emasc.h if self.emasc else None)
            if not self._tryon:
                raise _lib.NativeError("ladi_tryon_create failed: " + _lib.last_error())
        dev = self._execution_device
        dt = torch.float16 if image.dtype == torch.float16 else torch.float32
        B = pe.shape[0]
        # the C ABI takes raw device pointers: every shape the kernels assume is checked here (the reference fails in torch.cat at
        # tryon_pipe.py:724-729 for the same mistakes)
        h8, w8 = H // 8, W // 8
        for name, t, shp in (("image", image, (B, 3, H, W)), ("mask_image", mask_image, (B, 1, H, W)),
                             ("warped_cloth", cloth, (B, 3, H, W)), ("cloth posterior noise", n_cloth, (B, 4, h8, w8)),
                             ("latents", n_lat, (B, 4, h8, w8)), ("masked-image posterior noise", n_mask, (B, 4, h8, w8))):
            if t is not None and tuple(t.shape) != shp:
                raise ValueError("%s has shape %s, expected %s (batch from prompt_embeds, size from height/width)" % (name, tuple(t.shape), shp))
        if pose_map.dim() != 4 or pose_map.shape[0] != B or tuple(pose_map.shape[2:]) != (H, W):
            raise ValueError("pose_map has shape %s, expected (%d, P, %d, %d)" % (tuple(pose_map.shape), B, H, W))
        if neg is not None and tuple(neg.shape) != tuple(pe.shape):
            raise ValueError("negative_prompt_embeds shape %s != prompt_embeds shape %s" % (tuple(neg.shape), tuple(pe.shape)))
        keep = [t.to(device=dev, dtype=dt).contiguous() if t is not None else None for t in (image, mask_image, pose_map, cloth)]
        pe16 = pe.to(device=dev, dtype=torch.float16).contiguous()
        neg16 = neg.to(device=dev, dtype=torch.float16).contiguous() if neg is not None else None
        n_cloth, n_lat, n_mask = [t.to(device=dev, dtype=torch.float32).contiguous() if t is not None else None for t in (n_cloth, n_lat, n_mask)]
        inp = TryOnInputs()
        inp.batch, inp.height, inp.width, inp.in_dtype = B, H, W, dtype_code(keep[0])
        inp.image_dev, inp.mask_image_dev, inp.pose_map_dev = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
        inp.warped_cloth_dev = keep[3].data_ptr() if keep[3] is not None else None
        inp.pose_channels = keep[2].shape[1]
        inp.prompt_embeds_dev = pe16.data_ptr()
        inp.negative_prompt_embeds_dev = neg16.data_ptr() if neg16 is not None else None
        inp.L = pe16.shape[1]
        inp.noise_cloth_dev = n_cloth.data_ptr() if n_cloth is not None else None
        inp.noise_latents_dev, inp.noise_masked_dev = n_lat.data_ptr(), n_mask.data_ptr()
        inp.num_inference_steps, inp.guidance_scale = int(steps), float(guidance)
        inp.scheduler = self.scheduler.kind
        # tryon_pipe.py:654,718 in the reference's own float64 arithmetic: first evaluation index i with i >= steps - (1 - rate) * steps
        ccs = (1 - ccr) * steps
        inp.cloth_zero_from_eval = min(max(0, math.ceil(steps - ccs)), 1 << 30)
        inp.no_pose, inp.use_graph = int(bool(no_pose)), int(bool(use_graph))
        ac = self.scheduler.alphas_cumprod.to("cpu", torch.float32).contiguous()
        inp.alphas_cumprod_host = ac.data_ptr()
        images = torch.empty((B, H, W, 3), dtype=torch.uint8 if out_uint8 else torch.float32, device=dev)
        self.last_latents = torch.empty((B, 4, H // 8, W // 8), dtype=torch.float32, device=dev)
        tr = None
        if self.trace_evals > 0:
            tr = torch.zeros((2, self.trace_evals, B, h8 * w8, 4), dtype=torch.float32, device=dev)
            check(lib.ladi_tryon_set_trace(self._tryon, ptr(tr[0]), ptr(tr[1]), self.trace_evals), "ladi_tryon_set_trace")
        else:
            check(lib.ladi_tryon_set_trace(self._tryon, None, None, 0), "ladi_tryon_set_trace")
        # the fused loop rewrites the UNet's cross-attention K/V cache behind the shim's back
        self.unet._ctx_key = None
        if lanes is not None or self.lanes is not None:
            check(lib.ladi_tryon_set_lanes(self._tryon, int(lanes if lanes is not None else self.lanes)), "ladi_tryon_set_lanes")
        run = lib.ladi_tryon_run_u8 if out_uint8 else lib.ladi_tryon_run
        check(run(self._tryon, ctypes.byref(inp), ptr(images), ptr(self.last_latents), stream_ptr()), "ladi_tryon_run")
        if not return_device:
            # results go to the host: this is the synchronisation point anyway, so the decode's fp16-range guard is asked now (a run decodes
            # once and queues its flag; no host round trip inside the run).  An overflow raised the automatic range shift: run the batch again.
            # With return_device the flag is examined by the next call instead (or by check_overflow()), which fails loudly rather than hand out
            # a bad batch silently.
            for _ in range(2):
                po = lib.ladi_tryon_poll_overflow(self._tryon)
                if po == 0:
                    break
                if po < 0:
                    raise _lib.NativeError("ladi_tryon_poll_overflow: " + _lib.last_error())
                check(run(self._tryon, ctypes.byref(inp), ptr(images), ptr(self.last_latents), stream_ptr()), "ladi_tryon_run (re-run at a larger range shift)")
            else:
                if lib.ladi_tryon_poll_overflow(self._tryon) != 0:
                    raise _lib.NativeError("VAE decode: activations exceed the fp16 range even at range shift 8")
        if tr is not None:   # [evals, B, 4, h, w] like the reference's noise_pred / latents (tryon_pipe.py:732-740)
            nchw = tr.view(2, self.trace_evals, B, h8, w8, 4).permute(0, 1, 2, 5, 3, 4)
            self.last_trace = dict(noise_pred=nchw[0].contiguous(), latents=nchw[1].contiguous())
        if return_device:
            return images
        out = images.cpu().numpy()  # the reference's only sync point (tryon_pipe.py:358)
        ms = (ctypes.c_float * 3)()
        if lib.ladi_tryon_stage_ms(self._tryon, ms) == 0:
            self.last_stage_ms = list(ms)
        return out

    def check_overflow(self):
        """True if the last fused run's decode left the fp16 range (its images are invalid; the automatic range shift has been raised, so running
        the batch again gives the result).  For callers that keep results on the device (return_device=True): waits for the run to finish."""
        if not self._tryon:
            return False
        po = _lib.load().ladi_tryon_poll_overflow(self._tryon)
        if po < 0:
            raise _lib.NativeError("ladi_tryon_poll_overflow: " + _lib.last_error())
        return po == 1

    def lib_lanes(self):
        """sample-group lanes the last fused run used"""
        return _lib.load().ladi_tryon_lanes(self._tryon) if self._tryon else None

    def __del__(self):
        if getattr(self, "_tryon", None):
            _lib.load().ladi_tryon_destroy(self._tryon)
            self._tryon = None

    # -------------------------------------------------------------------------------------------------------
    def _run_modular(self, image, mask_image, pose_map, cloth, pe, neg, n_cloth, n_lat, n_mask, H, W, steps, guidance, ccr, no_pose, eta,
                     generator, callback, callback_steps):
        F = torch.nn.functional
        dev = self._execution_device
        do_cfg = neg is not None
        sf = self.vae.config.scaling_factor
        ehs = torch.cat([neg, pe]) if do_cfg else pe
        mask_image[mask_image < 0.5] = 0
        mask_image[mask_image >= 0.5] = 1
        mask = mask_image.to(dev)
        masked_image = image.to(dev).float() * (mask < 0.5)
        pose = F.interpolate(pose_map.to(dev).float(), size=(pose_map.shape[2] // 8, pose_map.shape[3] // 8), mode="bilinear")
        if no_pose:
            pose = torch.zeros_like(pose)
        cloth_lat = None
        if cloth is not None:
            cloth_lat = sf * self.vae.encode(cloth.to(dev))[0].latent_dist.sample(noise=n_cloth).float()
        self.scheduler.set_timesteps(steps, device=dev)
        timesteps = self.scheduler.timesteps
        ccs = (1 - ccr) * steps
        latents = n_lat * self.scheduler.init_noise_sigma
        mask_lat = F.interpolate(mask.float(), size=(H // 8, W // 8))
        enc, feats = self.vae.encode(masked_image)
        masked_lat = sf * enc.latent_dist.sample(noise=n_mask).float()
        inter = None
        if self.emasc:
            inter = [feats[i] for i in self.emasc_int_layers]
            inter = self.emasc(inter)
            inter = mask_features(inter, mask_image.to(dev))
        if do_cfg:
            mask_lat = torch.cat([mask_lat] * 2)
            masked_lat = torch.cat([masked_lat] * 2)
            pose = torch.cat([torch.zeros_like(pose), pose])
            if cloth_lat is not None:
                cloth_lat = torch.cat([torch.zeros_like(cloth_lat), cloth_lat])
        extra = {}
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        if "eta" in params:
            extra["eta"] = eta
        if "generator" in params:
            extra["generator"] = generator
        for i, t in enumerate(timesteps):
            x = torch.cat([latents] * 2) if do_cfg else latents
            if cloth_lat is not None and i >= (steps - ccs):
                cloth_lat = torch.zeros_like(cloth_lat)
            x = self.scheduler.scale_model_input(x, t)
            parts = [x, mask_lat, masked_lat, pose] + ([cloth_lat] if cloth_lat is not None else [])
            x = torch.cat([p.float() for p in parts], dim=1)
            eps = self.unet(x, t, encoder_hidden_states=ehs).sample.float()
            if do_cfg:
                eu, et = eps.chunk(2)
                eps = eu + guidance * (et - eu)
            latents = self.scheduler.step(eps, t, latents, **extra).prev_sample
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        self.last_latents = latents
        return self.decode_latents(latents, inter)
