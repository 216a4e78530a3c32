"""Multi-GPU: batch sharding of independent try-on pairs, one process per GPU, with ONE collective — an all-gather of the
decoded images (RCCL over xGMI; backend "nccl" is RCCL on ROCm).  SURVEY.md §8e: every sample is independent through the
whole path (GroupNorm / LayerNorm / attention are per-sample, CFG pairs stay on one GPU), weights are replicated.

The reference has no working multi-GPU inference path (src/inference.py:92-94 reads an undefined args.local_rank).
Noise is drawn for the GLOBAL batch and sliced per rank so results do not depend on the world size.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(batch, rank, world):
    """contiguous [lo, hi) of `batch` items owned by `rank`; ragged batches give the first (batch % world) ranks one extra"""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_inputs(inputs, rank, world):
    """slice every [B, ...] tensor of the pipeline-input dict along dim 0"""
    B = inputs["image"].shape[0]
    lo, hi = shard_bounds(B, rank, world)
    return {k: (v[lo:hi] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == B else v) for k, v in inputs.items()}, (lo, hi)


def to_uint8(images):
    """numpy_to_pil rounding: (x * 255).round() -> uint8, on the device.  The fused pipeline already hands over uint8
    (ladi_tryon_run_u8: the rounding happens in the decode epilogue); only float batches of other callers are converted here."""
    if images.dtype == torch.uint8:
        return images
    return (images * 255.0).round().clamp(0, 255).to(torch.uint8)


def all_gather_images(local_u8, batch, group=None):
    """all-gather of per-rank [b_r, H, W, 3] uint8 shards into [batch, H, W, 3] on every rank (padded to the largest shard)"""
    if not dist.is_available() or not dist.is_initialized():
        return local_u8
    if dist.get_world_size(group) == 1 and os.environ.get("LADI_FORCE_COLLECTIVE") != "1":   # the switch lets a 1-GPU box exercise RCCL
        return local_u8
    world = dist.get_world_size(group)
    sizes = [shard_bounds(batch, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = local_u8
    if local_u8.shape[0] < mx:
        pad = torch.zeros((mx,) + tuple(local_u8.shape[1:]), dtype=local_u8.dtype, device=local_u8.device)
        pad[:local_u8.shape[0]] = local_u8
    out = torch.empty((world * mx,) + tuple(local_u8.shape[1:]), dtype=local_u8.dtype, device=local_u8.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = [out[r * mx:r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def run_sharded(run_local, inputs, group=None, batch=None):
    """run_local(local_inputs) -> [b_r, H, W, 3] images on this rank's device (uint8, or float in [0,1]); returns the gathered uint8 batch.
    `inputs` is the GLOBAL batch: either a dict of [B, ...] tensors (sliced here), or a callable rows(lo, hi) -> dict that materialises
    rows [lo, hi) of the global batch (same values as slicing it; `batch` = B) so that a rank never builds the other ranks' samples."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if callable(inputs):
        if batch is None:
            raise ValueError("`batch` (global batch size) is required with a row materialiser")
        lo, hi = shard_bounds(batch, rank, world)
        local = inputs(lo, hi)
    else:
        batch = inputs["image"].shape[0]
        local, _ = shard_inputs(inputs, rank, world)
    imgs = run_local(local)
    return all_gather_images(to_uint8(imgs), batch, group)
