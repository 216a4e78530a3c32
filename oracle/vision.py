"""CPU fp32 oracle of the CLIP ViT-H/14 vision encoder.  TEST INFRASTRUCTURE ONLY: imported by tests/ and
tools/ baselines, never by the product.

The reference owns no code here: src/inference.py:269-273 calls `vision_encoder(pixel_values).last_hidden_state` on a
transformers `CLIPVisionModelWithProjection` (third-party: transformers==4.27.3, models/clip/modeling_clip.py, not vendored) and hands
the result to the inversion adapter (:276).  Restated from that module: CLIPVisionEmbeddings (bias-free patch conv, class token first,
learned positions), `pre_layrnorm` (sic), pre-LN encoder layers (q scaled by head_dim**-0.5, no mask, gelu MLP);
`last_hidden_state` is the encoder output WITHOUT post_layernorm, `pooler_output` = post_layernorm(last_hidden_state[:, 0]).

Pinned (tests/test_cpu.py::test_vision_oracle_matches_transformers_golden) against tests/golden/clip_vision_tiny.safetensors, produced by
oracle/make_golden.py from the installed transformers CLIPVisionModel (same arithmetic, 5.x packaging).
"""
import torch
import torch.nn.functional as F


def _get(sd, key):
    if key in sd:
        return sd[key]
    alt = key[len("vision_model."):] if key.startswith("vision_model.") else "vision_model." + key
    return sd[alt]


def _linear(sd, p, x):
    return F.linear(x, _get(sd, p + ".weight"), _get(sd, p + ".bias"))


def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), _get(sd, p + ".weight"), _get(sd, p + ".bias"), eps)


def clip_vision_forward(sd, cfg, pixel_values):
    """pixel_values [B,3,S,S] -> (last_hidden_state [B, 1+(S/ps)^2, H], pooler_output [B, H])"""
    eps, heads, ps = cfg["layer_norm_eps"], cfg["heads"], cfg["patch_size"]
    B = pixel_values.shape[0]
    patches = F.conv2d(pixel_values, _get(sd, "vision_model.embeddings.patch_embedding.weight"), None, stride=ps)   # [B,H,g,g]
    patches = patches.flatten(2).transpose(1, 2)                                                                   # [B,g*g,H]
    cls = _get(sd, "vision_model.embeddings.class_embedding").expand(B, 1, -1)
    x = torch.cat([cls, patches], dim=1) + _get(sd, "vision_model.embeddings.position_embedding.weight").unsqueeze(0)
    x = _ln(sd, "vision_model.pre_layrnorm", x, eps)
    T, H = x.shape[1], x.shape[2]
    d = H // heads
    for i in range(cfg["layers"]):
        p = "vision_model.encoder.layers.%d" % i
        a = _ln(sd, p + ".layer_norm1", x, eps)
        q = (_linear(sd, p + ".self_attn.q_proj", a) * d ** -0.5).view(B, T, heads, d).transpose(1, 2)
        k = _linear(sd, p + ".self_attn.k_proj", a).view(B, T, heads, d).transpose(1, 2)
        v = _linear(sd, p + ".self_attn.v_proj", a).view(B, T, heads, d).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
        x = x + _linear(sd, p + ".self_attn.out_proj", (w @ v).transpose(1, 2).reshape(B, T, H))
        a = _ln(sd, p + ".layer_norm2", x, eps)
        x = x + _linear(sd, p + ".mlp.fc2", F.gelu(_linear(sd, p + ".mlp.fc1", a)))
    pooled = _ln(sd, "vision_model.post_layernorm", x[:, 0], eps)
    return x, pooled
