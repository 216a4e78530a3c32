"""Plain-torch CPU fp32 restatement of the third-party arithmetic on the LaDI-VTON hot path:
diffusers 0.14.0 UNet2DConditionModel / VAE blocks, the EMASC-aware VAE wrappers of the reference
(src/models/vae.py:99-119,183-212; src/models/AutoencoderKL.py:145-188), EMASC (src/models/emasc.py:37-40),
mask_features (src/utils/data_utils.py:4-16) and the inversion adapter (src/models/inversion_adapter.py:22-28).
Functional style over a diffusers-format state_dict.  Test infrastructure only (see oracle/__init__.py)."""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------------------------
def conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def group_norm(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def layer_norm(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def timestep_embedding(t, dim):
    """diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0) — SURVEY.md A.1 step 1"""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def resnet(sd, p, x, temb, groups, eps):
    """diffusers ResnetBlock2D — SURVEY.md A.2"""
    h = conv(sd, p + ".conv1", F.silu(group_norm(sd, p + ".norm1", x, groups, eps)))
    if temb is not None:
        h = h + linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = conv(sd, p + ".conv2", F.silu(group_norm(sd, p + ".norm2", h, groups, eps)))
    if (p + ".conv_shortcut.weight") in sd:
        x = conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def attention(q, k, v, heads):
    n, T, C = q.shape
    d = C // heads
    q = q.view(n, T, heads, d).transpose(1, 2)
    k = k.view(n, k.shape[1], heads, d).transpose(1, 2)
    v = v.view(n, v.shape[1], heads, d).transpose(1, 2)
    if n * heads * T * k.shape[2] > (1 << 28):      # 1024x768 self-attention: scores would take several GB; softmax is per query row, so
        o = torch.cat([torch.matmul(torch.softmax(torch.matmul(q[:, :, i:i + 1024], k.transpose(-1, -2)) * (d ** -0.5), dim=-1), v)
                       for i in range(0, T, 1024)], dim=2)                   # chunking the queries is the same arithmetic row by row
    else:
        s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), v)
    return o.transpose(1, 2).reshape(n, T, C)


def transformer2d(sd, p, x, ehs, heads, groups):
    """diffusers Transformer2DModel(use_linear_projection) + BasicTransformerBlock — SURVEY.md A.3"""
    n, C, h, w = x.shape
    res = x
    t = group_norm(sd, p + ".norm", x, groups, 1e-6).permute(0, 2, 3, 1).reshape(n, h * w, C)
    t = linear(sd, p + ".proj_in", t)
    b = p + ".transformer_blocks.0"
    a = layer_norm(sd, b + ".norm1", t)
    t = linear(sd, b + ".attn1.to_out.0", attention(linear(sd, b + ".attn1.to_q", a), linear(sd, b + ".attn1.to_k", a),
                                                    linear(sd, b + ".attn1.to_v", a), heads)) + t
    a = layer_norm(sd, b + ".norm2", t)
    t = linear(sd, b + ".attn2.to_out.0", attention(linear(sd, b + ".attn2.to_q", a), linear(sd, b + ".attn2.to_k", ehs),
                                                    linear(sd, b + ".attn2.to_v", ehs), heads)) + t
    a = layer_norm(sd, b + ".norm3", t)
    u, g = linear(sd, b + ".ff.net.0.proj", a).chunk(2, dim=-1)
    t = linear(sd, b + ".ff.net.2", u * F.gelu(g)) + t
    t = linear(sd, p + ".proj_out", t)
    return t.reshape(n, h, w, C).permute(0, 3, 1, 2) + res


# ---------------------------------------------------------------------------------------------------------------
# UNet2DConditionModel.forward — SURVEY.md A.1 (called at tryon_pipe.py:732)
# ---------------------------------------------------------------------------------------------------------------
def unet_forward(sd, cfg, sample, timestep, ehs, return_probe=False):
    boc = cfg["block_out_channels"]
    heads = cfg["num_heads"]
    L = cfg["layers_per_block"]
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    n = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(n)
    temb = timestep_embedding(t, boc[0])
    temb = linear(sd, "time_embedding.linear_2", F.silu(linear(sd, "time_embedding.linear_1", temb)))
    x = conv(sd, "conv_in", sample)
    skips = [x]
    for i in range(4):
        for j in range(L):
            x = resnet(sd, "down_blocks.%d.resnets.%d" % (i, j), x, temb, G, eps)
            if i < 3:
                x = transformer2d(sd, "down_blocks.%d.attentions.%d" % (i, j), x, ehs, heads[i], G)
            skips.append(x)
        if i < 3:
            x = conv(sd, "down_blocks.%d.downsamplers.0.conv" % i, x, stride=2, padding=1)
            skips.append(x)
    x = resnet(sd, "mid_block.resnets.0", x, temb, G, eps)
    x = transformer2d(sd, "mid_block.attentions.0", x, ehs, heads[3], G)
    x = resnet(sd, "mid_block.resnets.1", x, temb, G, eps)
    probe = x
    for i in range(4):
        for j in range(L + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet(sd, "up_blocks.%d.resnets.%d" % (i, j), x, temb, G, eps)
            if i > 0:
                x = transformer2d(sd, "up_blocks.%d.attentions.%d" % (i, j), x, ehs, heads[3 - i], G)
        if i < 3:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv(sd, "up_blocks.%d.upsamplers.0.conv" % i, x)
    x = conv(sd, "conv_out", F.silu(group_norm(sd, "conv_norm_out", x, G, eps)))
    return (x, probe) if return_probe else x


# ---------------------------------------------------------------------------------------------------------------
# VAE (diffusers blocks) with the reference's EMASC wiring
# ---------------------------------------------------------------------------------------------------------------
def vae_attention(sd, p, x, groups):
    """diffusers 0.14 AttentionBlock (single head): softmax in fp32"""
    n, C, h, w = x.shape
    t = group_norm(sd, p + ".group_norm", x, groups, 1e-6).reshape(n, C, h * w).transpose(1, 2)
    q, k, v = linear(sd, p + ".query", t), linear(sd, p + ".key", t), linear(sd, p + ".value", t)
    s = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(C), dim=-1)
    o = linear(sd, p + ".proj_attn", torch.matmul(s, v))
    return o.transpose(1, 2).reshape(n, C, h, w) + x


def vae_encode(sd, cfg, x):
    """Encoder.forward (src/models/vae.py:99-119) + quant_conv (AutoencoderKL.py:150-151).
    Returns (moments, [x, conv_in(x), in(down0), in(down1), in(down2), in(down3)])."""
    L, G = cfg["layers_per_block"], cfg["norm_num_groups"]
    feats = [x]
    s = conv(sd, "encoder.conv_in", x)
    feats.append(s)
    for i in range(4):
        feats.append(s)
        for j in range(L):
            s = resnet(sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), s, None, G, 1e-6)
        if i < 3:
            s = conv(sd, "encoder.down_blocks.%d.downsamplers.0.conv" % i, F.pad(s, (0, 1, 0, 1)), stride=2, padding=0)
    s = resnet(sd, "encoder.mid_block.resnets.0", s, None, G, 1e-6)
    s = vae_attention(sd, "encoder.mid_block.attentions.0", s, G)
    s = resnet(sd, "encoder.mid_block.resnets.1", s, None, G, 1e-6)
    s = conv(sd, "encoder.conv_out", F.silu(group_norm(sd, "encoder.conv_norm_out", s, G, 1e-6)))
    return conv(sd, "quant_conv", s, padding=0), feats


def posterior_sample(moments, noise):
    """DiagonalGaussianDistribution (src/models/vae.py:329-348) with the generator draw supplied explicitly"""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return mean + std * noise


def vae_decode(sd, cfg, z, feats=None, int_layers=None):
    """AutoencoderKL._decode (:159-172) + Decoder.forward (vae.py:183-212).  `feats` is REVERSED IN PLACE like the reference."""
    L, G = cfg["layers_per_block"], cfg["norm_num_groups"]
    s = conv(sd, "post_quant_conv", z, padding=0)
    s = conv(sd, "decoder.conv_in", s)
    s = resnet(sd, "decoder.mid_block.resnets.0", s, None, G, 1e-6)
    s = vae_attention(sd, "decoder.mid_block.attentions.0", s, G)
    s = resnet(sd, "decoder.mid_block.resnets.1", s, None, G, 1e-6)
    if feats:
        feats.reverse()

    def up(i, s):
        for j in range(L + 1):
            s = resnet(sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), s, None, G, 1e-6)
        if i < 3:
            s = F.interpolate(s, scale_factor=2.0, mode="nearest")
            s = conv(sd, "decoder.up_blocks.%d.upsamplers.0.conv" % i, s)
        return s

    if feats:
        for i, f in zip(range(4), feats):
            s = s + f
            s = up(i, s)
    else:
        for i in range(4):
            s = up(i, s)
    s = F.silu(group_norm(sd, "decoder.conv_norm_out", s, G, 1e-6))
    if int_layers and 1 in int_layers:
        s = s + feats[len(int_layers) - 1 - int_layers.index(1)]
    s = conv(sd, "decoder.conv_out", s)
    if int_layers and 0 in int_layers:
        s = s + feats[len(int_layers) - 1 - int_layers.index(0)]
    return s


def emasc_forward(sd, feats):
    """EMASC.forward, type='nonlinear' (src/models/emasc.py:26-40)"""
    out = []
    for i, f in enumerate(feats):
        h = F.silu(conv(sd, "conv.%d.0" % i, f))
        out.append(conv(sd, "conv.%d.2" % i, h))
    return out


def mask_features(features, mask):
    """src/utils/data_utils.py:4-16 (progressive nearest resize of the already-resized mask)"""
    out = []
    for f in features:
        mask = F.interpolate(mask, size=f.shape[-2:])
        out.append(f * (1 - mask))
    return out


# ---------------------------------------------------------------------------------------------------------------
# inversion adapter (src/models/inversion_adapter.py:22-28; CLIPEncoderLayer of transformers 4.27.3)
# ---------------------------------------------------------------------------------------------------------------
def clip_encoder_layer(sd, p, x, heads, eps):
    n, T, H = x.shape
    d = H // heads
    a = layer_norm(sd, p + ".layer_norm1", x, eps)
    q = linear(sd, p + ".self_attn.q_proj", a) * (d ** -0.5)
    k = linear(sd, p + ".self_attn.k_proj", a)
    v = linear(sd, p + ".self_attn.v_proj", a)
    q = q.view(n, T, heads, d).transpose(1, 2)
    k = k.view(n, T, heads, d).transpose(1, 2)
    v = v.view(n, T, heads, d).transpose(1, 2)
    o = torch.matmul(torch.softmax(torch.matmul(q, k.transpose(-1, -2)), dim=-1), v).transpose(1, 2).reshape(n, T, H)
    x = x + linear(sd, p + ".self_attn.out_proj", o)
    a = layer_norm(sd, p + ".layer_norm2", x, eps)
    return x + linear(sd, p + ".mlp.fc2", F.gelu(linear(sd, p + ".mlp.fc1", a)))


def adapter_forward(sd, cfg, x):
    x = clip_encoder_layer(sd, "encoder_layers.0", x, cfg["heads"], cfg["layer_norm_eps"])
    x = layer_norm(sd, "post_layernorm", x[:, 0, :], cfg["layer_norm_eps"])
    x = F.gelu(linear(sd, "layers.0", x))       # Dropout(0.5) is identity in eval
    x = F.gelu(linear(sd, "layers.3", x))
    return linear(sd, "layers.6", x)
