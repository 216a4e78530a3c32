"""Frozen model configurations (the HF hub is unreachable; values per SURVEY.md App. A.0 / hubconf.py:17-55) and the
deterministic synthetic checkpoint generator.  Shared data: used by the product (bench / smoke build random-init models from it) and by the oracle."""
import zlib
from collections import OrderedDict

import torch

# ---------------------------------------------------------------------------------------------------------------
# configs: "full" = released LaDI-VTON hyper-parameters; "tiny" = same topology, small widths for CPU-sized tests
# ---------------------------------------------------------------------------------------------------------------
UNET_FULL = dict(in_channels=31, out_channels=4, block_out_channels=(320, 640, 1280, 1280), num_heads=(5, 10, 20, 20),
                 layers_per_block=2, cross_attention_dim=1024, norm_num_groups=32, norm_eps=1e-5)
UNET_TINY = dict(in_channels=31, out_channels=4, block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4),
                 layers_per_block=2, cross_attention_dim=128, norm_num_groups=32, norm_eps=1e-5)
VAE_FULL = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
VAE_TINY = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(64, 64, 128, 128),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
EMASC_FULL = dict(in_channels=(128, 128, 128, 256, 512), out_channels=(128, 256, 512, 512, 512))
EMASC_TINY = dict(in_channels=(64, 64, 64, 64, 128), out_channels=(64, 64, 128, 128, 128))
ADAPTER_FULL = dict(hidden=1280, heads=16, mlp_dim=5120, head_hidden=5120, out_dim=16384, layer_norm_eps=1e-5)
ADAPTER_TINY = dict(hidden=128, heads=2, mlp_dim=256, head_hidden=256, out_dim=16 * 128, layer_norm_eps=1e-5)
# CLIP text encoder of stabilityai/stable-diffusion-2-inpainting (OpenCLIP ViT-H/14 text tower minus its last layer)
TEXT_FULL = dict(vocab_size=49408, hidden=1024, heads=16, mlp_dim=4096, layers=23, max_positions=77, layer_norm_eps=1e-5, vstar_token_id=259)
TEXT_TINY = dict(vocab_size=320, hidden=128, heads=2, mlp_dim=256, layers=2, max_positions=77, layer_norm_eps=1e-5, vstar_token_id=259)


def emasc_for_vae(vae_cfg):
    """EMASC channel lists implied by a VAE config (hubconf.py:42-44 for the released one)."""
    b = vae_cfg["block_out_channels"]
    return dict(in_channels=(b[0], b[0], b[0], b[1], b[2]), out_channels=(b[0], b[1], b[2], b[3], b[3]))


# ---------------------------------------------------------------------------------------------------------------
# state-dict key/shape enumeration (diffusers 0.14 naming)
# ---------------------------------------------------------------------------------------------------------------
def _conv(sd, name, cin, cout, k):
    sd[name + ".weight"] = (cout, cin, k, k)
    sd[name + ".bias"] = (cout,)


def _lin(sd, name, cin, cout, bias=True):
    sd[name + ".weight"] = (cout, cin)
    if bias:
        sd[name + ".bias"] = (cout,)


def _norm(sd, name, c):
    sd[name + ".weight"] = (c,)
    sd[name + ".bias"] = (c,)


def _resnet(sd, p, cin, cout, temb):
    _norm(sd, p + ".norm1", cin)
    _conv(sd, p + ".conv1", cin, cout, 3)
    if temb:
        _lin(sd, p + ".time_emb_proj", temb, cout)
    _norm(sd, p + ".norm2", cout)
    _conv(sd, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(sd, p + ".conv_shortcut", cin, cout, 1)


def _transformer(sd, p, c, cross):
    _norm(sd, p + ".norm", c)
    _lin(sd, p + ".proj_in", c, c)
    b = p + ".transformer_blocks.0"
    for i in (1, 2, 3):
        _norm(sd, b + ".norm%d" % i, c)
    for a, kd in (("attn1", c), ("attn2", cross)):
        _lin(sd, b + "." + a + ".to_q", c, c, bias=False)
        _lin(sd, b + "." + a + ".to_k", kd, c, bias=False)
        _lin(sd, b + "." + a + ".to_v", kd, c, bias=False)
        _lin(sd, b + "." + a + ".to_out.0", c, c)
    _lin(sd, b + ".ff.net.0.proj", c, 8 * c)
    _lin(sd, b + ".ff.net.2", 4 * c, c)
    _lin(sd, p + ".proj_out", c, c)


def unet_shapes(cfg):
    sd = OrderedDict()
    boc = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    cross = cfg["cross_attention_dim"]
    temb = boc[0] * 4
    _conv(sd, "conv_in", cfg["in_channels"], boc[0], 3)
    _lin(sd, "time_embedding.linear_1", boc[0], temb)
    _lin(sd, "time_embedding.linear_2", temb, temb)
    ch = boc[0]
    for i in range(4):
        for j in range(L):
            _resnet(sd, "down_blocks.%d.resnets.%d" % (i, j), ch, boc[i], temb)
            ch = boc[i]
            if i < 3:
                _transformer(sd, "down_blocks.%d.attentions.%d" % (i, j), ch, cross)
        if i < 3:
            _conv(sd, "down_blocks.%d.downsamplers.0.conv" % i, ch, ch, 3)
    _resnet(sd, "mid_block.resnets.0", ch, ch, temb)
    _transformer(sd, "mid_block.attentions.0", ch, cross)
    _resnet(sd, "mid_block.resnets.1", ch, ch, temb)
    rb = list(reversed(boc))
    for i in range(4):
        out = rb[i]
        prev = rb[0] if i == 0 else rb[i - 1]
        inp = rb[min(i + 1, 3)]
        for j in range(L + 1):
            skip = inp if j == L else out
            rin = prev if j == 0 else out
            _resnet(sd, "up_blocks.%d.resnets.%d" % (i, j), rin + skip, out, temb)
            if i > 0:
                _transformer(sd, "up_blocks.%d.attentions.%d" % (i, j), out, cross)
        if i < 3:
            _conv(sd, "up_blocks.%d.upsamplers.0.conv" % i, out, out, 3)
    _norm(sd, "conv_norm_out", boc[0])
    _conv(sd, "conv_out", boc[0], cfg["out_channels"], 3)
    return sd


def _vae_attn(sd, p, c):
    _norm(sd, p + ".group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        _lin(sd, p + "." + n, c, c)


def vae_shapes(cfg):
    sd = OrderedDict()
    boc = cfg["block_out_channels"]
    L = cfg["layers_per_block"]
    z = cfg["latent_channels"]
    _conv(sd, "encoder.conv_in", cfg["in_channels"], boc[0], 3)
    ch = boc[0]
    for i in range(4):
        for j in range(L):
            _resnet(sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), ch, boc[i], None)
            ch = boc[i]
        if i < 3:
            _conv(sd, "encoder.down_blocks.%d.downsamplers.0.conv" % i, ch, ch, 3)
    _resnet(sd, "encoder.mid_block.resnets.0", ch, ch, None)
    _vae_attn(sd, "encoder.mid_block.attentions.0", ch)
    _resnet(sd, "encoder.mid_block.resnets.1", ch, ch, None)
    _norm(sd, "encoder.conv_norm_out", ch)
    _conv(sd, "encoder.conv_out", ch, 2 * z, 3)
    _conv(sd, "quant_conv", 2 * z, 2 * z, 1)
    _conv(sd, "post_quant_conv", z, z, 1)
    rb = list(reversed(boc))
    _conv(sd, "decoder.conv_in", z, rb[0], 3)
    _resnet(sd, "decoder.mid_block.resnets.0", rb[0], rb[0], None)
    _vae_attn(sd, "decoder.mid_block.attentions.0", rb[0])
    _resnet(sd, "decoder.mid_block.resnets.1", rb[0], rb[0], None)
    ch = rb[0]
    for i in range(4):
        for j in range(L + 1):
            _resnet(sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), ch, rb[i], None)
            ch = rb[i]
        if i < 3:
            _conv(sd, "decoder.up_blocks.%d.upsamplers.0.conv" % i, ch, ch, 3)
    _norm(sd, "decoder.conv_norm_out", ch)
    _conv(sd, "decoder.conv_out", ch, cfg["out_channels"], 3)
    return sd


def emasc_shapes(cfg):
    sd = OrderedDict()
    for i, (ci, co) in enumerate(zip(cfg["in_channels"], cfg["out_channels"])):
        _conv(sd, "conv.%d.0" % i, ci, ci, 3)
        _conv(sd, "conv.%d.2" % i, ci, co, 3)
    return sd


# warping module (hubconf.py:56-66): ConvNet_TPS(256, 192, input_nc=21, n_layer=3) + UNetVanilla(24, 3, bilinear=True)
TPS_FULL = dict(height=256, width=192, input_nc=21, n_layers=3, grid=5, ngf=64)
REFINE_FULL = dict(in_channels=24, out_channels=3, base=64)


def _bn(sd, name, c):
    for k in ("weight", "bias", "running_mean", "running_var"):
        sd[name + "." + k] = (c,)


def tps_shapes(cfg):
    """state_dict layout of src/models/ConvNet_TPS.py ConvNet_TPS (nn.Sequential indices; num_batches_tracked and the TPSGridGen
    buffers are derived, not parameters)"""
    sd = OrderedDict()
    ngf, nl = cfg["ngf"], cfg["n_layers"]

    def extraction(prefix, cin):
        i = 0
        _conv(sd, "%s.model.%d" % (prefix, i), cin, ngf, 4); _bn(sd, "%s.model.%d" % (prefix, i + 2), ngf); i += 3
        for l in range(nl):
            a = min(2 ** l * ngf, 512)
            b = 2 ** (l + 1) * ngf if 2 ** l * ngf < 512 else 512
            _conv(sd, "%s.model.%d" % (prefix, i), a, b, 4); _bn(sd, "%s.model.%d" % (prefix, i + 2), b); i += 3
        _conv(sd, "%s.model.%d" % (prefix, i), 512, 512, 3); _bn(sd, "%s.model.%d" % (prefix, i + 2), 512); i += 3
        _conv(sd, "%s.model.%d" % (prefix, i), 512, 512, 3)

    extraction("extractionA", 3)
    extraction("extractionB", cfg["input_nc"])
    r = "loc_net.regression.conv"
    corr = (cfg["height"] // 16) * (cfg["width"] // 16)
    for i, (a, b, k) in enumerate(((corr, 512, 4), (512, 256, 4), (256, 128, 3), (128, 64, 3))):
        _conv(sd, "%s.%d" % (r, 3 * i), a, b, k); _bn(sd, "%s.%d" % (r, 3 * i + 1), b)
    _lin(sd, "loc_net.regression.linear", 64 * (cfg["height"] // 64) * (cfg["width"] // 64), 2 * cfg["grid"] ** 2)
    return sd


def refine_shapes(cfg):
    """state_dict layout of src/models/UNet.py UNetVanilla(bilinear=True) (src/models/unet_parts.py)"""
    sd = OrderedDict()
    b = cfg["base"]

    def dconv(name, cin, cout, mid=None):
        mid = mid or cout
        sd[name + ".double_conv.0.weight"] = (mid, cin, 3, 3); _bn(sd, name + ".double_conv.1", mid)
        sd[name + ".double_conv.3.weight"] = (cout, mid, 3, 3); _bn(sd, name + ".double_conv.4", cout)

    dconv("inc", cfg["in_channels"], b)
    dconv("down1.maxpool_conv.1", b, 2 * b)
    dconv("down2.maxpool_conv.1", 2 * b, 4 * b)
    dconv("down3.maxpool_conv.1", 4 * b, 8 * b)
    dconv("down4.maxpool_conv.1", 8 * b, 8 * b)               # 1024 // 2 with bilinear upsampling
    dconv("up1.conv", 16 * b, 4 * b, 8 * b)
    dconv("up2.conv", 8 * b, 2 * b, 4 * b)
    dconv("up3.conv", 4 * b, b, 2 * b)
    dconv("up4.conv", 2 * b, b, b)
    _conv(sd, "outc.conv", b, cfg["out_channels"], 1)
    return sd


# CLIP ViT-H/14 vision tower of laion/CLIP-ViT-H-14-laion2B-s32B-b79K: feeds the inversion adapter
VISION_FULL = dict(hidden=1280, heads=16, mlp_dim=5120, layers=32, image_size=224, patch_size=14, layer_norm_eps=1e-5)
VISION_TINY = dict(hidden=320, heads=4, mlp_dim=640, layers=2, image_size=56, patch_size=14, layer_norm_eps=1e-5)


def vision_shapes(cfg):
    """transformers 4.27 CLIPVisionModel(WithProjection) key layout: vision_model.* (visual_projection is not on the path)"""
    sd = OrderedDict()
    h, ps = cfg["hidden"], cfg["patch_size"]
    ntok = (cfg["image_size"] // ps) ** 2 + 1
    sd["vision_model.embeddings.class_embedding"] = (h,)
    sd["vision_model.embeddings.patch_embedding.weight"] = (h, 3, ps, ps)
    sd["vision_model.embeddings.position_embedding.weight"] = (ntok, h)
    _norm(sd, "vision_model.pre_layrnorm", h)
    for i in range(cfg["layers"]):
        e = "vision_model.encoder.layers.%d" % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(sd, e + ".self_attn." + n, h, h)
        _norm(sd, e + ".layer_norm1", h)
        _lin(sd, e + ".mlp.fc1", h, cfg["mlp_dim"])
        _lin(sd, e + ".mlp.fc2", cfg["mlp_dim"], h)
        _norm(sd, e + ".layer_norm2", h)
    _norm(sd, "vision_model.post_layernorm", h)
    return sd


def text_shapes(cfg):
    """transformers 4.27 CLIPTextModel key layout (the layout of the released text_encoder checkpoint): text_model.*"""
    sd = OrderedDict()
    h = cfg["hidden"]
    sd["text_model.embeddings.token_embedding.weight"] = (cfg["vocab_size"], h)
    sd["text_model.embeddings.position_embedding.weight"] = (cfg["max_positions"], h)
    for i in range(cfg["layers"]):
        e = "text_model.encoder.layers.%d" % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(sd, e + ".self_attn." + n, h, h)
        _norm(sd, e + ".layer_norm1", h)
        _lin(sd, e + ".mlp.fc1", h, cfg["mlp_dim"])
        _lin(sd, e + ".mlp.fc2", cfg["mlp_dim"], h)
        _norm(sd, e + ".layer_norm2", h)
    _norm(sd, "text_model.final_layer_norm", h)
    return sd


def adapter_shapes(cfg):
    sd = OrderedDict()
    h = cfg["hidden"]
    e = "encoder_layers.0"
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
        _lin(sd, e + ".self_attn." + n, h, h)
    _norm(sd, e + ".layer_norm1", h)
    _lin(sd, e + ".mlp.fc1", h, cfg["mlp_dim"])
    _lin(sd, e + ".mlp.fc2", cfg["mlp_dim"], h)
    _norm(sd, e + ".layer_norm2", h)
    _norm(sd, "post_layernorm", h)
    _lin(sd, "layers.0", h, cfg["head_hidden"])
    _lin(sd, "layers.3", cfg["head_hidden"], cfg["head_hidden"])
    _lin(sd, "layers.6", cfg["head_hidden"], cfg["out_dim"])
    return sd


def param_count(shapes):
    n = 0
    for s in shapes.values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n


# ---------------------------------------------------------------------------------------------------------------
# deterministic synthetic checkpoint: order-independent, keyed by name
# ---------------------------------------------------------------------------------------------------------------
def _is_norm(key):
    leaf = key.rsplit(".", 2)[-2]
    return "norm" in leaf or leaf == "post_layernorm"


def synth_tensor(key, shape, scope=""):
    g = torch.Generator().manual_seed(zlib.crc32((scope + key).encode()) & 0x7FFFFFFF)
    u = torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1
    if key.endswith(".running_var"):
        return 0.5 + torch.rand(shape, generator=g, dtype=torch.float32)
    if key.endswith(".running_mean"):
        return 0.1 * u
    if _is_norm(key):
        return 1.0 + 0.1 * u if key.endswith(".weight") else 0.1 * u
    if key.endswith(".weight"):
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return u * (3.0 / fan_in) ** 0.5     # variance-preserving uniform
    return u * 0.1                            # biases


def synth_items(shapes, scope="", fp16_round=True):
    """generator of (key, tensor) pairs of the synthetic checkpoint: lets a loader stream 866 M parameters without holding the
    whole fp32 state_dict in host memory (one per rank under torchrun)"""
    for k, s in shapes.items():
        t = synth_tensor(k, s, scope)
        yield k, (t.half().float() if fp16_round else t)


def synth_state_dict(shapes, scope="", fp16_round=True):
    """fp32 tensors whose values are exactly representable in fp16 (so weight quantisation is not counted as error)."""
    sd = OrderedDict()
    for k, s in shapes.items():
        t = synth_tensor(k, s, scope)
        if fp16_round:
            t = t.half().float()
        sd[k] = t
    return sd
