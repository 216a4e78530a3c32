"""Generate the golden fixtures under tests/golden/ by running the REAL reference code where it is importable in the build
container: src/models/emasc.py EMASC, src/utils/data_utils.py mask_features, and the installed
transformers CLIPEncoderLayer (the class src/models/inversion_adapter.py:2,9 wraps; v5 call convention `layer(x, None)`), and
src/utils/encode_text_word_embedding.py on the installed transformers CLIPTextModel (through a 4.27-layout facade).

Run here (needs /root/reference):  python -m oracle.make_golden
The GPU box has no /root/reference: tests only read the committed .safetensors files.
"""
import os
import sys

import torch
from safetensors.torch import save_file

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.path.insert(0, REF)
    from src.models.emasc import EMASC  # noqa: E402  (reference code, imported not copied)
    from src.utils.data_utils import mask_features  # noqa: E402
    from ladi_vton_amd import configs as C

    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # ---- EMASC + mask_features (tiny channel config, 64x48 base resolution)
    ecfg = C.EMASC_TINY
    sd = C.synth_state_dict(C.emasc_shapes(ecfg), "emasc.")
    m = EMASC(list(ecfg["in_channels"]), list(ecfg["out_channels"]), kernel_size=3, padding=1, stride=1, type="nonlinear").eval()
    m.load_state_dict(sd, strict=True)
    B, H, W = 1, 32, 24
    g = torch.Generator().manual_seed(11)
    sizes = [(H, W), (H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    feats = [torch.randn((B, c, h, w), generator=g).half().float() for c, (h, w) in zip(ecfg["in_channels"], sizes)]
    mask = (torch.rand((B, 1, H, W), generator=g) > 0.6).float()
    with torch.no_grad():
        outs = m([f.clone() for f in feats])
        masked = mask_features([o.clone() for o in outs], mask.clone())
    blob = {}
    for i in range(5):
        blob["feat%d" % i] = feats[i].contiguous()
        blob["emasc%d" % i] = outs[i].contiguous()
        blob["masked%d" % i] = masked[i].contiguous()
    blob["mask"] = mask
    save_file(blob, os.path.join(OUT, "emasc_tiny.safetensors"))

    # ---- CLIPEncoderLayer (adapter's encoder layer), tiny vision config
    from transformers import CLIPVisionConfig
    from transformers.models.clip.modeling_clip import CLIPEncoderLayer
    acfg = C.ADAPTER_TINY
    vc = CLIPVisionConfig(hidden_size=acfg["hidden"], intermediate_size=acfg["mlp_dim"], num_attention_heads=acfg["heads"],
                          layer_norm_eps=acfg["layer_norm_eps"], hidden_act="gelu", attention_dropout=0.0)
    try:
        vc._attn_implementation = "eager"
    except Exception:
        pass
    layer = CLIPEncoderLayer(vc).eval()
    asd = C.synth_state_dict(C.adapter_shapes(acfg), "adapter.")
    lsd = {k[len("encoder_layers.0."):]: v for k, v in asd.items() if k.startswith("encoder_layers.0.")}
    layer.load_state_dict(lsd, strict=True)
    x = torch.randn((2, 17, acfg["hidden"]), generator=g).half().float()
    with torch.no_grad():
        y = layer(x, None)
        if isinstance(y, (tuple, list)):
            y = y[0]
    save_file({"x": x, "y": y.contiguous()}, os.path.join(OUT, "clip_encoder_layer_tiny.safetensors"))

    make_text_golden(g)
    make_vision_golden(g)
    print("wrote", os.listdir(OUT))


class _Emb42:
    """transformers-4.27 view of CLIPTextEmbeddings: the attributes encode_text_word_embedding.py:26-37 reaches into"""
    def __init__(self, emb, T):
        self.token_embedding, self.position_embedding = emb.token_embedding, emb.position_embedding
        self.position_ids = torch.arange(T).unsqueeze(0)


class _Out42(tuple):
    hidden_states = None
    attentions = None


class _TextModel42:
    """transformers-4.27 `CLIPTextModel.text_model` facade over the installed (5.x, flattened) CLIPTextModel: same modules, same
    arithmetic; only attribute names / call conventions are adapted so that the reference function runs unmodified"""
    def __init__(self, m, T):
        self.m, self.embeddings, self.final_layer_norm = m, _Emb42(m.embeddings, T), m.final_layer_norm

    @staticmethod
    def _build_causal_attention_mask(bsz, seq_len, dtype):   # modeling_clip.py (4.27.3) CLIPTextTransformer._build_causal_attention_mask
        mask = torch.empty(bsz, seq_len, seq_len, dtype=dtype)
        mask.fill_(torch.finfo(dtype).min)
        mask.triu_(1)
        return mask.unsqueeze(1)

    def encoder(self, inputs_embeds, attention_mask, causal_attention_mask, output_attentions, output_hidden_states, return_dict):
        out = self.m.encoder(inputs_embeds=inputs_embeds, attention_mask=causal_attention_mask, is_causal=True)
        return _Out42((out.last_hidden_state,))


def make_vision_golden(g):
    """tests/golden/clip_vision_tiny.safetensors: installed transformers CLIPVisionModel (the class behind inference.py:269-273's
    vision_encoder) with the deterministic tiny checkpoint: last_hidden_state (what the adapter consumes) and pooler_output"""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from ladi_vton_amd import configs as C
    vc = C.VISION_TINY
    hc = CLIPVisionConfig(hidden_size=vc["hidden"], intermediate_size=vc["mlp_dim"], num_hidden_layers=vc["layers"],
                          num_attention_heads=vc["heads"], image_size=vc["image_size"], patch_size=vc["patch_size"], hidden_act="gelu",
                          layer_norm_eps=vc["layer_norm_eps"], attention_dropout=0.0)
    try:
        hc._attn_implementation = "eager"
    except Exception:
        pass
    m = CLIPVisionModel(hc).eval()
    sd = C.synth_state_dict(C.vision_shapes(vc), "vision.")
    own = m.state_dict()
    flat = {(k[len("vision_model."):] if k[len("vision_model."):] in own else k): v for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(flat, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    x = (torch.randn((3, 3, vc["image_size"], vc["image_size"]), generator=g) * 1.2).half().float()   # CLIP-normalised pixel range
    with torch.no_grad():
        out = m(pixel_values=x)
    save_file({"pixel_values": x, "last_hidden_state": out.last_hidden_state.contiguous(), "pooler_output": out.pooler_output.contiguous()},
              os.path.join(OUT, "clip_vision_tiny.safetensors"))


def make_text_golden(g):
    """tests/golden/clip_text_tiny.safetensors: the reference's own encode_text_word_embedding (imported, not copied) on the installed
    transformers CLIPTextModel with the deterministic tiny checkpoint; rows with and without '$', num_vstar = 3"""
    from transformers import CLIPTextConfig, CLIPTextModel
    from src.utils.encode_text_word_embedding import encode_text_word_embedding  # noqa: E402  (reference code)
    from ladi_vton_amd import configs as C
    tc = C.TEXT_TINY
    hc = CLIPTextConfig(vocab_size=tc["vocab_size"], hidden_size=tc["hidden"], intermediate_size=tc["mlp_dim"], num_hidden_layers=tc["layers"],
                        num_attention_heads=tc["heads"], max_position_embeddings=tc["max_positions"], hidden_act="gelu",
                        layer_norm_eps=tc["layer_norm_eps"], attention_dropout=0.0, bos_token_id=1, eos_token_id=2)
    try:
        hc._attn_implementation = "eager"
    except Exception:
        pass
    m = CLIPTextModel(hc).eval()
    sd = C.synth_state_dict(C.text_shapes(tc), "text.")
    own = m.state_dict()
    flat = {(k[len("text_model."):] if k[len("text_model."):] in own else k): v for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(flat, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    B, T, NV = 4, tc["max_positions"], 3
    ids = torch.randint(3, 250, (B, T), generator=g)
    ids[:, 0] = 300                                  # "bos"-like high id is NOT the argmax ...
    eot = [20, 9, 40, 76]
    for b in range(B):
        ids[b, eot[b]] = 319                         # ... the highest id marks the pooled position (eot convention, :62-65)
        ids[b, eot[b] + 1:] = 0
    ids[0, 10:13] = 259                              # three consecutive '$' (the prompt template of inference.py:289)
    ids[2, 5] = 259; ids[2, 30] = 259                # only the FIRST '$' of a sentence anchors the splice
    # rows 1 and 3 have no '$': untouched
    we = torch.randn((B, NV, tc["hidden"]), generator=g).half().float()
    shim = type("TextEncoder42", (), {})()
    shim.text_model = _TextModel42(m, T)
    with torch.no_grad():
        out = encode_text_word_embedding(shim, ids.clone(), we.clone(), NV)
        plain = m(input_ids=ids)                     # sanity of the facade: without splice rows it must equal the model's own forward
        out_nosplice = encode_text_word_embedding(shim, ids[[1, 3]].clone(), we[[1, 3]].clone(), NV)
    assert torch.allclose(out_nosplice.last_hidden_state, plain.last_hidden_state[[1, 3]], atol=1e-5), "4.27 facade != installed forward"
    save_file({"input_ids": ids.int(), "word_embeddings": we, "last_hidden_state": out.last_hidden_state.contiguous(),
               "pooler_output": out.pooler_output.contiguous()}, os.path.join(OUT, "clip_text_tiny.safetensors"))


if __name__ == "__main__":
    main()
