"""CPU fp32 oracle of the prompt-embedding producer.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.

Restates
  * `encode_text_word_embedding` (reference src/utils/encode_text_word_embedding.py:6-72): the '$' (token id 259) splice of the
    inversion adapter's pseudo-word embeddings into the token embeddings (:12-35), position embeddings (:37-38), causal CLIP text
    transformer (:40-56), final LayerNorm (:57), pooled output at argmax(input_ids) (:62-65);
  * the `CLIPTextTransformer` internals it drives -- third-party: transformers==4.27.3 (requirements of the reference; not vendored),
    models/clip/modeling_clip.py: CLIPEncoderLayer = pre-LN block, CLIPAttention scales q by head_dim**-0.5 and adds an additive
    causal mask before the softmax, CLIPMLP = fc2(act(fc1(x))), hidden_act "gelu" for the SD2 text encoder.

Pinned (tests/test_cpu.py::test_text_oracle_matches_reference_golden) against tests/golden/clip_text_tiny.safetensors, which
oracle/make_golden.py produces by running the reference's OWN encode_text_word_embedding on the installed transformers CLIPTextModel
(through a thin adapter that presents the 4.27 attribute layout the function reaches into).
"""
import torch
import torch.nn.functional as F


def _get(sd, key):
    if key in sd:
        return sd[key]
    alt = key[len("text_model."):] if key.startswith("text_model.") else "text_model." + key
    return sd[alt]


def _linear(sd, p, x):
    return F.linear(x, _get(sd, p + ".weight"), _get(sd, p + ".bias"))


def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), _get(sd, p + ".weight"), _get(sd, p + ".bias"), eps)


def splice_word_embeddings(input_embeds, input_ids, word_embeddings, num_vstar, vstar_id=259):
    """encode_text_word_embedding.py:12-35 — in every sentence that contains '$', the num_vstar positions starting at its FIRST '$'
    are overwritten with that sentence's pseudo-word embeddings; sentences without '$' are left untouched."""
    out = input_embeds.clone()
    if word_embeddings is None:
        return out
    if word_embeddings.dim() == 2:
        word_embeddings = word_embeddings.unsqueeze(1)
    B, T = input_ids.shape
    assert word_embeddings.shape[0] == B
    for b in range(B):
        pos = (input_ids[b] == vstar_id).nonzero()
        if pos.numel() == 0:
            continue
        f = int(pos[0])
        if f + num_vstar > T:
            raise IndexError("pseudo-word slots run past the sequence end (the reference raises here too)")
        out[b, f:f + num_vstar] = word_embeddings[b, :num_vstar].to(out.dtype)
    return out


def clip_text_forward(sd, cfg, input_ids, word_embeddings=None, num_vstar=1):
    """returns (last_hidden_state [B,T,H] after final_layer_norm, pooled_output [B,H])"""
    eps, heads = cfg["layer_norm_eps"], cfg["heads"]
    ids = input_ids.view(-1, input_ids.shape[-1]).long()
    B, T = ids.shape
    x = F.embedding(ids, _get(sd, "text_model.embeddings.token_embedding.weight"))
    x = splice_word_embeddings(x, ids, word_embeddings, num_vstar, cfg.get("vstar_token_id", 259))
    x = x + _get(sd, "text_model.embeddings.position_embedding.weight")[:T].unsqueeze(0)
    H = x.shape[-1]
    d = H // heads
    causal = torch.full((T, T), float("-inf")).triu_(1)
    for i in range(cfg["layers"]):
        p = "text_model.encoder.layers.%d" % i
        a = _ln(sd, p + ".layer_norm1", x, eps)
        q = (_linear(sd, p + ".self_attn.q_proj", a) * d ** -0.5).view(B, T, heads, d).transpose(1, 2)
        k = _linear(sd, p + ".self_attn.k_proj", a).view(B, T, heads, d).transpose(1, 2)
        v = _linear(sd, p + ".self_attn.v_proj", a).view(B, T, heads, d).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-1, -2) + causal, dim=-1)
        x = x + _linear(sd, p + ".self_attn.out_proj", (w @ v).transpose(1, 2).reshape(B, T, H))
        a = _ln(sd, p + ".layer_norm2", x, eps)
        x = x + _linear(sd, p + ".mlp.fc2", F.gelu(_linear(sd, p + ".mlp.fc1", a)))
    x = _ln(sd, "text_model.final_layer_norm", x, eps)
    pooled = x[torch.arange(B), ids.to(torch.int).argmax(dim=-1)]
    return x, pooled
