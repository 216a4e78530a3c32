"""Prompt-embedding producer over libladi_native: drop-in for the reference's `encode_text_word_embedding`
(src/utils/encode_text_word_embedding.py:6-72, called at src/inference.py:291-295) and for the CLIPTextModel it drives.

    text_encoder = NativeCLIPTextEncoder(configs.TEXT_FULL, CLIPTextModel.state_dict())      # once
    ehs = encode_text_word_embedding(text_encoder, tokenized_text, word_embeddings, num_vstar).last_hidden_state

Same argument meaning and error behaviour as the reference: input_ids [B, 77] integer tensor (any device), word_embeddings
[B, num_vstar, 1024] (or [B, 1024] for a single pseudo-word), '$' = token id 259; sentences without '$' are encoded unchanged; pseudo-word
slots that would run past the sequence raise IndexError.  There is no CPU fallback.
"""
import ctypes
import torch

from . import _lib
from ._lib import NativeError, TextConfig, check, ptr, stream_ptr
from .modules import _Shim, _Weights


class TextEncoderOutput:
    """the fields of transformers' BaseModelOutputWithPooling the reference touches; also indexable like it ([0] = last_hidden_state,
    as in `self.text_encoder(ids, attention_mask=None)[0]`, tryon_pipe.py:249-253,297-301)"""

    def __init__(self, last_hidden_state, pooler_output):
        self.last_hidden_state, self.pooler_output, self.hidden_states, self.attentions = last_hidden_state, pooler_output, None, None

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]


class NativeCLIPTextEncoder(_Shim):
    """holds the text-encoder weights on the device (fp16) behind a ladi_text_encoder handle"""

    def __init__(self, cfg, state_dict):
        _lib.require_gpu()
        self.lib = _lib.load()
        c = TextConfig()
        c.vocab_size, c.hidden, c.heads, c.mlp_dim, c.layers = cfg["vocab_size"], cfg["hidden"], cfg["heads"], cfg["mlp_dim"], cfg["layers"]
        c.max_positions, c.vstar_token_id, c.layer_norm_eps = cfg["max_positions"], cfg.get("vstar_token_id", 259), cfg["layer_norm_eps"]
        with _Weights(state_dict) as w:
            self.h = self.lib.ladi_text_encoder_create(ctypes.byref(c), w.h)
        if not self.h:
            raise NativeError("ladi_text_encoder_create failed: " + _lib.last_error())
        self.cfg = dict(cfg)
        self.dtype = torch.float16
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ladi_text_encoder_destroy(self.h)
            self.h = None

    def __call__(self, input_ids, word_embeddings=None, num_vstar=1, attention_mask=None):
        if attention_mask is not None:
            raise ValueError("attention_mask is not supported (the SD2 text encoder config has no use_attention_mask: tryon_pipe.py:244-247 passes None)")
        # ids on the host (tokenizer output): validated there, errors like the reference.  ids already on the device (the reference's
        # `tokenized_text.to(device)`, inference.py:291): they stay there -- no device -> host copy, hence no synchronisation
        on_dev = input_ids.is_cuda
        ids = input_ids.reshape(-1, input_ids.shape[-1]).to(dtype=torch.int32).contiguous()
        if on_dev and ids.device != self.device:
            ids = ids.to(self.device)          # ids on ANOTHER GPU: their pointer means nothing to this encoder's device
        B, T = ids.shape
        H = self.cfg["hidden"]
        we = None
        if word_embeddings is not None:
            we = word_embeddings.unsqueeze(1) if word_embeddings.dim() == 2 else word_embeddings
            if we.shape[0] != B:
                raise AssertionError("word_embeddings batch %d != input_ids batch %d" % (we.shape[0], B))   # reference :31 assert
            if we.shape[1] < num_vstar or we.shape[2] != H:
                raise ValueError("word_embeddings must be [B, >= num_vstar, %d]" % H)
            we = we[:, :num_vstar].to(device=self.device, dtype=torch.float16).contiguous()
            if not on_dev:
                vs = self.cfg.get("vstar_token_id", 259)
                first = (ids == vs).int().argmax(dim=1)
                has = (ids == vs).any(dim=1)
                if bool((has & (first + num_vstar > T)).any()):
                    raise IndexError("pseudo-word slots run past the sequence end")      # what the reference's advanced indexing raises
        hidden = torch.empty((B, T, H), dtype=torch.float16, device=self.device)
        pooled = torch.empty((B, H), dtype=torch.float16, device=self.device)
        fwd = self.lib.ladi_text_encoder_forward_dev if on_dev else self.lib.ladi_text_encoder_forward
        check(fwd(self.h, ctypes.c_void_p(ids.data_ptr()), B, T, ptr(we) if we is not None else None, num_vstar, ptr(hidden), ptr(pooled),
                  stream_ptr()), "ladi_text_encoder_forward")
        self._ids_keepalive = ids          # the host-id form copies from this buffer asynchronously
        return TextEncoderOutput(hidden, pooled)


def encode_text_word_embedding(text_encoder, input_ids, word_embeddings, num_vstar=1):
    """same signature and result fields as the reference function; `text_encoder` is a NativeCLIPTextEncoder"""
    return text_encoder(input_ids, word_embeddings, num_vstar)
