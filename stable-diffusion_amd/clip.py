"""Host-side mirror of the reference's text conditioner on top of libsdmi.

``FrozenCLIPEmbedder`` (stable_diffusion.py:2553-2579) tokenises on the host and runs transformers'
``CLIPTextModel``; here the transformer runs on the MI355X (``sdmi_clip_text_*``) and the tokenizer is
supplied by the caller: its vocabulary is a download the reference fetches by name, so it cannot be
part of this package.  ``forward`` therefore accepts either token ids (a (B, T) integer tensor) or
text, the latter only when a ``tokenizer`` callable was given.
"""
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, dtype_code

# CLIPTextConfig of openai/clip-vit-large-patch14, the ``version`` the reference names (:2555)
CLIP_VIT_L14_TEXT = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                         num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5)


def clip_text_param_shapes(config=None) -> "OrderedDict[str, tuple]":
    """Checkpoint keys (below ``cond_stage_model.transformer.``) and shapes of the text transformer."""
    c = dict(CLIP_VIT_L14_TEXT, **(config or {}))
    d, f = c["hidden_size"], c["intermediate_size"]
    out = OrderedDict()
    out["text_model.embeddings.token_embedding.weight"] = (c["vocab_size"], d)
    out["text_model.embeddings.position_embedding.weight"] = (c["max_position_embeddings"], d)
    for i in range(c["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[f"{p}.self_attn.{n}.weight"] = (d, d)
            out[f"{p}.self_attn.{n}.bias"] = (d,)
        out[f"{p}.layer_norm1.weight"] = (d,)
        out[f"{p}.layer_norm1.bias"] = (d,)
        out[f"{p}.mlp.fc1.weight"] = (f, d)
        out[f"{p}.mlp.fc1.bias"] = (f,)
        out[f"{p}.mlp.fc2.weight"] = (d, f)
        out[f"{p}.mlp.fc2.bias"] = (d,)
        out[f"{p}.layer_norm2.weight"] = (d,)
        out[f"{p}.layer_norm2.bias"] = (d,)
    out["text_model.final_layer_norm.weight"] = (d,)
    out["text_model.final_layer_norm.bias"] = (d,)
    return out


class FrozenCLIPEmbedder:
    """Uses the CLIP transformer encoder for text (stable_diffusion.py:2553).

    ``version`` is kept for signature compatibility; nothing is downloaded.  ``config`` overrides
    entries of the ViT-L/14 text configuration (tests use small ones)."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, tokenizer=None,
                 config=None, compute_dtype=None):
        self.version = version
        self.device = torch.device(device)
        self.max_length = max_length
        self.tokenizer = tokenizer
        self.config = dict(CLIP_VIT_L14_TEXT, **(config or {}))
        compute_dtype = compute_dtype or "fp32"      # the reference's text encoder computes in fp32; 16 bit is opt-in
        self.compute_dtype = compute_dtype
        c = self.config
        cfg = _lib.ClipCfg()
        cfg.vocab_size, cfg.hidden, cfg.intermediate = c["vocab_size"], c["hidden_size"], c["intermediate_size"]
        cfg.layers, cfg.heads, cfg.max_positions = (c["num_hidden_layers"], c["num_attention_heads"],
                                                    c["max_position_embeddings"])
        cfg.ln_eps = c["layer_norm_eps"]
        cfg.compute_dtype = dtype_code(compute_dtype)
        self._h = ctypes.c_void_p()
        check(_lib.lib().sdmi_clip_text_create(ctypes.byref(cfg), ctypes.byref(self._h)), "clip_text_create")
        self._finalized = False

    def freeze(self):
        return self

    def eval(self):
        return self

    def state_dict_shapes(self):
        lib = _lib.lib()
        out = OrderedDict()
        key = ctypes.c_char_p()
        shape = (ctypes.c_int64 * 4)()
        nd = ctypes.c_int()
        for i in range(lib.sdmi_clip_text_num_weights(self._h)):
            check(lib.sdmi_clip_text_weight_info(self._h, i, ctypes.byref(key), shape, ctypes.byref(nd)), "weight_info")
            out[key.value.decode()] = tuple(shape[j] for j in range(nd.value))
        return out

    def load_state_dict(self, state_dict, strict=True, prefix=""):
        """Keys as in an SD checkpoint below ``cond_stage_model.`` (``transformer.text_model.*``), as in a
        CLIPTextModel state dict (``text_model.*``), or without the ``text_model.`` level (newer
        transformers releases)."""
        lib = _lib.lib()
        shapes = self.state_dict_shapes()

        def find(k):
            for cand in (prefix + "transformer." + k, prefix + k, prefix + k[len("text_model."):]):
                if cand in state_dict:
                    return cand
            return None

        missing = [k for k in shapes if find(k) is None]
        if missing:
            raise RuntimeError(f"Error(s) in loading state_dict for FrozenCLIPEmbedder: missing {missing[:5]}"
                               f"{' ...' if len(missing) > 5 else ''}")
        used = set()
        for k, shp in shapes.items():
            src = find(k)
            used.add(src)
            t = state_dict[src]
            if tuple(t.shape) != shp:
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {shp}")
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            shape = (ctypes.c_int64 * len(shp))(*shp)
            check(lib.sdmi_clip_text_set_weight(self._h, k.encode(), ptr(t), shape, len(shp), _lib.F32, stream_ptr()),
                  f"set_weight({k})")
            torch.cuda.current_stream().synchronize()
        check(lib.sdmi_clip_text_finalize(self._h, stream_ptr()), "clip_text_finalize")
        self._finalized = True
        unexpected = [k for k in state_dict if k.startswith(prefix) and k not in used
                      and not k.endswith("position_ids")]
        return [], unexpected

    def tokenize(self, text):
        if self.tokenizer is None:
            raise RuntimeError("FrozenCLIPEmbedder: no tokenizer was supplied (the CLIP vocabulary is a download and "
                               "is not bundled); pass token ids, or construct with tokenizer=<callable>")
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    def forward(self, text):
        """text: str / list of str (needs a tokenizer) or a (B, T) integer tensor of token ids ->
        last_hidden_state (B, T, hidden) fp32 on the device (:2569-2576)."""
        if not self._finalized:
            raise RuntimeError("FrozenCLIPEmbedder.forward called before load_state_dict")
        tokens = text if torch.is_tensor(text) else self.tokenize(text)
        tokens = torch.as_tensor(tokens)
        if tokens.dim() != 2:
            raise ValueError("token ids must have shape (batch, tokens)")
        if not tokens.is_cuda:
            lo, hi = int(tokens.min()), int(tokens.max())
            if lo < 0 or hi >= self.config["vocab_size"]:
                raise IndexError(f"token id out of range [0, {self.config['vocab_size']}): {lo}..{hi}")
        tokens = tokens.to(device=self.device, dtype=torch.int64).contiguous()
        b, t = tokens.shape
        out = torch.empty((b, t, self.config["hidden_size"]), dtype=torch.float32, device=self.device)
        check(_lib.lib().sdmi_clip_text_forward(self._h, ptr(tokens), ptr(out), b, t, stream_ptr()), "clip_text_forward")
        return out

    __call__ = forward

    def encode(self, text):
        return self(text)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().sdmi_clip_text_destroy(self._h)
                self._h = None
        except Exception:
            pass
