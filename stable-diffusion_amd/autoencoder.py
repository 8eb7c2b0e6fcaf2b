"""Host-side mirror of the reference's first-stage decode path on top of libsdmi.

``AutoencoderKL(ddconfig, lossconfig, embed_dim).decode(z)`` and ``Decoder(**ddconfig)`` keep the
reference's names and checkpoint keys (stable_diffusion.py:1370-1419, :422-527).  Only decode() is
on the device path; encode() belongs to the callers that run once per image and raises.
"""
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, dtype_code
from . import structure as S


def _create_handle(ddconfig: dict, embed_dim: int, compute_dtype="fp16"):
    S.vae_decoder_structure(ddconfig, embed_dim)        # validates unsupported options loudly
    cfg = _lib.VaeCfg()
    cfg.z_channels = ddconfig["z_channels"]
    cfg.embed_dim = embed_dim
    cfg.ch = ddconfig["ch_out"]
    cfg.out_ch = ddconfig["out_ch"]
    mult = list(ddconfig.get("ch_mult", (1, 2, 4, 8)))
    cfg.n_ch_mult = len(mult)
    for i, m in enumerate(mult):
        cfg.ch_mult[i] = m
    cfg.num_res_blocks = ddconfig["num_res_blocks"]
    cfg.compute_dtype = dtype_code(compute_dtype)
    h = ctypes.c_void_p()
    check(_lib.lib().sdmi_vae_decoder_create(ctypes.byref(cfg), ctypes.byref(h)), "vae_decoder_create")
    return h


def _handle_weight_shapes(h):
    lib = _lib.lib()
    out = OrderedDict()
    key = ctypes.c_char_p()
    shape = (ctypes.c_int64 * 4)()
    nd = ctypes.c_int()
    for i in range(lib.sdmi_vae_decoder_num_weights(h)):
        check(lib.sdmi_vae_decoder_weight_info(h, i, ctypes.byref(key), shape, ctypes.byref(nd)), "weight_info")
        out[key.value.decode()] = tuple(shape[j] for j in range(nd.value))
    return out


class AutoencoderKL:
    """Drop-in for stable_diffusion.AutoencoderKL on the decode path."""

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, compute_dtype="fp16", device=None):
        assert ddconfig["double_z"]
        if ckpt_path is not None:
            raise NotImplementedError("pass tensors through load_state_dict (no checkpoint files offline)")
        self.ddconfig = dict(ddconfig)
        self.embed_dim = embed_dim
        self.compute_dtype = compute_dtype
        self.device = torch.device(device if device is not None else "cuda")
        self._h = _create_handle(self.ddconfig, embed_dim, compute_dtype)
        self._shapes = _handle_weight_shapes(self._h)
        self._finalized = False

    def eval(self):
        return self

    def state_dict_shapes(self):
        return OrderedDict(self._shapes)

    def load_state_dict(self, state_dict, strict=True, prefix=""):
        """Decoder-side keys only (post_quant_conv.*, decoder.*); encoder keys are ignored."""
        if self._finalized:
            raise RuntimeError("weights were already loaded into this AutoencoderKL")
        lib = _lib.lib()
        missing = [k for k in self._shapes if prefix + k not in state_dict]
        if missing:
            raise RuntimeError(f"Error(s) in loading state_dict for AutoencoderKL: missing {missing[:5]}")
        for k, shp in self._shapes.items():
            t = state_dict[prefix + k]
            if tuple(t.shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {tuple(shp)}")
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            shape = (ctypes.c_int64 * len(shp))(*shp)
            check(lib.sdmi_vae_decoder_set_weight(self._h, k.encode(), ptr(t), shape, len(shp), _lib.F32, stream_ptr()),
                  f"set_weight({k})")
            torch.cuda.current_stream().synchronize()
        check(lib.sdmi_vae_decoder_finalize(self._h, stream_ptr()), "vae_decoder_finalize")
        self._finalized = True
        unexpected = [k for k in state_dict if k.startswith(prefix) and k[len(prefix):] not in self._shapes]
        return [], unexpected

    def decode(self, z, _z_scale=1.0):
        """dec = Decoder(post_quant_conv(z)); z (B,4,h,w) fp32 -> (B,3,8h,8w) fp32."""
        if not self._finalized:
            raise RuntimeError("AutoencoderKL.decode called before load_state_dict")
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        b, c, h, w = z.shape
        nup = len(self.ddconfig.get("ch_mult", (1, 2, 4, 8))) - 1
        out = torch.empty((b, self.ddconfig["out_ch"], h << nup, w << nup), dtype=torch.float32, device=self.device)
        check(_lib.lib().sdmi_vae_decoder_forward(self._h, ptr(z), float(_z_scale), ptr(out), b, h, w, stream_ptr()),
              "vae_decoder_forward")
        return out

    def encode(self, x):
        raise NotImplementedError("AutoencoderKL.encode is outside the MI355X denoising path "
                                  "(runs once per image in img2img/inpaint callers)")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().sdmi_vae_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass
