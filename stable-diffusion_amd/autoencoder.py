"""Host-side mirror of the reference's first stage on top of libsdmi.

``AutoencoderKL(ddconfig, lossconfig, embed_dim)`` keeps the reference's names and checkpoint keys
(stable_diffusion.py:1370-1419): ``decode(z)`` (Decoder :422-527) and ``encode(x)`` (Encoder :326-420)
returning a ``DiagonalGaussianDistribution`` (:624-657).  The encoder half is built when its weights
(``encoder.*``, ``quant_conv.*``) are present in the state dict.
"""
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, dtype_code
from . import structure as S


def _make_cfg(ddconfig: dict, embed_dim: int, compute_dtype):
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
    return cfg


def _create_handle(ddconfig: dict, embed_dim: int, compute_dtype="fp16"):
    S.vae_decoder_structure(ddconfig, embed_dim)        # validates unsupported options loudly
    cfg = _make_cfg(ddconfig, embed_dim, compute_dtype)
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


class DiagonalGaussianDistribution:
    """Posterior of the VAE encoder given by its moments tensor (stable_diffusion.py:624-657).

    ``sample()`` and ``mode()`` run the fused device kernel; ``mean`` / ``logvar`` / ``std`` / ``var`` are
    views and elementwise tensors for callers that inspect them."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.deterministic = deterministic

    @property
    def mean(self):
        return torch.chunk(self.parameters, 2, dim=1)[0]

    @property
    def logvar(self):
        return torch.clamp(torch.chunk(self.parameters, 2, dim=1)[1], -30.0, 20.0)

    @property
    def std(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(self.logvar)

    def _draw(self, noise, scale):
        p = self.parameters.contiguous()
        b, c2, h, w = p.shape
        z = torch.empty((b, c2 // 2, h, w), dtype=torch.float32, device=p.device)
        check(_lib.lib().sdmi_gaussian_sample(ptr(p), ptr(noise), float(scale), ptr(z), b, c2 // 2, h * w, stream_ptr()),
              "gaussian_sample")
        return z

    def sample(self, noise=None, _scale=1.0):
        """mean + std * randn (:636-638); ``noise`` may be injected (parity tests)."""
        if self.deterministic:
            return self._draw(None, _scale)
        if noise is None:
            noise = torch.randn(self.mean.shape, device=self.parameters.device)
        noise = noise.to(device=self.parameters.device, dtype=torch.float32).contiguous()
        return self._draw(noise, _scale)

    def mode(self, _scale=1.0):
        return self._draw(None, _scale)


class AutoencoderKL:
    """Drop-in for stable_diffusion.AutoencoderKL on the decode path."""

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, compute_dtype=None, device=None):
        assert ddconfig["double_z"]
        if ckpt_path is not None:
            raise NotImplementedError("pass tensors through load_state_dict (no checkpoint files offline)")
        self.ddconfig = dict(ddconfig)
        self.embed_dim = embed_dim
        # the reference's autoencoder computes in fp32; the 16-bit fast paths are opt-in
        compute_dtype = compute_dtype or "fp32"
        self.compute_dtype = compute_dtype
        self.device = torch.device(device if device is not None else "cuda")
        self._h = _create_handle(self.ddconfig, embed_dim, compute_dtype)
        self._shapes = _handle_weight_shapes(self._h)
        self._finalized = False
        self._he = None                      # encoder handle: created when encoder weights are loaded
        self._enc_shapes = S.vae_encoder_param_shapes(S.vae_encoder_structure(self.ddconfig, embed_dim))

    def eval(self):
        return self

    def state_dict_shapes(self, encoder=False):
        """Checkpoint keys and shapes of decode(); with ``encoder=True`` those of encode() as well."""
        out = OrderedDict(self._shapes)
        if encoder:
            out.update(self._enc_shapes)
        return out

    def load_state_dict(self, state_dict, strict=True, prefix=""):
        """Decoder-side keys (post_quant_conv.*, decoder.*) are required; the encoder half (encoder.*,
        quant_conv.*) is built when all of its keys are present."""
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
        have_enc = [k for k in self._enc_shapes if prefix + k in state_dict]
        if have_enc:
            if len(have_enc) != len(self._enc_shapes):
                miss = [k for k in self._enc_shapes if prefix + k not in state_dict]
                raise RuntimeError(f"Error(s) in loading state_dict for AutoencoderKL: missing {miss[:5]}")
            self._load_encoder(state_dict, prefix)
        known = set(self._shapes) | set(self._enc_shapes)
        unexpected = [k for k in state_dict if k.startswith(prefix) and k[len(prefix):] not in known]
        return [], unexpected

    def _load_encoder(self, state_dict, prefix):
        lib = _lib.lib()
        cfg = _make_cfg(self.ddconfig, self.embed_dim, self.compute_dtype)
        h = ctypes.c_void_p()
        check(lib.sdmi_vae_encoder_create(ctypes.byref(cfg), ctypes.byref(h)), "vae_encoder_create")
        key = ctypes.c_char_p()
        shape4 = (ctypes.c_int64 * 4)()
        nd = ctypes.c_int()
        for i in range(lib.sdmi_vae_encoder_num_weights(h)):
            check(lib.sdmi_vae_encoder_weight_info(h, i, ctypes.byref(key), shape4, ctypes.byref(nd)), "weight_info")
            k = key.value.decode()
            shp = tuple(shape4[j] for j in range(nd.value))
            t = state_dict[prefix + k]
            if tuple(t.shape) != shp:
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {shp}")
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            shape = (ctypes.c_int64 * len(shp))(*shp)
            check(lib.sdmi_vae_encoder_set_weight(h, k.encode(), ptr(t), shape, len(shp), _lib.F32, stream_ptr()),
                  f"set_weight({k})")
            torch.cuda.current_stream().synchronize()
        check(lib.sdmi_vae_encoder_finalize(h, stream_ptr()), "vae_encoder_finalize")
        self._he = h

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
        """posterior = DiagonalGaussianDistribution(quant_conv(Encoder(x))) (:1410-1414); x (B,3,H,W) fp32."""
        if self._he is None:
            raise RuntimeError("AutoencoderKL.encode: no encoder weights (encoder.*, quant_conv.*) were loaded")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        b, c, h, w = x.shape
        ndown = len(self.ddconfig.get("ch_mult", (1, 2, 4, 8))) - 1
        moments = torch.empty((b, 2 * self.embed_dim, h >> ndown, w >> ndown), dtype=torch.float32, device=self.device)
        check(_lib.lib().sdmi_vae_encoder_forward(self._he, ptr(x), ptr(moments), b, h, w, stream_ptr()),
              "vae_encoder_forward")
        return DiagonalGaussianDistribution(moments)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().sdmi_vae_decoder_destroy(self._h)
                self._h = None
            if getattr(self, "_he", None):
                _lib.lib().sdmi_vae_encoder_destroy(self._he)
                self._he = None
        except Exception:
            pass
