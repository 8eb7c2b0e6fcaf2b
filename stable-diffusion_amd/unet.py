"""Host-side mirror of the reference's denoiser classes on top of libsdmi.

``UNetModel`` and ``DiffusionWrapper`` keep the reference's constructor kwargs, ``forward``
signatures, checkpoint key layout and assertion behaviour (stable_diffusion.py:3263-3552,
:3635-3664); all arithmetic runs in hand-written gfx950 kernels behind the C ABI.  There is no CPU
path: constructing a model on a machine without the library or without a GPU raises.
"""
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, dtype_code
from . import structure as S


def _create_handle(unet_config: dict, compute_dtype="fp16"):
    st = S.unet_structure(**unet_config)          # validates the kwargs like the reference's asserts
    cfg = _lib.UNetCfg()
    cfg.in_channels = st.in_channels
    cfg.model_channels = st.model_channels
    cfg.out_channels = st.out_channels
    cfg.num_res_blocks = unet_config["num_res_blocks"]
    mult = list(unet_config.get("channel_mult", (1, 2, 4, 8)))
    cfg.n_channel_mult = len(mult)
    for i, m in enumerate(mult):
        cfg.channel_mult[i] = m
    ar = list(unet_config["attention_resolutions"])
    cfg.n_attention_resolutions = len(ar)
    for i, a in enumerate(ar):
        cfg.attention_resolutions[i] = a
    cfg.num_heads = unet_config.get("num_heads", -1)
    cfg.num_head_channels = unet_config.get("num_head_channels", -1)
    cfg.transformer_depth = unet_config.get("transformer_depth", 1)
    cfg.context_dim = unet_config["context_dim"]
    cfg.legacy = 1 if unet_config.get("legacy", True) else 0
    cfg.compute_dtype = dtype_code(compute_dtype)
    h = ctypes.c_void_p()
    check(_lib.lib().sdmi_unet_create(ctypes.byref(cfg), ctypes.byref(h)), "unet_create")
    return h


def _handle_weight_shapes(h):
    lib = _lib.lib()
    out = OrderedDict()
    key = ctypes.c_char_p()
    shape = (ctypes.c_int64 * 4)()
    nd = ctypes.c_int()
    for i in range(lib.sdmi_unet_num_weights(h)):
        check(lib.sdmi_unet_weight_info(h, i, ctypes.byref(key), shape, ctypes.byref(nd)), "weight_info")
        out[key.value.decode()] = tuple(shape[j] for j in range(nd.value))
    return out


class UNetModel:
    """Drop-in for stable_diffusion.UNetModel (inference only).

    Same kwargs as stable_diffusion.py:3263-3286; one extra, ``compute_dtype``: "fp16" / "bf16" (MFMA
    operands in 16 bits, fp32 accumulation: the fast path, within 2e-2 of the reference per call) or "fp32"
    (parity mode, within 1e-4).  When it is not given the reference's own switch decides, as it does there
    (:3302: ``self.dtype = th.float16 if use_fp16 else th.float32``): ``use_fp16=True`` selects "fp16", the
    default ``use_fp16=False`` selects "fp32", so a caller that swaps the class in keeps fp32 numerics until
    it asks for the fast path.
    Weights arrive through ``load_state_dict`` with the reference's keys and layouts.
    """

    def __init__(self, image_size=None, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 8), num_heads=-1, num_head_channels=-1,
                 dropout=0., conv_resampler=True, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, num_classes=None, use_checkpoint=False, n_embed=None, legacy=True,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False, dims=2,
                 use_fp16=False, compute_dtype=None, device=None):
        self.config = dict(image_size=image_size, in_channels=in_channels, model_channels=model_channels,
                           out_channels=out_channels, num_res_blocks=num_res_blocks,
                           attention_resolutions=attention_resolutions, channel_mult=channel_mult,
                           num_heads=num_heads, num_head_channels=num_head_channels, dropout=dropout,
                           conv_resampler=conv_resampler, use_spatial_transformer=use_spatial_transformer,
                           transformer_depth=transformer_depth, context_dim=context_dim, num_classes=num_classes,
                           use_checkpoint=use_checkpoint, n_embed=n_embed, legacy=legacy,
                           use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                           use_new_attention_order=use_new_attention_order, dims=dims, use_fp16=use_fp16)
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_classes = num_classes
        self.compute_dtype = compute_dtype or ("fp16" if use_fp16 else "fp32")
        self.dtype = torch.float32          # boundary dtype: x and eps are fp32 NCHW like the reference
        self.device = torch.device(device if device is not None else "cuda")
        self._h = _create_handle(self.config, self.compute_dtype)
        self._shapes = _handle_weight_shapes(self._h)
        self._finalized = False
        self._ctx_ref, self._ctx_version = None, -1
        self._keep = None

    # ---- weights -------------------------------------------------------------------------------
    def state_dict_shapes(self):
        return OrderedDict(self._shapes)

    def load_state_dict(self, state_dict, strict=True, prefix=""):
        """Copy the reference-layout tensors into the handle and repack them (once)."""
        if self._finalized:
            raise RuntimeError("weights were already loaded into this UNetModel")
        lib = _lib.lib()
        missing = [k for k in self._shapes if prefix + k not in state_dict]
        unexpected = [k for k in state_dict if k.startswith(prefix) and k[len(prefix):] not in self._shapes]
        if prefix == "":
            pass
        else:
            unexpected = [k for k in unexpected]     # keys outside the prefix belong to other sub-models
        if missing or (strict and unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for UNetModel: missing {missing[:5]}"
                               f"{'...' if len(missing) > 5 else ''} unexpected {unexpected[:5]}")
        for k, shp in self._shapes.items():
            t = state_dict[prefix + k]
            if tuple(t.shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {tuple(shp)}")
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            shape = (ctypes.c_int64 * len(shp))(*shp)
            check(lib.sdmi_unet_set_weight(self._h, k.encode(), ptr(t), shape, len(shp), _lib.F32, stream_ptr()),
                  f"set_weight({k})")
            torch.cuda.current_stream().synchronize()      # t may be a temporary
        check(lib.sdmi_unet_finalize(self._h, stream_ptr()), "unet_finalize")
        self._finalized = True
        return [], unexpected

    # ---- per-prompt context ----------------------------------------------------------------------
    def set_context(self, context):
        """Project the text context through every cross-attention's to_k/to_v once (cached in the
        handle).  Called implicitly by forward() when it sees a new context tensor."""
        ctx = context.detach().to(device=self.device, dtype=torch.float32).contiguous()
        b, n, d = ctx.shape
        check(_lib.lib().sdmi_unet_set_context(self._h, ptr(ctx), b, n, stream_ptr()), "set_context")
        # holding the caller's tensor keeps its storage from being recycled under the same address
        self._ctx_ref, self._ctx_version = context, context._version

    def set_uniform_timesteps(self, enable: bool):
        """Promise that all samples of the following forwards share one timestep (sampler loops do)."""
        check(_lib.lib().sdmi_unet_set_uniform_timesteps(self._h, 1 if enable else 0), "set_uniform_timesteps")

    def set_cfg_duplicate(self, enable: bool):
        """Promise that samples b and b + B/2 of the following forwards share x and t (CFG pairs)."""
        check(_lib.lib().sdmi_unet_set_cfg_duplicate(self._h, 1 if enable else 0), "set_cfg_duplicate")

    def set_graph_mode(self, enable: bool):
        check(_lib.lib().sdmi_unet_set_graph_mode(self._h, 1 if enable else 0), "set_graph_mode")

    def graph_launches(self):
        """Kernel launches of the forward graph currently recorded (0 if the last forward ran eagerly)."""
        return int(_lib.lib().sdmi_unet_graph_launches(self._h))

    def device_bytes(self):
        return int(_lib.lib().sdmi_unet_device_bytes(self._h))

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, x, timesteps=None, context=None, y=None, out=None, _concat=None, **kwargs):
        """eps = UNet(x, timesteps, context).  x: (N,C,H,W) fp32; timesteps: (N,) int64; context:
        (N,77,context_dim).  ``out`` optionally names the result buffer (keeps pointers stable for
        graph replay)."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if not self._finalized:
            raise RuntimeError("UNetModel.forward called before load_state_dict")
        if context is None:
            raise RuntimeError("the spatial-transformer UNet needs a context tensor")
        if context is not self._ctx_ref or context._version != self._ctx_version:
            self.set_context(context)
        x = x.to(device=self.device, dtype=torch.float32)
        if not x.is_contiguous():
            x = x.contiguous()
        t = timesteps.to(device=self.device, dtype=torch.int64).contiguous()
        b, cx, h, w = x.shape
        concat = [c.to(device=self.device, dtype=torch.float32).contiguous() for c in (_concat or [])]
        assert cx + sum(c.shape[1] for c in concat) == self.in_channels, \
            f"expected {self.in_channels} input channels, got {cx + sum(c.shape[1] for c in concat)}"
        if out is None:
            out = torch.empty((b, self.out_channels, h, w), dtype=torch.float32, device=self.device)
        n = len(concat)
        cptrs = (ctypes.c_void_p * max(n, 1))(*[c.data_ptr() for c in concat]) if n else None
        cch = (ctypes.c_int * max(n, 1))(*[c.shape[1] for c in concat]) if n else None
        check(_lib.lib().sdmi_unet_forward(self._h, ptr(x), cptrs, cch, n, ptr(t), ptr(out), b, h, w, stream_ptr()),
              "unet_forward")
        self._keep = (x, t, concat)       # keep inputs alive until the stream has consumed them
        return out

    __call__ = forward

    def run_block(self, kind, index, x, emb=None, skip=None, context=None):
        """One ResBlock (``kind`` "res": ResBlock.forward(x, emb), stable_diffusion.py:3197) or SpatialTransformer
        ("st": forward(x, context), :2885) of this model on its own, through the executor's code for it.  x / skip:
        NHWC tensors in the compute dtype; emb: fp32 (B, 4 * model_channels); returns NHWC."""
        if not self._finalized:
            raise RuntimeError("UNetModel.run_block called before load_state_dict")
        if context is not None and (context is not self._ctx_ref or context._version != self._ctx_version):
            self.set_context(context)
        b, h, w, _ = x.shape
        shapes = list(self._shapes.items())
        if kind == "res":
            names = [k for k, _ in shapes if k.endswith(".in_layers.2.weight")]
            cout = self._shapes[names[index]][0]
        else:
            names = [k for k, _ in shapes if k.endswith(".proj_out.weight")]
            cout = self._shapes[names[index]][0]
        x = x.contiguous()
        out = torch.empty((b, h, w, cout), dtype=x.dtype, device=x.device)
        skip = None if skip is None else skip.contiguous()
        emb = None if emb is None else emb.to(self.device, torch.float32).contiguous()
        check(_lib.lib().sdmi_unet_run_block(self._h, 0 if kind == "res" else 1, index, ptr(x), ptr(skip),
                                             0 if skip is None else skip.shape[-1], ptr(emb), ptr(out), b, h, w,
                                             stream_ptr()), "unet_run_block")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().sdmi_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass


class DiffusionWrapper:
    """Drop-in for stable_diffusion.DiffusionWrapper (:3635-3664): routes the conditioning."""

    def __init__(self, unet_model_config, conditioning_key, compute_dtype=None, device=None):
        self.diffusion_model = UNetModel(**unet_model_config, compute_dtype=compute_dtype, device=device)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, 'concat', 'crossattn', 'hybrid', 'adm']

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, out=None):
        if self.conditioning_key == 'crossattn':
            cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
            return self.diffusion_model(x, t, context=cc, out=out)
        if self.conditioning_key == 'hybrid':
            # xc = cat([x] + c_concat[0] + c_concat[1], dim=1): the concat is folded into the first
            # convolution's loader instead of being materialised
            cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
            extra = list(c_concat[0]) + list(c_concat[1])
            return self.diffusion_model(x, t, context=cc, out=out, _concat=extra)
        if self.conditioning_key in (None, 'concat', 'adm'):
            raise NotImplementedError(
                f"conditioning_key={self.conditioning_key!r}: only the spatial-transformer UNet "
                "('crossattn', 'hybrid') is on the MI355X path")
        raise NotImplementedError()

    __call__ = forward
