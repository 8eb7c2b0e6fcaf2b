"""ctypes binding of libsdmi.so (the C ABI declared in include/sdmi.h).

The product has no CPU fallback: if the shared library is missing or a call fails, this module
raises.  ``build()`` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDMI_LIB") or os.path.join(_HERE, "libsdmi.so")
CSRC = os.path.join(_HERE, "csrc")

c_void_p, c_int, c_float, c_int64, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_int64, ctypes.c_char_p)
P = ctypes.POINTER

F32, F16, BF16 = 0, 1, 2


class UNetCfg(ctypes.Structure):
    _fields_ = [("in_channels", c_int), ("model_channels", c_int), ("out_channels", c_int),
                ("num_res_blocks", c_int), ("n_channel_mult", c_int), ("channel_mult", c_int * 8),
                ("n_attention_resolutions", c_int), ("attention_resolutions", c_int * 8),
                ("num_heads", c_int), ("num_head_channels", c_int), ("transformer_depth", c_int),
                ("context_dim", c_int), ("legacy", c_int), ("compute_dtype", c_int)]


class VaeCfg(ctypes.Structure):
    _fields_ = [("z_channels", c_int), ("embed_dim", c_int), ("ch", c_int), ("out_ch", c_int),
                ("n_ch_mult", c_int), ("ch_mult", c_int * 8), ("num_res_blocks", c_int),
                ("compute_dtype", c_int)]


class ClipCfg(ctypes.Structure):
    _fields_ = [("vocab_size", c_int), ("hidden", c_int), ("intermediate", c_int), ("layers", c_int),
                ("heads", c_int), ("max_positions", c_int), ("ln_eps", c_float), ("compute_dtype", c_int)]


# name -> (restype, argtypes); every symbol of include/sdmi.h
SIGNATURES = {
    "sdmi_version": (c_int, []),
    "sdmi_last_error": (c_char_p, []),
    "sdmi_unet_create": (c_int, [P(UNetCfg), P(c_void_p)]),
    "sdmi_unet_num_weights": (c_int, [c_void_p]),
    "sdmi_unet_weight_info": (c_int, [c_void_p, c_int, P(c_char_p), P(c_int64), P(c_int)]),
    "sdmi_unet_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int, c_int, c_void_p]),
    "sdmi_unet_finalize": (c_int, [c_void_p, c_void_p]),
    "sdmi_unet_set_context": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sdmi_unet_forward": (c_int, [c_void_p, c_void_p, P(c_void_p), P(c_int), c_int, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_void_p]),
    "sdmi_unet_run_block": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                    c_int, c_void_p]),
    "sdmi_unet_device_bytes": (c_int64, [c_void_p]),
    "sdmi_unet_set_graph_mode": (c_int, [c_void_p, c_int]),
    "sdmi_unet_graph_launches": (c_int, [c_void_p]),
    "sdmi_unet_set_uniform_timesteps": (c_int, [c_void_p, c_int]),
    "sdmi_unet_set_cfg_duplicate": (c_int, [c_void_p, c_int]),
    "sdmi_unet_destroy": (None, [c_void_p]),
    "sdmi_ddim_step": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float,
                               c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "sdmi_ddpm_step": (c_int, [c_void_p, c_void_p, c_float, c_float, c_int, c_float, c_float, c_float, c_void_p,
                               c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    "sdmi_cfg_combine": (c_int, [c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "sdmi_axpby": (c_int, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_int64, c_void_p]),
    "sdmi_q_sample_blend": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_int, c_int,
                                    c_int, c_int, c_void_p]),
    "sdmi_vae_decoder_create": (c_int, [P(VaeCfg), P(c_void_p)]),
    "sdmi_vae_decoder_num_weights": (c_int, [c_void_p]),
    "sdmi_vae_decoder_weight_info": (c_int, [c_void_p, c_int, P(c_char_p), P(c_int64), P(c_int)]),
    "sdmi_vae_decoder_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int, c_int, c_void_p]),
    "sdmi_vae_decoder_finalize": (c_int, [c_void_p, c_void_p]),
    "sdmi_vae_decoder_forward": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sdmi_vae_decoder_destroy": (None, [c_void_p]),
    "sdmi_clip_text_create": (c_int, [P(ClipCfg), P(c_void_p)]),
    "sdmi_clip_text_num_weights": (c_int, [c_void_p]),
    "sdmi_clip_text_weight_info": (c_int, [c_void_p, c_int, P(c_char_p), P(c_int64), P(c_int)]),
    "sdmi_clip_text_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int, c_int, c_void_p]),
    "sdmi_clip_text_finalize": (c_int, [c_void_p, c_void_p]),
    "sdmi_clip_text_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sdmi_clip_text_destroy": (None, [c_void_p]),
    "sdmi_vae_encoder_create": (c_int, [P(VaeCfg), P(c_void_p)]),
    "sdmi_vae_encoder_num_weights": (c_int, [c_void_p]),
    "sdmi_vae_encoder_weight_info": (c_int, [c_void_p, c_int, P(c_char_p), P(c_int64), P(c_int)]),
    "sdmi_vae_encoder_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, P(c_int64), c_int, c_int, c_void_p]),
    "sdmi_vae_encoder_finalize": (c_int, [c_void_p, c_void_p]),
    "sdmi_vae_encoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sdmi_vae_encoder_destroy": (None, [c_void_p]),
    "sdmi_gaussian_sample": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sdmi_to_image": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sdmi_to_image_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sdmi_inpaint_composite": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sdmi_groupnorm_scratch_bytes": (c_int64, [c_int, c_int]),
    "sdmi_groupnorm": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                               c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "sdmi_groupnorm_static": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                      c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "sdmi_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "sdmi_layernorm_linear_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "sdmi_layernorm_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sdmi_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sdmi_pack_geglu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sdmi_conv2d": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                            c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_void_p, c_int, c_void_p]),
    "sdmi_conv_in": (c_int, [P(c_void_p), P(c_int), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_int, c_int, c_void_p]),
    "sdmi_conv_out": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                              c_int, c_void_p]),
    "sdmi_conv2d_splits": (c_int, [c_int, c_int, c_int, c_int]),
    "sdmi_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "sdmi_timestep_embedding": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "sdmi_small_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "sdmi_convert_from_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sdmi_convert_to_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sdmi_prof_enable": (c_int, [c_int]),
    "sdmi_prof_reset": (c_int, []),
    "sdmi_prof_dump": (c_int, [c_char_p]),
    "sdmi_prof_calibrate": (c_int, [c_int, c_void_p, P(ctypes.c_double)]),
    "sdmi_prof_get": (c_int, [c_char_p, P(ctypes.c_double), P(c_int64), P(ctypes.c_double), P(ctypes.c_double)]),
}

_lib = None


def build(verbose=False, jobs=8):
    """Compile libsdmi.so in-tree (hipcc, --offload-arch=gfx950)."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libsdmi.so failed (see output above)")
    return LIB_PATH


def lib():
    """The loaded library with typed entry points; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C {CSRC}` "
                               "(there is no CPU fallback for the device path)")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("SDMI_LIB_ALLOW_MISSING") == "1" and not hasattr(l, name):
                continue                   # (A/B runs against an older build named by SDMI_LIB)
            fn = getattr(l, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code, what=""):
    if code != 0:
        msg = lib().sdmi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libsdmi {what} failed ({code}): {msg}")


def stream_ptr():
    """The current torch HIP stream as a void* for the C ABI."""
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def dtype_code(dt):
    import torch
    if isinstance(dt, int):
        return dt
    if isinstance(dt, str):
        return {"fp32": F32, "float32": F32, "fp16": F16, "float16": F16, "half": F16,
                "bf16": BF16, "bfloat16": BF16}[dt]
    return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[dt]


def torch_dtype(code):
    import torch
    return {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16}[dtype_code(code)]
