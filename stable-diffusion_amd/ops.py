"""Per-op Python wrappers over the C ABI (device tensors in, device tensors out).

Activations are NHWC / token-major in the compute dtype.  Used by the unit parity tests and
usable on their own; every call goes through libsdmi.so and raises if it is missing.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, dtype_code, torch_dtype


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()
    return t


def convert(x_f32, dtype):
    x_f32 = _f32(x_f32.contiguous())
    out = torch.empty(x_f32.shape, dtype=torch_dtype(dtype), device=x_f32.device)
    check(_lib.lib().sdmi_convert_from_f32(ptr(x_f32), ptr(out), x_f32.numel(), dtype_code(dtype), stream_ptr()),
          "convert_from_f32")
    return out


def to_f32(x):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib.lib().sdmi_convert_to_f32(ptr(x), ptr(out), x.numel(), dtype_code(x.dtype), stream_ptr()),
          "convert_to_f32")
    return out


def pack_weight(w_f32, dtype):
    """(O,I,kh,kw) or (O,I) fp32 -> packed [O][taps*I] in dtype."""
    w_f32 = _f32(w_f32.contiguous())
    O, I = w_f32.shape[0], w_f32.shape[1]
    taps = 1 if w_f32.dim() == 2 else w_f32.shape[2] * w_f32.shape[3]
    out = torch.empty((O, taps * I), dtype=torch_dtype(dtype), device=w_f32.device)
    check(_lib.lib().sdmi_pack_weight(ptr(w_f32), ptr(out), O, I, taps, dtype_code(dtype), stream_ptr()),
          "pack_weight")
    return out


def pack_geglu(w_f32, b_f32, dtype):
    w_f32, b_f32 = _f32(w_f32.contiguous()), _f32(b_f32.contiguous())
    F2, I = w_f32.shape
    wout = torch.empty((F2, I), dtype=torch_dtype(dtype), device=w_f32.device)
    bout = torch.empty((F2,), dtype=torch.float32, device=w_f32.device)
    check(_lib.lib().sdmi_pack_geglu(ptr(w_f32), ptr(b_f32), ptr(wout), ptr(bout), F2 // 2, I, dtype_code(dtype),
                                     stream_ptr()), "pack_geglu")
    return wout, bout


def conv2d(x, w_packed, bias=None, *, taps=9, stride=1, up=False, x1=None, rowadd=None, resid=None,
           geglu=False, splits=1, n_out=None, pad_lo=1):
    """x: NHWC (B,H,W,C0) [+ x1 (B,H,W,C1)] -> NHWC (B,Ho,Wo,N) (N/2 with geglu)."""
    B, H, W, C0 = x.shape
    C1 = x1.shape[3] if x1 is not None else 0
    N = n_out if n_out is not None else w_packed.shape[0]
    Hup, Wup = (2 * H, 2 * W) if up else (H, W)
    if taps == 9:
        Ho, Wo = (Hup + pad_lo + 1 - 3) // stride + 1, (Wup + pad_lo + 1 - 3) // stride + 1
    else:
        Ho, Wo = (Hup - 1) // stride + 1, (Wup - 1) // stride + 1
    No = N // 2 if geglu else N
    out = torch.empty((B, Ho, Wo, No), dtype=x.dtype, device=x.device)
    partial = None
    if splits is None:          # the executor's own choice for this shape
        splits = _lib.lib().sdmi_conv2d_splits(B * Ho * Wo, N, taps * (C0 + C1), dtype_code(x.dtype))
    if splits > 1:
        partial = torch.empty((splits, B * Ho * Wo, N), dtype=torch.float32, device=x.device)
    check(_lib.lib().sdmi_conv2d(ptr(x), C0, ptr(x1), C1, ptr(w_packed), ptr(bias), ptr(rowadd), ptr(resid),
                                 resid.shape[-1] if resid is not None else 0, ptr(out), No, B, H, W, taps, stride,
                                 pad_lo, 1 if up else 0, N, 1 if geglu else 0, splits, ptr(partial), dtype_code(x.dtype),
                                 stream_ptr()), "conv2d")
    return out


def conv_in(sources, w, bias, dtype):
    """First convolution: fp32 NCHW sources (channel-concatenated) * OIHW fp32 weights -> NHWC in ``dtype``."""
    import ctypes
    sources = [_f32(s.contiguous()) for s in sources]
    w, bias = _f32(w.contiguous()), _f32(bias.contiguous())
    b, _, h, wd = sources[0].shape
    cout = w.shape[0]
    assert w.shape[1] == sum(s.shape[1] for s in sources) and tuple(w.shape[2:]) == (3, 3)
    out = torch.empty((b, h, wd, cout), dtype=torch_dtype(dtype), device=w.device)
    wpack = torch.empty((w.numel(),), dtype=torch.float32, device=w.device)
    n = len(sources)
    ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in sources])
    chans = (ctypes.c_int * n)(*[s.shape[1] for s in sources])
    check(_lib.lib().sdmi_conv_in(ptrs, chans, n, ptr(w), ptr(bias), ptr(wpack), ptr(out), b, h, wd, cout,
                                  dtype_code(dtype), stream_ptr()), "conv_in")
    return out


def conv_out(x, w, bias):
    """Last convolution: NHWC activations * OIHW fp32 weights -> fp32 NCHW."""
    x = x.contiguous()
    w, bias = _f32(w.contiguous()), _f32(bias.contiguous())
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    assert w.shape[1] == cin and tuple(w.shape[2:]) == (3, 3)
    out = torch.empty((b, cout, h, wd), dtype=torch.float32, device=x.device)
    wpack = torch.empty((w.numel(),), dtype=torch.float32, device=x.device)
    check(_lib.lib().sdmi_conv_out(ptr(x), ptr(w), ptr(bias), ptr(wpack), ptr(out), b, h, wd, cin, cout,
                                   dtype_code(x.dtype), stream_ptr()), "conv_out")
    return out


def linear(x, w_packed, bias=None, resid=None, geglu=False, splits=1):
    """x: (rows, K) -> (rows, N)."""
    rows, K = x.shape
    y = conv2d(x.view(1, 1, rows, K), w_packed, bias, taps=1, resid=resid, geglu=geglu, splits=splits)
    return y.view(rows, -1)


def groupnorm(x, gamma, beta, eps, silu=False, x1=None, groups=32):
    """x: NHWC (B,H,W,C0) [+ x1] -> NHWC (B,H,W,C0+C1)."""
    B, H, W, C0 = x.shape
    C1 = x1.shape[3] if x1 is not None else 0
    out = torch.empty((B, H, W, C0 + C1), dtype=x.dtype, device=x.device)
    nbytes = _lib.lib().sdmi_groupnorm_scratch_bytes(B, groups)
    scratch = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    check(_lib.lib().sdmi_groupnorm(ptr(x), C0, ptr(x1), C1, B, H * W, groups, eps, ptr(_f32(gamma)), ptr(_f32(beta)),
                                    1 if silu else 0, ptr(out), ptr(scratch), dtype_code(x.dtype), stream_ptr()),
          "groupnorm")
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    rows, d = x.shape
    out = torch.empty_like(x)
    check(_lib.lib().sdmi_layernorm(ptr(x), ptr(out), ptr(_f32(gamma)), ptr(_f32(beta)), rows, d, eps,
                                    dtype_code(x.dtype), stream_ptr()), "layernorm")
    return out


def layernorm_linear(x, w, bias, gamma, beta, eps=1e-5, geglu=False, splits=1):
    """Linear(LayerNorm(x)) with the normalisation folded into the GEMM.  x: (rows, K) in the compute dtype; w:
    fp32 (N, K) as in the checkpoint; returns (rows, N), or (rows, N/2) = value * gelu(gate) with ``geglu``."""
    x = x.contiguous()
    rows, K = x.shape
    w = _f32(w.contiguous())
    N = w.shape[0]
    if splits is None:
        splits = _lib.lib().sdmi_conv2d_splits(rows, N, K, dtype_code(x.dtype))
    out = torch.empty((rows, N // 2 if geglu else N), dtype=x.dtype, device=x.device)
    nbytes = _lib.lib().sdmi_layernorm_linear_workspace_bytes(rows, K, N, splits, dtype_code(x.dtype))
    wsp = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    check(_lib.lib().sdmi_layernorm_linear(ptr(x), ptr(w), ptr(bias), ptr(_f32(gamma)), ptr(_f32(beta)), eps, ptr(out),
                                           rows, K, N, 1 if geglu else 0, splits, ptr(wsp), dtype_code(x.dtype),
                                           stream_ptr()), "layernorm_linear")
    return out


def attention(q, k, v, heads, scale=None, impl=0, causal=False):
    """q: (B,Nq,heads*d), k/v: (B,Nk,heads*d) (may be strided views over the last dim) -> (B,Nq,heads*d)."""
    impl = (impl & 1) | (2 if causal else 0)
    B, Nq, C = q.shape
    Nk = k.shape[1]
    d = C // heads
    scale = d ** -0.5 if scale is None else scale
    out = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
    for t in (q, k, v):
        assert t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1)
    check(_lib.lib().sdmi_attention(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(out), C, B,
                                    heads, Nq, Nk, d, scale, dtype_code(q.dtype), impl, stream_ptr()), "attention")
    return out


def timestep_embedding(t, dim, max_period=10000.0):
    assert t.dtype == torch.int64 and t.is_cuda
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    check(_lib.lib().sdmi_timestep_embedding(ptr(t), ptr(out), t.shape[0], dim, max_period, stream_ptr()),
          "timestep_embedding")
    return out


def small_linear(x, w, b=None, act_in=0, act_out=0):
    x, w = _f32(x.contiguous()), _f32(w.contiguous())
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(_lib.lib().sdmi_small_linear(ptr(x), ptr(w), ptr(b), ptr(y), M, K, N, act_in, act_out, stream_ptr()),
          "small_linear")
    return y


def ddim_step(x, eps, cfg, scale, a_t, a_prev, sigma, sqrt_1m_at, noise=None, temperature=1.0, want_pred_x0=True):
    x, eps = _f32(x), _f32(eps)
    x_prev = torch.empty_like(x)
    pred = torch.empty_like(x) if want_pred_x0 else None
    check(_lib.lib().sdmi_ddim_step(ptr(x), ptr(eps), 1 if cfg else 0, float(scale), float(a_t), float(a_prev),
                                    float(sigma), float(sqrt_1m_at), ptr(noise), float(temperature), ptr(x_prev),
                                    ptr(pred), x.numel(), stream_ptr()), "ddim_step")
    return x_prev, pred


def ddpm_step(x, eps, sqrt_recip_acp, sqrt_recipm1_acp, coef1, coef2, noise_coef, noise=None, temperature=1.0,
              clip_denoised=False, want_pred_x0=True, out=None):
    x, eps = _f32(x), _f32(eps)
    x_prev = torch.empty_like(x) if out is None else out
    pred = torch.empty_like(x) if want_pred_x0 else None
    check(_lib.lib().sdmi_ddpm_step(ptr(x), ptr(eps), float(sqrt_recip_acp), float(sqrt_recipm1_acp),
                                    1 if clip_denoised else 0, float(coef1), float(coef2), float(noise_coef),
                                    ptr(noise), float(temperature), ptr(x_prev), ptr(pred), x.numel(),
                                    stream_ptr()), "ddpm_step")
    return x_prev, pred


def cfg_combine(eps2, scale, out=None):
    """e_uncond + scale * (e_cond - e_uncond) for eps2 = cat([e_uncond, e_cond]) on the batch axis."""
    eps2 = _f32(eps2)
    b = eps2.shape[0] // 2
    out = torch.empty_like(eps2[:b]) if out is None else _f32(out)
    check(_lib.lib().sdmi_cfg_combine(ptr(eps2), float(scale), ptr(out), out.numel(), stream_ptr()), "cfg_combine")
    return out


def axpby(a, b, ca, cb, out=None):
    a, b = _f32(a), _f32(b)
    out = torch.empty_like(a) if out is None else _f32(out)
    check(_lib.lib().sdmi_axpby(ptr(a), ptr(b), float(ca), float(cb), ptr(out), a.numel(), stream_ptr()), "axpby")
    return out


def noise_to_timesteps(x0, noise, t, table_a, table_b):
    """table_a[t] * x0 + table_b[t] * noise with per-sample table indices ``t`` (extract_into_tensor semantics,
    stable_diffusion.py:3600-3603: a one-element ``t`` serves every sample).  One launch per run of equal
    indices, so the usual case of one timestep for the whole batch is a single launch."""
    x0, noise = _f32(x0), _f32(noise)
    b = x0.shape[0]
    t = torch.as_tensor(t).reshape(-1).cpu().long()
    if t.numel() == 1:
        t = t.expand(b)
    assert t.numel() == b, f"{t.numel()} timesteps for a batch of {b}"
    out = torch.empty_like(x0)
    lo = 0
    while lo < b:
        hi = lo + 1
        while hi < b and t[hi] == t[lo]:
            hi += 1
        axpby(x0[lo:hi], noise[lo:hi], float(table_a[t[lo]]), float(table_b[t[lo]]), out=out[lo:hi])
        lo = hi
    return out


def q_sample_blend(x0, noise, mask, img, sqrt_acp, sqrt_one_minus_acp, out=None):
    """(sqrt_acp * x0 + sqrt_one_minus_acp * noise) * mask + (1 - mask) * img: the DDIM loop's known-region blend
    (one timestep for the batch); mask (B,1,H,W) or (B,C,H,W).  ``out`` may be ``img``."""
    x0, noise, mask, img = _f32(x0), _f32(noise), _f32(mask), _f32(img)
    b, c, h, w = x0.shape
    assert noise.shape == x0.shape and img.shape == x0.shape
    assert mask.dim() == 4 and mask.shape[0] == b and tuple(mask.shape[2:]) == (h, w), f"mask shape {tuple(mask.shape)}"
    out = torch.empty_like(x0) if out is None else _f32(out)
    check(_lib.lib().sdmi_q_sample_blend(ptr(x0), ptr(noise), ptr(mask), ptr(img), float(sqrt_acp),
                                         float(sqrt_one_minus_acp), ptr(out), b, c, h * w, mask.shape[1],
                                         stream_ptr()), "q_sample_blend")
    return out


def to_image(x):
    x = _f32(x)
    out = torch.empty_like(x)
    check(_lib.lib().sdmi_to_image(ptr(x), ptr(out), x.numel(), stream_ptr()), "to_image")
    return out


def to_image_u8(x):
    """x NCHW fp32 in [-1, 1] -> uint8 NHWC, the array the reference hands to Image.fromarray."""
    x = _f32(x)
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), dtype=torch.uint8, device=x.device)
    check(_lib.lib().sdmi_to_image_u8(ptr(x), ptr(out), b, c, h * w, stream_ptr()), "to_image_u8")
    return out


def inpaint_composite(image, mask, predicted):
    """(1 - mask) * clamp((image + 1) / 2, 0, 1) + mask * predicted; mask (B,1,H,W)."""
    image, mask, predicted = _f32(image), _f32(mask), _f32(predicted)
    b, c, h, w = image.shape
    assert tuple(mask.shape) == (b, 1, h, w) and predicted.shape == image.shape
    out = torch.empty_like(image)
    check(_lib.lib().sdmi_inpaint_composite(ptr(image), ptr(mask), ptr(predicted), ptr(out), b, c, h * w, stream_ptr()),
          "inpaint_composite")
    return out
