"""fp32 torch-CPU restatement of the reference VAE decode and encode paths (TEST INFRASTRUCTURE ONLY).

State-dict keys are relative to ``first_stage_model.``; file cited = /root/reference/stable_diffusion.py.
"""
import importlib

import torch
import torch.nn.functional as F

_S = importlib.import_module("stable-diffusion_amd.structure")


def swish(x):
    """nonlinearity :73-74."""
    return x * torch.sigmoid(x)


def _gn(x, sd, p):
    """Normalize :32-33 / ResnetBlock norms :113,117 -- 32 groups, eps 1e-6."""
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _conv(x, sd, p, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def resnet_block(x, sd, r):
    """ResnetBlock.forward :126-143 with temb=None."""
    h = _conv(swish(_gn(x, sd, r.prefix + ".norm1")), sd, r.prefix + ".conv1")
    h = _conv(swish(_gn(h, sd, r.prefix + ".norm2")), sd, r.prefix + ".conv2")
    if r.cin != r.cout:
        x = _conv(x, sd, r.prefix + ".nin_shortcut", padding=0)
    return x + h


def attn_block(x, sd, a):
    """AttnBlock.forward :161-183: single head over h*w tokens, scale c^-0.5."""
    p = a.prefix
    h_ = _gn(x, sd, p + ".norm")
    q, k, v = (_conv(h_, sd, f"{p}.{n}", padding=0) for n in ("q", "k", "v"))
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(h_, sd, p + ".proj_out", padding=0)


@torch.no_grad()
def decoder_forward(sd, st, z):
    """Decoder.forward :496-527."""
    h = _conv(z, sd, "decoder.conv_in")
    for l in st.layers:
        if isinstance(l, _S.VaeResSpec):
            h = resnet_block(h, sd, l)
        elif isinstance(l, _S.VaeAttnSpec):
            h = attn_block(h, sd, l)
        elif isinstance(l, _S.VaeUpSpec):
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")       # UpsampleDecoder :83
            h = _conv(h, sd, l.prefix)
    h = swish(_gn(h, sd, "decoder.norm_out"))
    return _conv(h, sd, "decoder.conv_out")


@torch.no_grad()
def decode(sd, st, z):
    """AutoencoderKL.decode :1416-1419."""
    return decoder_forward(sd, st, _conv(z, sd, "post_quant_conv", padding=0))


@torch.no_grad()
def decode_first_stage(sd, st, z, scale_factor=0.18215):
    """LatentDiffusion.decode_first_stage :4300, :4346."""
    return decode(sd, st, 1. / scale_factor * z)


@torch.no_grad()
def encoder_forward(sd, st, x):
    """Encoder.forward :393-420 (no attention in the down levels for the SD config)."""
    h = _conv(x, sd, "encoder.conv_in")
    for l in st.layers:
        if isinstance(l, _S.VaeResSpec):
            h = resnet_block(h, sd, l)
        elif isinstance(l, _S.VaeAttnSpec):
            h = attn_block(h, sd, l)
        elif isinstance(l, _S.VaeDownSpec):
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)         # DownsampleEncoder :94-96
            h = F.conv2d(h, sd[l.prefix + ".weight"], sd[l.prefix + ".bias"], stride=2, padding=0)
    h = swish(_gn(h, sd, "encoder.norm_out"))
    return _conv(h, sd, "encoder.conv_out")


@torch.no_grad()
def encode(sd, st, x):
    """AutoencoderKL.encode :1410-1414 -> the moments tensor (mean | logvar) behind the posterior."""
    return _conv(encoder_forward(sd, st, x), sd, "quant_conv", padding=0)


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution :624-638: mean + exp(0.5 clamp(logvar, -30, 20)) * noise."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def get_first_stage_encoding(moments, noise, scale_factor=0.18215):
    """LatentDiffusion.get_first_stage_encoding :4264-4271 on a posterior given by its moments."""
    return scale_factor * gaussian_sample(moments, noise)


def to_image(x):
    """txt2img.py:307 -- clamp((x+1)/2, 0, 1)."""
    return torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
