"""Static structure of the denoiser and the VAE decoder, derived from the constructor kwargs.

One place decides which blocks exist, in which order, with which channel counts, and which
checkpoint keys/shapes they own.  The Python host (weight loading, validation), the weight
synthesiser, the CPU oracle and the C++ executor (which receives the flattened plan through the
C-ABI) all consume this description, so they cannot drift apart.

Reference construction followed here: UNetModel.__init__ stable_diffusion.py:3263-3496
(ResBlock :3145-3195, SpatialTransformer :2872-2883, BasicTransformerBlock :2800-2810,
CrossAttention :2760-2774, GEGLU/FeedForward :2697-2721, Downsample :3087-3098, Upsample :3048-3055)
and Decoder.__init__ stable_diffusion.py:423-494 (ResnetBlock :106-124, AttnBlock :151-159,
UpsampleDecoder :77-81), AutoencoderKL.__init__ :1388-1389.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional, Tuple


# --------------------------------------------------------------------------------------------
# UNet
# --------------------------------------------------------------------------------------------
@dataclass
class ResSpec:
    prefix: str          # checkpoint key prefix, e.g. "input_blocks.1.0"
    cin: int
    cout: int

    @property
    def has_skip_conv(self):
        return self.cin != self.cout


@dataclass
class STSpec:
    prefix: str          # e.g. "input_blocks.1.1"
    ch: int              # image channels in/out
    heads: int
    d_head: int
    depth: int
    context_dim: int

    @property
    def inner(self):
        return self.heads * self.d_head


@dataclass
class DownSpec:
    prefix: str          # "<block>.op" conv3x3 stride 2
    ch: int


@dataclass
class UpSpec:
    prefix: str          # "<block>.conv" conv3x3 after nearest x2
    ch: int


@dataclass
class UNetStructure:
    in_channels: int
    model_channels: int
    out_channels: int
    time_embed_dim: int
    context_dim: int
    input_blocks: List[list] = field(default_factory=list)    # list of layer lists
    middle_block: list = field(default_factory=list)
    output_blocks: List[list] = field(default_factory=list)
    skip_channels: List[int] = field(default_factory=list)    # channels pushed by each input block
    final_channels: int = 0


def unet_structure(image_size=None, in_channels=4, model_channels=320, out_channels=4,
                   num_res_blocks=2, attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 8),
                   num_heads=-1, num_head_channels=-1, dropout=0., conv_resampler=True,
                   use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                   num_classes=None, use_checkpoint=False, n_embed=None, legacy=True,
                   use_scale_shift_norm=False, resblock_updown=False,
                   use_new_attention_order=False, dims=2, use_fp16=False) -> UNetStructure:
    """Walk the same construction loops as stable_diffusion.py:3334-3496 and record the blocks.

    Only the branch every script of the reference exercises on this path is supported on the
    device: spatial-transformer attention, convolutional resampling, no scale-shift norm, no
    class conditioning, 2-D.  Other settings raise, as loudly as the reference's own asserts.
    """
    if use_spatial_transformer:
        assert context_dim is not None, \
            "Fool!! You forgot to include the dimension of your cross-attention conditioning..."
    if context_dim is not None:
        assert use_spatial_transformer, \
            "Fool!! You forgot to use the spatial transformer for your cross-attention conditioning..."
    if num_heads == -1:
        assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set."
    if num_head_channels == -1:
        assert num_heads != -1, "Either num_heads or num_head_channels has to be set."
    unsupported = []
    if not use_spatial_transformer:
        unsupported.append("use_spatial_transformer=False (AttentionBlock variant)")
    if not conv_resampler:
        unsupported.append("conv_resampler=False")
    if use_scale_shift_norm:
        unsupported.append("use_scale_shift_norm")
    if resblock_updown:
        unsupported.append("resblock_updown")
    if num_classes is not None:
        unsupported.append("num_classes")
    if n_embed is not None:
        unsupported.append("n_embed")
    if dims != 2:
        unsupported.append("dims != 2")
    if dropout != 0.:
        unsupported.append("dropout != 0 (inference path only)")
    if unsupported:
        raise NotImplementedError("not on the MI355X denoising path: " + ", ".join(unsupported))

    mc = model_channels
    st = UNetStructure(in_channels=in_channels, model_channels=mc, out_channels=out_channels,
                       time_embed_dim=4 * mc, context_dim=context_dim)

    def heads_for(ch):
        nh = num_heads
        if num_head_channels == -1:
            dh = ch // nh
        else:
            nh = ch // num_head_channels
            dh = num_head_channels
        if legacy:
            dh = ch // nh        # the spatial-transformer arm of the legacy rule
        return nh, dh

    st.input_blocks.append([("conv_in", "input_blocks.0.0", in_channels, mc)])
    chans = [mc]
    ch, ds = mc, 1
    idx = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            layers = [ResSpec(f"input_blocks.{idx}.0", ch, mult * mc)]
            ch = mult * mc
            if ds in attention_resolutions:
                nh, dh = heads_for(ch)
                layers.append(STSpec(f"input_blocks.{idx}.1", ch, nh, dh, transformer_depth, context_dim))
            st.input_blocks.append(layers)
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            st.input_blocks.append([DownSpec(f"input_blocks.{idx}.0.op", ch)])
            chans.append(ch)
            idx += 1
            ds *= 2
    st.skip_channels = list(chans)

    nh, dh = heads_for(ch)
    st.middle_block = [ResSpec("middle_block.0", ch, ch),
                       STSpec("middle_block.1", ch, nh, dh, transformer_depth, context_dim),
                       ResSpec("middle_block.2", ch, ch)]

    idx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            layers = [ResSpec(f"output_blocks.{idx}.0", ch + ich, mult * mc)]
            ch = mult * mc
            if ds in attention_resolutions:
                nh, dh = heads_for(ch)
                layers.append(STSpec(f"output_blocks.{idx}.{len(layers)}", ch, nh, dh,
                                     transformer_depth, context_dim))
            if level and i == num_res_blocks:
                layers.append(UpSpec(f"output_blocks.{idx}.{len(layers)}.conv", ch))
                ds //= 2
            st.output_blocks.append(layers)
            idx += 1
    st.final_channels = ch
    return st


def _res_params(r: ResSpec, temb: int, out: OrderedDict):
    p = r.prefix
    out[f"{p}.in_layers.0.weight"] = (r.cin,)
    out[f"{p}.in_layers.0.bias"] = (r.cin,)
    out[f"{p}.in_layers.2.weight"] = (r.cout, r.cin, 3, 3)
    out[f"{p}.in_layers.2.bias"] = (r.cout,)
    out[f"{p}.emb_layers.1.weight"] = (r.cout, temb)
    out[f"{p}.emb_layers.1.bias"] = (r.cout,)
    out[f"{p}.out_layers.0.weight"] = (r.cout,)
    out[f"{p}.out_layers.0.bias"] = (r.cout,)
    out[f"{p}.out_layers.3.weight"] = (r.cout, r.cout, 3, 3)
    out[f"{p}.out_layers.3.bias"] = (r.cout,)
    if r.has_skip_conv:
        out[f"{p}.skip_connection.weight"] = (r.cout, r.cin, 1, 1)
        out[f"{p}.skip_connection.bias"] = (r.cout,)


def _st_params(s: STSpec, out: OrderedDict):
    p, d = s.prefix, s.inner
    out[f"{p}.norm.weight"] = (s.ch,)
    out[f"{p}.norm.bias"] = (s.ch,)
    out[f"{p}.proj_in.weight"] = (d, s.ch, 1, 1)
    out[f"{p}.proj_in.bias"] = (d,)
    for b in range(s.depth):
        q = f"{p}.transformer_blocks.{b}"
        out[f"{q}.attn1.to_q.weight"] = (d, d)
        out[f"{q}.attn1.to_k.weight"] = (d, d)
        out[f"{q}.attn1.to_v.weight"] = (d, d)
        out[f"{q}.attn1.to_out.0.weight"] = (d, d)
        out[f"{q}.attn1.to_out.0.bias"] = (d,)
        out[f"{q}.ff.net.0.proj.weight"] = (8 * d, d)
        out[f"{q}.ff.net.0.proj.bias"] = (8 * d,)
        out[f"{q}.ff.net.2.weight"] = (d, 4 * d)
        out[f"{q}.ff.net.2.bias"] = (d,)
        out[f"{q}.attn2.to_q.weight"] = (d, d)
        out[f"{q}.attn2.to_k.weight"] = (d, s.context_dim)
        out[f"{q}.attn2.to_v.weight"] = (d, s.context_dim)
        out[f"{q}.attn2.to_out.0.weight"] = (d, d)
        out[f"{q}.attn2.to_out.0.bias"] = (d,)
        for n in (1, 2, 3):
            out[f"{q}.norm{n}.weight"] = (d,)
            out[f"{q}.norm{n}.bias"] = (d,)
    out[f"{p}.proj_out.weight"] = (s.ch, d, 1, 1)
    out[f"{p}.proj_out.bias"] = (s.ch,)


def unet_param_shapes(st: UNetStructure) -> "OrderedDict[str, Tuple[int, ...]]":
    """Checkpoint keys (relative to ``model.diffusion_model.``) and their shapes."""
    out = OrderedDict()
    mc, te = st.model_channels, st.time_embed_dim
    out["time_embed.0.weight"] = (te, mc)
    out["time_embed.0.bias"] = (te,)
    out["time_embed.2.weight"] = (te, te)
    out["time_embed.2.bias"] = (te,)

    def walk(layers):
        for l in layers:
            if isinstance(l, tuple):
                _, p, cin, cout = l
                out[f"{p}.weight"] = (cout, cin, 3, 3)
                out[f"{p}.bias"] = (cout,)
            elif isinstance(l, ResSpec):
                _res_params(l, te, out)
            elif isinstance(l, STSpec):
                _st_params(l, out)
            elif isinstance(l, (DownSpec, UpSpec)):
                out[f"{l.prefix}.weight"] = (l.ch, l.ch, 3, 3)
                out[f"{l.prefix}.bias"] = (l.ch,)

    for blk in st.input_blocks:
        walk(blk)
    walk(st.middle_block)
    for blk in st.output_blocks:
        walk(blk)
    out["out.0.weight"] = (st.final_channels,)
    out["out.0.bias"] = (st.final_channels,)
    out["out.2.weight"] = (st.out_channels, st.model_channels, 3, 3)
    out["out.2.bias"] = (st.out_channels,)
    return out


def unet_zero_init_keys(st: UNetStructure) -> List[str]:
    """Tensors the reference zero-initialises (zero_module, stable_diffusion.py:3188,2883,3495).

    A random-weight run must re-randomise them or the UNet output is identically zero."""
    keys = []

    def walk(layers):
        for l in layers:
            if isinstance(l, ResSpec):
                keys.extend([f"{l.prefix}.out_layers.3.weight", f"{l.prefix}.out_layers.3.bias"])
            elif isinstance(l, STSpec):
                keys.extend([f"{l.prefix}.proj_out.weight", f"{l.prefix}.proj_out.bias"])

    for blk in st.input_blocks:
        walk(blk)
    walk(st.middle_block)
    for blk in st.output_blocks:
        walk(blk)
    keys.extend(["out.2.weight", "out.2.bias"])
    return keys


# --------------------------------------------------------------------------------------------
# VAE decoder
# --------------------------------------------------------------------------------------------
@dataclass
class VaeResSpec:
    prefix: str
    cin: int
    cout: int


@dataclass
class VaeAttnSpec:
    prefix: str
    ch: int


@dataclass
class VaeUpSpec:
    prefix: str      # "<...>.upsample.conv"
    ch: int


@dataclass
class VaeDownSpec:
    prefix: str      # "<...>.downsample.conv"
    ch: int


@dataclass
class VaeDecoderStructure:
    z_channels: int
    embed_dim: int
    block_in: int
    out_ch: int
    layers: list = field(default_factory=list)   # execution order after conv_in
    final_channels: int = 0


def vae_decoder_structure(ddconfig: dict, embed_dim: int = 4) -> VaeDecoderStructure:
    """Execution-order description of Decoder (stable_diffusion.py:422-527)."""
    ch_out = ddconfig["ch_out"]
    ch_mult = tuple(ddconfig.get("ch_mult", (1, 2, 4, 8)))
    nrb = ddconfig["num_res_blocks"]
    attn_res = list(ddconfig.get("attn_resolutions", []))
    resolution = ddconfig["resolution"]
    z_channels = ddconfig["z_channels"]
    if ddconfig.get("dropout", 0.0) != 0.0:
        raise NotImplementedError("dropout != 0 (inference path only)")
    if not ddconfig.get("resamp_with_conv", True):
        raise NotImplementedError("resamp_with_conv=False")
    if ddconfig.get("attn_type", "vanilla") != "vanilla" or ddconfig.get("use_linear_attn", False):
        raise NotImplementedError("only vanilla attention in the decoder mid block")
    if ddconfig.get("give_pre_end", False) or ddconfig.get("tanh_out", False):
        raise NotImplementedError("give_pre_end / tanh_out")
    nres = len(ch_mult)
    block_in = ch_out * ch_mult[nres - 1]
    curr_res = resolution // 2 ** (nres - 1)
    st = VaeDecoderStructure(z_channels=z_channels, embed_dim=embed_dim, block_in=block_in,
                             out_ch=ddconfig["out_ch"])
    st.layers.append(VaeResSpec("decoder.mid.block_1", block_in, block_in))
    st.layers.append(VaeAttnSpec("decoder.mid.attn_1", block_in))
    st.layers.append(VaeResSpec("decoder.mid.block_2", block_in, block_in))
    for i_level in reversed(range(nres)):
        block_out = ch_out * ch_mult[i_level]
        for i_block in range(nrb + 1):
            st.layers.append(VaeResSpec(f"decoder.up.{i_level}.block.{i_block}", block_in, block_out))
            block_in = block_out
            if curr_res in attn_res:
                st.layers.append(VaeAttnSpec(f"decoder.up.{i_level}.attn.{i_block}", block_in))
        if i_level != 0:
            st.layers.append(VaeUpSpec(f"decoder.up.{i_level}.upsample.conv", block_in))
            curr_res *= 2
    st.final_channels = block_in
    return st


def _vae_layer_shapes(layers) -> "OrderedDict[str, Tuple[int, ...]]":
    out = OrderedDict()
    for l in layers:
        p = l.prefix
        if isinstance(l, VaeResSpec):
            out[f"{p}.norm1.weight"] = (l.cin,)
            out[f"{p}.norm1.bias"] = (l.cin,)
            out[f"{p}.conv1.weight"] = (l.cout, l.cin, 3, 3)
            out[f"{p}.conv1.bias"] = (l.cout,)
            out[f"{p}.norm2.weight"] = (l.cout,)
            out[f"{p}.norm2.bias"] = (l.cout,)
            out[f"{p}.conv2.weight"] = (l.cout, l.cout, 3, 3)
            out[f"{p}.conv2.bias"] = (l.cout,)
            if l.cin != l.cout:
                out[f"{p}.nin_shortcut.weight"] = (l.cout, l.cin, 1, 1)
                out[f"{p}.nin_shortcut.bias"] = (l.cout,)
        elif isinstance(l, VaeAttnSpec):
            out[f"{p}.norm.weight"] = (l.ch,)
            out[f"{p}.norm.bias"] = (l.ch,)
            for n in ("q", "k", "v", "proj_out"):
                out[f"{p}.{n}.weight"] = (l.ch, l.ch, 1, 1)
                out[f"{p}.{n}.bias"] = (l.ch,)
        elif isinstance(l, (VaeUpSpec, VaeDownSpec)):
            out[f"{p}.weight"] = (l.ch, l.ch, 3, 3)
            out[f"{p}.bias"] = (l.ch,)
    return out


def vae_decoder_param_shapes(st: VaeDecoderStructure) -> "OrderedDict[str, Tuple[int, ...]]":
    """Checkpoint keys (relative to ``first_stage_model.``) used by decode(), and their shapes."""
    out = OrderedDict()
    out["post_quant_conv.weight"] = (st.z_channels, st.embed_dim, 1, 1)
    out["post_quant_conv.bias"] = (st.z_channels,)
    out["decoder.conv_in.weight"] = (st.block_in, st.z_channels, 3, 3)
    out["decoder.conv_in.bias"] = (st.block_in,)
    out.update(_vae_layer_shapes(st.layers))
    out["decoder.norm_out.weight"] = (st.final_channels,)
    out["decoder.norm_out.bias"] = (st.final_channels,)
    out["decoder.conv_out.weight"] = (st.out_ch, st.final_channels, 3, 3)
    out["decoder.conv_out.bias"] = (st.out_ch,)
    return out


# --------------------------------------------------------------------------------------------
# VAE encoder
# --------------------------------------------------------------------------------------------
@dataclass
class VaeEncoderStructure:
    z_channels: int
    embed_dim: int
    ch: int
    in_channels: int
    layers: list = field(default_factory=list)   # execution order after conv_in
    final_channels: int = 0


def vae_encoder_structure(ddconfig: dict, embed_dim: int = 4) -> VaeEncoderStructure:
    """Execution-order description of Encoder (stable_diffusion.py:326-420)."""
    ch_out = ddconfig["ch_out"]
    ch_mult = tuple(ddconfig.get("ch_mult", (1, 2, 4, 8)))
    nrb = ddconfig["num_res_blocks"]
    if list(ddconfig.get("attn_resolutions", [])):
        raise NotImplementedError("attention inside the encoder's down levels (attn_resolutions)")
    if ddconfig.get("dropout", 0.0) != 0.0:
        raise NotImplementedError("dropout != 0 (inference path only)")
    if not ddconfig.get("resamp_with_conv", True):
        raise NotImplementedError("resamp_with_conv=False")
    if ddconfig.get("attn_type", "vanilla") != "vanilla" or ddconfig.get("use_linear_attn", False):
        raise NotImplementedError("only vanilla attention in the encoder mid block")
    if not ddconfig.get("double_z", True):
        raise NotImplementedError("double_z=False")
    st = VaeEncoderStructure(z_channels=ddconfig["z_channels"], embed_dim=embed_dim, ch=ch_out,
                             in_channels=ddconfig.get("in_channels", ddconfig["out_ch"]))
    block_in = ch_out
    for i_level in range(len(ch_mult)):
        block_out = ch_out * ch_mult[i_level]
        for i_block in range(nrb):
            st.layers.append(VaeResSpec(f"encoder.down.{i_level}.block.{i_block}", block_in, block_out))
            block_in = block_out
        if i_level != len(ch_mult) - 1:
            st.layers.append(VaeDownSpec(f"encoder.down.{i_level}.downsample.conv", block_in))
    st.layers.append(VaeResSpec("encoder.mid.block_1", block_in, block_in))
    st.layers.append(VaeAttnSpec("encoder.mid.attn_1", block_in))
    st.layers.append(VaeResSpec("encoder.mid.block_2", block_in, block_in))
    st.final_channels = block_in
    return st


def vae_encoder_param_shapes(st: VaeEncoderStructure) -> "OrderedDict[str, Tuple[int, ...]]":
    """Checkpoint keys (relative to ``first_stage_model.``) used by encode(), and their shapes."""
    out = OrderedDict()
    out["encoder.conv_in.weight"] = (st.ch, st.in_channels, 3, 3)
    out["encoder.conv_in.bias"] = (st.ch,)
    out.update(_vae_layer_shapes(st.layers))
    out["encoder.norm_out.weight"] = (st.final_channels,)
    out["encoder.norm_out.bias"] = (st.final_channels,)
    out["encoder.conv_out.weight"] = (2 * st.z_channels, st.final_channels, 3, 3)
    out["encoder.conv_out.bias"] = (2 * st.z_channels,)
    out["quant_conv.weight"] = (2 * st.embed_dim, 2 * st.z_channels, 1, 1)
    out["quant_conv.bias"] = (2 * st.embed_dim,)
    return out


# --------------------------------------------------------------------------------------------
# The model/config dictionaries the reference scripts hard-code (txt2img.py:14-42,
# inpaint_with_text.py:14-43): same keys, same values, so a caller can pass either.
# --------------------------------------------------------------------------------------------
def sd_v1_unet_config(in_channels: int = 4) -> dict:
    return {
        'image_size': 32, 'in_channels': in_channels, 'out_channels': 4,
        'model_channels': 320, 'attention_resolutions': [4, 2, 1],
        'num_res_blocks': 2, 'channel_mult': [1, 2, 4, 4],
        'num_heads': 8, 'use_spatial_transformer': True,
        'transformer_depth': 1, 'context_dim': 768,
        'use_checkpoint': True, 'legacy': False,
    }


def sd_v1_ddconfig() -> dict:
    return {
        'double_z': True, 'z_channels': 4, 'resolution': 512, 'in_channels': 3,
        'out_ch': 3, 'ch_out': 128, 'ch_mult': [1, 2, 4, 4], 'num_res_blocks': 2,
        'attn_resolutions': [], 'dropout': 0.0,
    }


def sd_v1_model_params(in_channels: int = 4, conditioning_key: str = 'crossattn') -> dict:
    return {
        'linear_start': 0.00085, 'linear_end': 0.012, 'timesteps': 1000,
        'image_size': 64, 'channels': 4, 'scale_factor': 0.18215, 'use_ema': False,
        'first_stage_config': {'embed_dim': 4, 'ddconfig': sd_v1_ddconfig(),
                               'lossconfig': {'target': 'torch.nn.Identity'}},
        'cond_stage_config': '__is_first_stage__',
        'conditioning_key': conditioning_key,
        'unet_config': sd_v1_unet_config(in_channels),
    }
