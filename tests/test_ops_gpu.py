"""GPU: every HIP kernel, called through the C ABI, against (a) plain fp32 torch-CPU references of
the same op and (b) the golden block vectors the reference produced (tests/golden/g2_blocks.npz).

Tolerances are max|a-b|/max|b|: fp32 mode 2e-5 per op (fp32 MFMA accumulation order differs from the
CPU's), fp16 3e-3, bf16 2e-2 per op -- well inside the per-UNet-call bars (1e-4 / 2e-2).
"""
import importlib
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import golden, max_rel_err, g2_u, structure as S, synth

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-5, "fp16": 3e-3, "bf16": 2e-2}
DTYPES = ["fp32", "fp16", "bf16"]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return importlib.import_module("stable-diffusion_amd.ops")


def tdt(name):
    return {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[name]


def nhwc(x, dt):
    """NCHW fp32 cpu -> NHWC device tensor of dtype dt."""
    return x.permute(0, 2, 3, 1).contiguous().cuda().to(tdt(dt))


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def q(x, dt):
    """Quantise a CPU fp32 tensor to what the device sees in dtype dt (so the reference uses equal inputs)."""
    return x.to(tdt(dt)).float()


def rnd(key, shape, lo=-1., hi=1.):
    return synth.uniform(3, "ops." + key, shape, lo, hi)


CONV_CASES = [
    # name, B, H, W, C0, C1, Cout, taps, stride, up, splits
    ("small", 2, 8, 8, 64, 0, 128, 9, 1, 0, 1),
    ("edge_mn", 2, 6, 6, 64, 0, 320, 9, 1, 0, 1),
    ("c320", 1, 16, 16, 320, 0, 320, 9, 1, 0, 1),
    ("down", 2, 16, 16, 128, 0, 128, 9, 2, 0, 1),
    ("down_odd", 1, 9, 9, 64, 0, 64, 9, 2, 0, 1),
    ("up", 2, 8, 8, 128, 0, 64, 9, 1, 1, 1),
    ("concat", 2, 8, 8, 128, 64, 192, 9, 1, 0, 1),
    ("concat1x1", 2, 8, 8, 128, 64, 192, 1, 1, 0, 1),
    ("splitk", 2, 8, 8, 256, 0, 128, 9, 1, 0, 4),
    ("splitk_concat", 1, 8, 8, 128, 128, 64, 9, 1, 0, 3),
    ("pointwise", 1, 24, 24, 128, 0, 256, 1, 1, 0, 1),
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(ops, dt, case):
    name, B, H, W, C0, C1, Cout, taps, stride, up, splits = case
    k = 3 if taps == 9 else 1
    x = q(rnd(name + ".x", (B, C0 + C1, H, W)), dt)
    w = q(rnd(name + ".w", (Cout, C0 + C1, k, k), -0.05, 0.05), dt)
    b = rnd(name + ".b", (Cout,))
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=1 if taps == 9 else 0)
    wp = ops.pack_weight(w.cuda(), dt)
    x0 = nhwc(x[:, :C0], dt)
    x1 = nhwc(x[:, C0:], dt) if C1 else None
    y = ops.conv2d(x0, wp, b.cuda(), taps=taps, stride=stride, up=bool(up), x1=x1, splits=splits)
    assert tuple(y.shape) == (B, ref.shape[2], ref.shape[3], Cout)
    assert max_rel_err(nchw(y), ref) < TOL[dt], name


# Large-tile (256 x 320 / 256 x 256, two wave groups) kernel: forced through the environment switch
# because the planner only picks it for problems far larger than a CPU reference should chew on.
PP_CASES = [
    # name, B, H, W, C0, C1, Cout, taps, stride, up, splits
    ("pp_c320", 1, 16, 16, 320, 0, 320, 9, 1, 0, 1),
    ("pp_two_tiles_n", 2, 16, 16, 128, 0, 640, 9, 1, 0, 1),
    ("pp_ragged_m", 3, 10, 10, 64, 0, 320, 9, 1, 0, 1),
    ("pp_n256", 2, 12, 12, 128, 0, 512, 9, 1, 0, 1),
    ("pp_concat", 2, 16, 16, 128, 64, 320, 9, 1, 0, 1),
    ("pp_down", 2, 32, 32, 64, 0, 320, 9, 2, 0, 1),
    ("pp_up", 2, 8, 8, 128, 0, 320, 9, 1, 1, 1),
    ("pp_splitk", 2, 16, 16, 192, 0, 320, 9, 1, 0, 3),
    ("pp_splitk_concat", 1, 16, 16, 128, 128, 256, 9, 1, 0, 4),
    ("pp_pointwise", 1, 24, 24, 256, 0, 640, 1, 1, 0, 1),
    ("pp_one_ktile_per_split", 1, 16, 16, 128, 0, 320, 1, 1, 0, 2),
]


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("case", PP_CASES, ids=[c[0] for c in PP_CASES])
def test_conv2d_large_tiles(ops, dt, case, monkeypatch):
    monkeypatch.setenv("SDMI_GEMM_SHAPE", "2")
    test_conv2d(ops, dt, case)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("case", PP_CASES, ids=[c[0] for c in PP_CASES])
def test_conv2d_tiles_256x160(ops, dt, case, monkeypatch):
    """Same cases through the three-stage 256 x 160 shape (taken where N is a multiple of 160)."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", "3")
    test_conv2d(ops, dt, case)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("case", PP_CASES, ids=[c[0] for c in PP_CASES])
def test_conv2d_tiles_128x160(ops, dt, case, monkeypatch):
    """Same cases through the 128 x 160 shape (two row fragments per wave)."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", "4")
    test_conv2d(ops, dt, case)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("case", PP_CASES, ids=[c[0] for c in PP_CASES])
def test_conv2d_tiles_64x160(ops, dt, case, monkeypatch):
    """Same cases through the 64 x 160 shape (one row fragment per wave; small problems at UNet batch 2)."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", "7")
    test_conv2d(ops, dt, case)


PP128_CASES = [
    # name, B, H, W, C0, C1, Cout, taps, stride, up, splits   (N a multiple of 128)
    ("pp128_c128", 2, 24, 24, 128, 0, 128, 9, 1, 0, 1),
    ("pp128_three_tiles_n_ragged_m", 3, 10, 10, 64, 0, 384, 9, 1, 0, 1),
    ("pp128_concat_up", 1, 12, 12, 128, 64, 128, 9, 1, 1, 1),
    ("pp128_down_splitk", 2, 32, 32, 192, 0, 256, 9, 2, 0, 3),
    ("pp128_pointwise", 1, 24, 24, 256, 0, 640, 1, 1, 0, 1),
]


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("order", ["0", "1"])
@pytest.mark.parametrize("case", PP128_CASES, ids=[c[0] for c in PP128_CASES])
def test_conv2d_tiles_256x128(ops, dt, order, case, monkeypatch):
    """The 256 x 128 shape (the planner's choice for the VAE decoder's 128-channel stage), both K orders."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", "6")
    monkeypatch.setenv("SDMI_GEMM_TAP_INNER", order)
    test_conv2d(ops, dt, case)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("order", ["0", "1"])
@pytest.mark.parametrize("shape", ["2", "3", "4", "7"])
def test_large_tiles_both_k_orders(ops, dt, order, shape, monkeypatch):
    """3x3 convolutions through the tap-major and the tap-inner K walk of every large-tile shape: concat
    sources, stride 2, ragged M, split-K (splits start mid-slice), long K, the encoder's asymmetric padding."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", shape)
    monkeypatch.setenv("SDMI_GEMM_TAP_INNER", order)
    for case in PP_CASES:
        if case[7] == 9:
            test_conv2d(ops, dt, case)
    test_conv2d(ops, dt, ("pp_long_k_split", 1, 16, 16, 448, 192, 320, 9, 1, 0, 5))
    x = q(rnd("order.x", (1, 64, 12, 20)), dt)
    w = q(rnd("order.w", (320, 64, 3, 3), -0.05, 0.05), dt)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2, padding=0)
    y = ops.conv2d(nhwc(x, dt), ops.pack_weight(w.cuda(), dt), None, taps=9, stride=2, pad_lo=0)
    assert max_rel_err(nchw(y), ref) < TOL[dt]


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("shape", ["2", "3", "4", "7"])
def test_large_tiles_direct_store_path(ops, dt, shape, monkeypatch):
    """The store path for outputs that are not 16-byte addressable (straight from the accumulator layout)."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", shape)
    monkeypatch.setenv("SDMI_GEMM_DIRECT_STORE", "1")
    for case in PP_CASES[:4]:
        test_conv2d(ops, dt, case)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("shape", ["2", "3", "4", "7"])
def test_large_tiles_long_k(ops, dt, shape, monkeypatch):
    """Many K-tiles across taps and both concat sources: ring wrap-around and segment changes."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", shape)
    test_conv2d(ops, dt, ("pp_long_k", 1, 16, 16, 448, 192, 320, 9, 1, 0, 1))
    test_conv2d(ops, dt, ("pp_long_k_split", 1, 16, 16, 448, 192, 320, 9, 1, 0, 5))


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_large_tiles_epilogues(ops, dt, monkeypatch):
    """bias + time-embedding row add + residual and the GEGLU gate through the large-tile kernel."""
    B, H, W, C, Cout = 2, 16, 16, 128, 320
    x = q(rnd("ppepi.x", (B, C, H, W)), dt)
    w = q(rnd("ppepi.w", (Cout, C, 3, 3), -0.05, 0.05), dt)
    b = rnd("ppepi.b", (Cout,))
    emb = rnd("ppepi.emb", (B, Cout))
    res = q(rnd("ppepi.res", (B, Cout, H, W)), dt)
    ref = F.conv2d(x, w, b, padding=1) + emb[:, :, None, None] + res
    wp = ops.pack_weight(w.cuda(), dt)
    for shape in ("2", "3", "4", "7"):
        monkeypatch.setenv("SDMI_GEMM_SHAPE", shape)
        for splits in (1, 2):
            y = ops.conv2d(nhwc(x, dt), wp, b.cuda(), taps=9, rowadd=emb.cuda(), resid=nhwc(res, dt), splits=splits)
            assert max_rel_err(nchw(y), ref) < TOL[dt], (shape, splits)
    monkeypatch.setenv("SDMI_GEMM_SHAPE", "2")
    rows, d = 300, 320                                   # N = 8 d = 2560 = 8 x 320
    x = q(rnd("ppgeglu.x", (rows, d)), dt)
    w = q(rnd("ppgeglu.w", (8 * d, d), -0.1, 0.1), dt)
    b = rnd("ppgeglu.b", (8 * d,))
    a, g = F.linear(x, w, b).chunk(2, dim=-1)
    ref = a * F.gelu(g)
    wp, bp = ops.pack_geglu(w.cuda(), b.cuda(), dt)
    for splits in (1, 2):
        y = ops.linear(x.cuda().to(tdt(dt)), wp, bp, geglu=True, splits=splits)
        assert max_rel_err(y.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", ["1", "2"])
def test_conv2d_encoder_downsample(ops, dt, shape, monkeypatch):
    """DownsampleEncoder: F.pad(x, (0,1,0,1)) then conv 3x3 stride 2 padding 0."""
    monkeypatch.setenv("SDMI_GEMM_SHAPE", shape)
    for (B, H, W, C, Cout) in ((2, 16, 16, 128, 256), (1, 12, 20, 64, 320)):
        x = q(rnd("encdown.x", (B, C, H, W)), dt)
        w = q(rnd("encdown.w", (Cout, C, 3, 3), -0.05, 0.05), dt)
        b = rnd("encdown.b", (Cout,))
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2, padding=0)
        y = ops.conv2d(nhwc(x, dt), ops.pack_weight(w.cuda(), dt), b.cuda(), taps=9, stride=2, pad_lo=0)
        assert tuple(y.shape) == (B, H // 2, W // 2, Cout)
        assert max_rel_err(nchw(y), ref) < TOL[dt]


def test_conv2d_planned_splits(ops):
    """splits=None asks the library for the executor's own split count."""
    x = q(rnd("plan.x", (2, 256, 8, 8)), "fp16")
    w = q(rnd("plan.w", (128, 256, 3, 3), -0.05, 0.05), "fp16")
    ref = F.conv2d(x, w, None, padding=1)
    y = ops.conv2d(nhwc(x, "fp16"), ops.pack_weight(w.cuda(), "fp16"), None, taps=9, splits=None)
    assert max_rel_err(nchw(y), ref) < TOL["fp16"]


@pytest.mark.parametrize("dt", DTYPES)
def test_conv2d_epilogues(ops, dt):
    """bias + per-sample row add (time embedding) + residual, with and without split-K."""
    B, H, W, C, Cout = 2, 8, 8, 128, 192
    x = q(rnd("epi.x", (B, C, H, W)), dt)
    w = q(rnd("epi.w", (Cout, C, 3, 3), -0.05, 0.05), dt)
    b = rnd("epi.b", (Cout,))
    emb = rnd("epi.emb", (B, 256))           # wider row stride than Cout is exercised through a slice
    res = q(rnd("epi.res", (B, Cout, H, W)), dt)
    ref = F.conv2d(x, w, b, padding=1) + emb[:, :Cout, None, None] + res
    wp = ops.pack_weight(w.cuda(), dt)
    for splits in (1, 3):
        y = ops.conv2d(nhwc(x, dt), wp, b.cuda(), taps=9, rowadd=emb[:, :Cout].contiguous().cuda(),
                       resid=nhwc(res, dt), splits=splits)
        assert max_rel_err(nchw(y), ref) < TOL[dt], splits


@pytest.mark.parametrize("dt", DTYPES)
def test_linear_geglu(ops, dt):
    rows, d = 200, 128
    x = q(rnd("geglu.x", (rows, d)), dt)
    w = q(rnd("geglu.w", (8 * d, d), -0.1, 0.1), dt)
    b = rnd("geglu.b", (8 * d,))
    a, g = F.linear(x, w, b).chunk(2, dim=-1)
    ref = a * F.gelu(g)
    wp, bp = ops.pack_geglu(w.cuda(), b.cuda(), dt)
    for splits in (1, 2):
        y = ops.linear(x.cuda().to(tdt(dt)), wp, bp, geglu=True, splits=splits)
        assert tuple(y.shape) == (rows, 4 * d)
        assert max_rel_err(y.float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C0,C1,HW", [(320, 0, 16), (128, 0, 24), (1280, 640, 8), (2560, 0, 4), (64, 64, 10),
                                      (320, 0, 64), (640, 320, 48), (640, 320, 32)])
def test_groupnorm(ops, dt, C0, C1, HW):
    C = C0 + C1
    x = q(rnd(f"gn{C0}_{C1}.x", (2, C, HW, HW), -3., 5.), dt)
    g = rnd("gn.g", (C,), 0.5, 1.5)
    b = rnd("gn.b", (C,))
    for silu, eps in ((True, 1e-5), (False, 1e-6)):
        ref = F.group_norm(x, 32, g, b, eps)
        if silu:
            ref = F.silu(ref)
        y = ops.groupnorm(nhwc(x[:, :C0], dt), g.cuda(), b.cuda(), eps, silu=silu,
                          x1=nhwc(x[:, C0:], dt) if C1 else None)
        assert max_rel_err(nchw(y), ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,groups", [(1, 32), (3, 32), (8, 32), (16, 32), (5, 4)])
def test_groupnorm_batches_and_group_counts(ops, dt, B, groups):
    """Small tensors take the one-launch kernel, whose workgroup -> (sample, group) mapping depends on whether the
    batch and the group count are multiples of 8; every mapping must cover every (sample, group) once."""
    C, HW = 640, 12
    x = q(rnd(f"gnb{B}.x", (B, C, HW, HW), -3., 5.), dt)
    g = rnd("gnb.g", (C,), 0.5, 1.5)
    b = rnd("gnb.b", (C,))
    ref = F.silu(F.group_norm(x, groups, g, b, 1e-5))
    y = ops.groupnorm(nhwc(x, dt), g.cuda(), b.cuda(), 1e-5, silu=True, groups=groups)
    assert max_rel_err(nchw(y), ref) < TOL[dt]


@pytest.mark.parametrize("shape", [(16, 32, 640, 0), (16, 32, 1280, 0), (8, 32, 1280, 640), (4, 32, 2560, 0),
                                   (16, 16, 1280, 0), (16, 8, 1280, 1280), (16, 64, 320, 0)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_groupnorm_unet_sizes(ops, shape):
    """GroupNorm + SiLU at the UNet's batch-16 tensor sizes: every register budget of the one-launch kernel (8 to 48
    words per thread), the sample-per-XCD workgroup mapping, a two-source input, and the two-launch path (64 x 64)."""
    B, HW, C0, C1 = shape
    C = C0 + C1
    x = q(rnd(f"gnfull{HW}_{C0}_{C1}.x", (B, C, HW, HW), -3., 5.), "fp16")
    g = rnd("gnfull.g", (C,), 0.5, 1.5)
    b = rnd("gnfull.b", (C,))
    ref = F.silu(F.group_norm(x, 32, g, b, 1e-5))
    y = ops.groupnorm(nhwc(x[:, :C0], "fp16"), g.cuda(), b.cuda(), 1e-5, silu=True,
                      x1=nhwc(x[:, C0:], "fp16") if C1 else None)
    assert max_rel_err(nchw(y), ref) < TOL["fp16"]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("d", [320, 640, 1280, 512])
def test_layernorm(ops, dt, d):
    x = q(rnd(f"ln{d}.x", (70, d), -3., 5.), dt)
    g = rnd("ln.g", (d,), 0.5, 1.5)
    b = rnd("ln.b", (d,))
    ref = F.layer_norm(x, (d,), g, b, 1e-5)
    y = ops.layernorm(x.cuda().to(tdt(dt)), g.cuda(), b.cuda())
    assert max_rel_err(y.float().cpu(), ref) < TOL[dt]


def _attn_ref(qv, kv, vv, heads):
    B, Nq, C = qv.shape
    d = C // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)
    s = torch.matmul(sp(qv), sp(kv).transpose(-1, -2)) * d ** -0.5
    return torch.matmul(s.softmax(-1), sp(vv)).permute(0, 2, 1, 3).reshape(B, Nq, C)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("d,Nq,Nk", [(40, 256, 256), (40, 144, 77), (80, 64, 64), (80, 200, 77), (160, 64, 64),
                                     (160, 36, 77), (40, 1024, 1024), (40, 1100, 1100), (40, 1024, 1100),
                                     (80, 300, 333),
                                     # enough workgroups for the fp32 kernel's two-query-tiles-per-wave variant
                                     (40, 4096, 150), (160, 4100, 50)])
def test_attention(ops, dt, impl, d, Nq, Nk):
    heads, B = 8, 2
    C = heads * d
    # scores of a few units so that the softmax is far from uniform
    qv = q(rnd(f"att{d}.q", (B, Nq, C), -2., 2.), dt)
    kv = q(rnd(f"att{d}.k", (B, Nk, C), -2., 2.), dt)
    vv = q(rnd(f"att{d}.v", (B, Nk, C), -1., 1.), dt)
    ref = _attn_ref(qv, kv, vv, heads)
    dev = lambda t: t.cuda().to(tdt(dt))
    y = ops.attention(dev(qv), dev(kv), dev(vv), heads, impl=impl)
    assert max_rel_err(y.float().cpu(), ref) < TOL[dt] * 2
    if Nq == Nk:
        # fused q|k|v buffer with row stride 3C, as the UNet executor lays it out
        qkv = torch.cat([qv, kv, vv], dim=-1).cuda().to(tdt(dt))
        y2 = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], heads, impl=impl)
        assert torch.equal(y2, y)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("Nq,Nk,qpw", [(16384 + 70, 77, None), (1000, 77, 3), (1024, 64, 8), (700, 128, 2),
                                       (512, 20, 5), (130, 100, 64)])
def test_attention_short_keys_resident(ops, monkeypatch, dt, Nq, Nk, qpw):
    """d = 40 with at most 128 keys (the prompt tokens at the 64x64 level): the variant that keeps both key tiles in
    LDS and walks several query blocks per workgroup.  qpw None: the launcher's own choice (enough blocks for 2 per
    workgroup, ragged last block and last chunk); otherwise forced.  Same arithmetic as the one-block-per-workgroup
    kernel, so the two must agree bit for bit; a spiked key in the second tile forces the rescale path."""
    heads, B, d = 8, 2, 40
    C = heads * d
    qv = q(rnd("attres.q", (B, Nq, C), -2., 2.), dt)
    kv = rnd("attres.k", (B, Nk, C), -2., 2.)
    if Nk > 70:
        kv[:, 70] = qv[:, 5].float() * 4.0
    kv = q(kv, dt)
    vv = q(rnd("attres.v", (B, Nk, C), -1., 1.), dt)
    ref = _attn_ref(qv, kv, vv, heads)
    dev = lambda t: t.cuda().to(tdt(dt))
    monkeypatch.setenv("SDMI_XATTN_QPW", "1")
    y1 = ops.attention(dev(qv), dev(kv), dev(vv), heads)
    if qpw is None:
        monkeypatch.delenv("SDMI_XATTN_QPW")
    else:
        monkeypatch.setenv("SDMI_XATTN_QPW", str(qpw))
    y = ops.attention(dev(qv), dev(kv), dev(vv), heads)
    assert max_rel_err(y.float().cpu(), ref) < TOL[dt] * 2
    assert torch.equal(y, y1)


@pytest.mark.parametrize("d", [40, 64, 80, 160])
@pytest.mark.parametrize("Nq,Nk", [(256, 1024), (512, 77)])
def test_attention_large_common_score(ops, d, Nq, Nk):
    """Every score near +110 (in base-2 units): the running maximum is far above any partial dot product, so a
    maximum taken from accumulators the matrix pipe has not finished writing would move the reference point by
    tens of units and flush the probabilities to zero.  Also: two calls agree bit for bit."""
    heads, B, a = 8, 1, 3.5
    C = heads * d
    qv = q(a + 0.05 * rnd(f"attbig{d}.q", (B, Nq, C), -1., 1.), "fp16")
    kv = q(a * (40.0 / d) ** 0.75 + 0.05 * rnd(f"attbig{d}.k", (B, Nk, C), -1., 1.), "fp16")
    vv = q(rnd(f"attbig{d}.v", (B, Nk, C), -1., 1.), "fp16")
    ref = _attn_ref(qv, kv, vv, heads)
    dev = lambda t: t.cuda().half()
    y = ops.attention(dev(qv), dev(kv), dev(vv), heads)
    assert max_rel_err(y.float().cpu(), ref) < TOL["fp16"] * 2
    assert torch.equal(y, ops.attention(dev(qv), dev(kv), dev(vv), heads))


def test_attention_forced_rescale(ops):
    """A key far above the running max late in the sequence forces the online-softmax rescale path."""
    heads, B, d, N = 8, 1, 40, 256
    C = heads * d
    qv = rnd("attspike.q", (B, N, C), -1., 1.)
    kv = rnd("attspike.k", (B, N, C), -1., 1.)
    vv = rnd("attspike.v", (B, N, C), -1., 1.)
    kv[:, 200] = qv[:, 17] * 6.0          # spikes the score of query 17 (and others) at key 200
    for dt in ("fp32", "fp16"):
        qq, kk, vq = q(qv, dt), q(kv, dt), q(vv, dt)
        ref = _attn_ref(qq, kk, vq, heads)
        dev = lambda t: t.cuda().to(tdt(dt))
        for impl in (0, 1):
            y = ops.attention(dev(qq), dev(kk), dev(vq), heads, impl=impl)
            assert max_rel_err(y.float().cpu(), ref) < TOL[dt] * 2


@pytest.mark.parametrize("d", [40, 80, 160])
def test_attention_ramping_scores(ops, d):
    """Scores that keep growing along the key axis: the running maximum moves in many small steps
    (below and above the rescale threshold), through both the ones-column and the VALU row-sum path."""
    heads, B, N = 8, 1, 512
    C = heads * d
    qv = rnd(f"attramp{d}.q", (B, N, C), -1., 1.)
    kv = rnd(f"attramp{d}.k", (B, N, C), -1., 1.) + qv * torch.linspace(0., 1.5, N)[None, :, None]
    vv = rnd(f"attramp{d}.v", (B, N, C), -1., 1.)
    for dt in ("fp16", "bf16"):
        qq, kk, vq = q(qv, dt), q(kv, dt), q(vv, dt)
        ref = _attn_ref(qq, kk, vq, heads)
        dev = lambda t: t.cuda().to(tdt(dt))
        y = ops.attention(dev(qq), dev(kk), dev(vq), heads)
        assert max_rel_err(y.float().cpu(), ref) < TOL[dt] * 2


@pytest.mark.parametrize("N,d,heads", [(77, 64, 12), (200, 64, 4), (333, 40, 8)])
def test_attention_causal(ops, N, d, heads):
    """Causal mask of the text transformer (query i sees keys 0..i), one and several key tiles, both kernels."""
    B, C = 2, heads * d
    qv = rnd(f"attc{N}.q", (B, N, C), -2., 2.)
    kv = rnd(f"attc{N}.k", (B, N, C), -2., 2.)
    vv = rnd(f"attc{N}.v", (B, N, C), -1., 1.)
    mask = torch.full((N, N), float("-inf")).triu(1)
    for dt in ("fp32", "fp16", "bf16"):
        qq, kk, vq = q(qv, dt), q(kv, dt), q(vv, dt)
        sp = lambda t: t.reshape(B, N, heads, d).transpose(1, 2).double()
        w = (sp(qq) @ sp(kk).transpose(-1, -2)) * d ** -0.5 + mask
        ref = (w.softmax(-1) @ sp(vq)).transpose(1, 2).reshape(B, N, C).float()
        dev = lambda t: t.cuda().to(tdt(dt))
        for impl in (0, 1):         # 0: the matrix-core kernel of the dtype, 1: the vector-ALU kernel
            y = ops.attention(dev(qq), dev(kk), dev(vq), heads, impl=impl, causal=True)
            assert max_rel_err(y.float().cpu(), ref) < TOL[dt] * 2, (dt, impl)


def test_to_image_u8_and_inpaint_composite(ops):
    """Bit-exact against the scripts' post-processing (txt2img.py:307,316-317; inpaint_with_text.py:286-291)."""
    x = rnd("u8.x", (2, 3, 24, 40), -1.3, 1.3)
    ref = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
    ref_u8 = (255. * ref.permute(0, 2, 3, 1).numpy()).astype(np.uint8)
    y = ops.to_image_u8(x.cuda()).cpu().numpy()
    assert y.shape == (2, 24, 40, 3) and np.array_equal(y, ref_u8)
    mask = (rnd("u8.mask", (2, 1, 24, 40), 0., 1.) > 0.5).float()
    pred = rnd("u8.pred", (2, 3, 24, 40), 0., 1.)
    comp_ref = (1 - mask) * ref + mask * pred
    comp = ops.inpaint_composite(x.cuda(), mask.cuda(), pred.cuda()).cpu()
    assert torch.equal(comp, comp_ref)


def test_timestep_embedding_g1(ops):
    g = golden("g1_timestep_embedding")
    y = ops.timestep_embedding(g["t"].cuda(), 320).cpu()
    # cos/sin of arguments up to ~1e3: a 1-ulp difference in exp() moves the result by <= 1.2e-4
    assert float((y - g["emb"]).abs().max()) < 2.5e-4
    assert max_rel_err(y[:, :8], g["emb"][:, :8]) < 2.5e-4


def test_small_linear(ops):
    x = rnd("sl.x", (5, 320))
    w = rnd("sl.w", (1280, 320), -0.1, 0.1)
    b = rnd("sl.b", (1280,))
    ref = F.silu(F.linear(F.silu(x), w, b))
    y = ops.small_linear(x.cuda(), w.cuda(), b.cuda(), act_in=1, act_out=1).cpu()
    assert max_rel_err(y, ref) < 2e-6
    x = rnd("sl.x2", (18, 64))
    ref = F.linear(x, w[:, :64].contiguous(), None)
    y = ops.small_linear(x.cuda(), w[:, :64].contiguous().cuda()).cpu()
    assert max_rel_err(y, ref) < 2e-6
    # two rows of a long x (96 KiB parked in LDS, beyond the default dynamic limit) on the narrow-output path
    x = rnd("sl.x3", (2, 12288))
    w3 = rnd("sl.w3", (256, 12288), -0.02, 0.02)
    ref = F.linear(x, w3, None)
    y = ops.small_linear(x.cuda(), w3.cuda()).cpu()
    assert max_rel_err(y, ref) < 1e-5
    y1 = ops.small_linear(x[:1].contiguous().cuda(), w3.cuda()).cpu()
    assert max_rel_err(y1, ref[:1]) < 1e-5


def test_ddim_step_g5_bit_exact(ops):
    """The fused CFG + DDIM update reproduces the reference's fp32 results bit for bit."""
    from oracle import sampler_ref
    g = golden("g5_sampler_kat")
    tab = sampler_ref.ddpm_tables()
    x, noise = g["x"].cuda(), g["noise"].cuda()
    eps = torch.cat([g["e_u"], g["e_c"]]).cuda()
    for eta in (0.0, 0.5):
        d = sampler_ref.ddim_tables(tab["alphas_cumprod"], 50, eta)
        for index in (49, 20, 0):
            xp, p0 = ops.ddim_step(x, eps, True, 7.5, d["alphas"][index], d["alphas_prev"][index],
                                   d["sigmas"][index], d["sqrt_one_minus_alphas"][index], noise=noise)
            assert torch.equal(p0.cpu(), g[f"eta{eta}.i{index}.pred_x0"])
            assert torch.equal(xp.cpu(), g[f"eta{eta}.i{index}.x_prev"])
    d = sampler_ref.ddim_tables(tab["alphas_cumprod"], 50, 0.0)
    for tidx in (36, 49):
        ca = float(torch.sqrt(torch.tensor(d["alphas"][tidx])))
        out = ops.axpby(x, noise, ca, float(d["sqrt_one_minus_alphas"][tidx]))
        assert torch.equal(out.cpu(), g[f"stochastic_encode.t{tidx}"])


def test_ddpm_step_g11_bit_exact(ops):
    """The fused ancestral (DDPM) update reproduces LatentDiffusion.p_sample's fp32 results bit for bit."""
    from oracle import sampler_ref
    g = golden("g11_ddpm_kat")
    tab = sampler_ref.ddpm_tables()
    x, eps, noise = g["x"].cuda(), g["eps"].cuda(), g["noise"].cuda()
    sd = (0.5 * tab["posterior_log_variance_clipped"]).exp()
    for tv in (999, 500, 1, 0):
        for clip in (False, True):
            for temp in (1.0, 0.7):
                xp, x0 = ops.ddpm_step(x, eps, tab["sqrt_recip_alphas_cumprod"][tv],
                                       tab["sqrt_recipm1_alphas_cumprod"][tv], tab["posterior_mean_coef1"][tv],
                                       tab["posterior_mean_coef2"][tv], (0.0 if tv == 0 else 1.0) * float(sd[tv]),
                                       noise=noise, temperature=temp, clip_denoised=clip)
                assert torch.equal(x0.cpu(), g[f"t{tv}.clip{int(clip)}.temp{temp}.x0"])
                assert torch.equal(xp.cpu(), g[f"t{tv}.clip{int(clip)}.temp{temp}.x_prev"])
    # noise pointer omitted at t == 0
    xp, _ = ops.ddpm_step(x, eps, tab["sqrt_recip_alphas_cumprod"][0], tab["sqrt_recipm1_alphas_cumprod"][0],
                          tab["posterior_mean_coef1"][0], tab["posterior_mean_coef2"][0], 0.0, want_pred_x0=False)
    assert torch.equal(xp.cpu(), g["t0.clip0.temp1.0.x_prev"])


# ---- golden block vectors (reference classes' own outputs) -------------------------------------------
def _sd(shapes):
    return synth.synth_state_dict(shapes, 0)


@pytest.mark.parametrize("dt", DTYPES)
def test_g2_norms(ops, dt):
    g = golden("g2_blocks")
    sd = _sd({"g2.gn32.norm.weight": (320,), "g2.gn32.norm.bias": (320,), "g2.normalize.norm.weight": (640,),
              "g2.normalize.norm.bias": (640,), "g2.ln.norm.weight": (320,), "g2.ln.norm.bias": (320,)})
    y = ops.groupnorm(nhwc(g2_u("gn32.x", (2, 320, 8, 8)), dt), sd["g2.gn32.norm.weight"].cuda(),
                      sd["g2.gn32.norm.bias"].cuda(), 1e-5, silu=True)
    assert max_rel_err(nchw(y), g["gn32_silu"]) < TOL[dt]
    y = ops.groupnorm(nhwc(g2_u("normalize.x", (2, 640, 8, 8)), dt), sd["g2.normalize.norm.weight"].cuda(),
                      sd["g2.normalize.norm.bias"].cuda(), 1e-6)
    assert max_rel_err(nchw(y), g["normalize"]) < TOL[dt]
    y = ops.layernorm(g2_u("ln.x", (2, 64, 320)).reshape(128, 320).cuda().to(tdt(dt)), sd["g2.ln.norm.weight"].cuda(),
                      sd["g2.ln.norm.bias"].cuda())
    assert max_rel_err(y.float().cpu().reshape(2, 64, 320), g["layernorm"]) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("dh", [40, 80, 160])
def test_g2_cross_attention_module(ops, dt, dh):
    """CrossAttention.forward (to_q/to_k/to_v -> softmax(qk^T)v -> to_out) composed from the device ops."""
    g = golden("g2_blocks")
    d = 8 * dh
    ctx = g2_u("ctx", (2, 77, 768))
    for kind, cdim, context in (("self", d, None), ("cross", 768, ctx)):
        p = f"g2.{kind}{dh}"
        sd = _sd({f"{p}.to_q.weight": (d, d), f"{p}.to_k.weight": (d, cdim), f"{p}.to_v.weight": (d, cdim),
                  f"{p}.to_out.0.weight": (d, d), f"{p}.to_out.0.bias": (d,)})
        x = g2_u(f"{kind}{dh}.x", (2, 64, d)).reshape(128, d).cuda().to(tdt(dt))
        c = x if context is None else context.reshape(154, 768).cuda().to(tdt(dt))
        pw = lambda k: ops.pack_weight(sd[k].cuda(), dt)
        qv = ops.linear(x, pw(f"{p}.to_q.weight")).view(2, 64, d)
        kv = ops.linear(c, pw(f"{p}.to_k.weight")).view(2, -1, d)
        vv = ops.linear(c, pw(f"{p}.to_v.weight")).view(2, -1, d)
        o = ops.attention(qv, kv, vv, 8).view(128, d)
        y = ops.linear(o, pw(f"{p}.to_out.0.weight"), sd[f"{p}.to_out.0.bias"].cuda())
        assert max_rel_err(y.float().cpu().view(2, 64, d), g[f"{kind}{dh}"]) < TOL[dt] * 2, p


@pytest.mark.parametrize("dt", DTYPES)
def test_g2_feedforward_down_up(ops, dt):
    g = golden("g2_blocks")
    p = "g2.ff"
    sd = _sd({f"{p}.net.0.proj.weight": (2560, 320), f"{p}.net.0.proj.bias": (2560,),
              f"{p}.net.2.weight": (320, 1280), f"{p}.net.2.bias": (320,)})
    x = g2_u("ff.x", (2, 64, 320)).reshape(128, 320).cuda().to(tdt(dt))
    wp, bp = ops.pack_geglu(sd[f"{p}.net.0.proj.weight"].cuda(), sd[f"{p}.net.0.proj.bias"].cuda(), dt)
    h = ops.linear(x, wp, bp, geglu=True)
    y = ops.linear(h, ops.pack_weight(sd[f"{p}.net.2.weight"].cuda(), dt), sd[f"{p}.net.2.bias"].cuda())
    assert max_rel_err(y.float().cpu().view(2, 64, 320), g["ff"]) < TOL[dt] * 2
    sd = _sd({"g2.down.op.weight": (320, 320, 3, 3), "g2.down.op.bias": (320,),
              "g2.up.conv.weight": (320, 320, 3, 3), "g2.up.conv.bias": (320,)})
    y = ops.conv2d(nhwc(g2_u("down.x", (2, 320, 8, 8)), dt), ops.pack_weight(sd["g2.down.op.weight"].cuda(), dt),
                   sd["g2.down.op.bias"].cuda(), stride=2)
    assert max_rel_err(nchw(y), g["down"]) < TOL[dt]
    y = ops.conv2d(nhwc(g2_u("up.x", (2, 320, 8, 8)), dt), ops.pack_weight(sd["g2.up.conv.weight"].cuda(), dt),
                   sd["g2.up.conv.bias"].cuda(), up=True)
    assert max_rel_err(nchw(y), g["up"]) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("name,cin", [("res_a", 320), ("res_b", 640)])
def test_g2_resblock(ops, dt, name, cin):
    """ResBlock._forward composed from the device ops in the executor's order."""
    from collections import OrderedDict
    g = golden("g2_blocks")
    spec = S.ResSpec("g2." + name, cin, 320)
    shapes = OrderedDict()
    S._res_params(spec, 1280, shapes)
    sd = {k: v.cuda() for k, v in _sd(shapes).items()}
    p = spec.prefix
    x = nhwc(g2_u(name + ".x", (2, cin, 8, 8)), dt)
    emb = g2_u("emb", (2, 1280)).cuda()
    e = ops.small_linear(emb, sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"], act_in=1)
    h = ops.groupnorm(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], 1e-5, silu=True)
    h = ops.conv2d(h, ops.pack_weight(sd[p + ".in_layers.2.weight"], dt), sd[p + ".in_layers.2.bias"], rowadd=e)
    h = ops.groupnorm(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], 1e-5, silu=True)
    if cin != 320:
        skip = ops.conv2d(x, ops.pack_weight(sd[p + ".skip_connection.weight"], dt),
                          sd[p + ".skip_connection.bias"], taps=1)
    else:
        skip = x
    y = ops.conv2d(h, ops.pack_weight(sd[p + ".out_layers.3.weight"], dt), sd[p + ".out_layers.3.bias"], resid=skip)
    assert max_rel_err(nchw(y), g[name]) < TOL[dt] * 2


# ---- boundary convolutions (NCHW fp32 <-> NHWC) -------------------------------------------------------
IN_CASES = [   # name, B, H, W, source channels, Cout
    ("unet", 2, 16, 16, (4,), 320),
    ("hybrid", 1, 12, 20, (4, 1, 4), 320),          # W not a multiple of the 16-pixel groups
    ("ragged", 3, 5, 7, (4,), 64),                  # 105 pixels: the last group is short, groups straddle samples
    ("vae_enc", 1, 9, 9, (3,), 128),
    ("vae_dec", 2, 8, 8, (4,), 512),
    ("scalar_shape", 1, 6, 6, (5,), 40),            # no MFMA kernel for this Cin / Cout
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", IN_CASES, ids=[c[0] for c in IN_CASES])
def test_conv_in(ops, dt, case, monkeypatch):
    """First convolution on both kernels (matrix-core and scalar) against F.conv2d in fp32."""
    name, B, H, W, cs, Co = case
    srcs = [rnd(f"cin.{name}.x{i}", (B, c, H, W), -2, 2) for i, c in enumerate(cs)]
    w = rnd(f"cin.{name}.w", (Co, sum(cs), 3, 3), -0.3, 0.3)
    b = rnd(f"cin.{name}.b", (Co,))
    ref = F.conv2d(torch.cat(srcs, 1), w, b, padding=1)
    outs = []
    for scalar in ("0", "1"):
        monkeypatch.setenv("SDMI_BOUNDARY_SCALAR", scalar)
        y = ops.conv_in([s.cuda() for s in srcs], w.cuda(), b.cuda(), dt)
        torch.cuda.synchronize()
        assert max_rel_err(nchw(y), ref) < TOL[dt], (name, dt, scalar)
        outs.append(y)
    # same products, same summation order (bias, taps, channels): the two kernels agree to the last bit
    assert torch.equal(outs[0], outs[1])


OUT_CASES = [  # name, B, H, W, Cin, Cout
    ("unet", 2, 16, 16, 320, 4),
    ("unet_ragged", 3, 5, 7, 320, 4),
    ("vae", 1, 24, 20, 128, 3),
    ("moments", 1, 8, 8, 128, 8),
    ("scalar_shape", 1, 6, 10, 64, 4),
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", OUT_CASES, ids=[c[0] for c in OUT_CASES])
def test_conv_out(ops, dt, case, monkeypatch):
    """Last convolution on both kernels against F.conv2d on the same (dtype-rounded) activations; the weights
    stay fp32 in both (the matrix-core kernel carries them as hi + lo halves)."""
    name, B, H, W, Ci, Co = case
    x = rnd(f"cout.{name}.x", (B, Ci, H, W), -2, 2)
    w = rnd(f"cout.{name}.w", (Co, Ci, 3, 3), -0.05, 0.05)
    b = rnd(f"cout.{name}.b", (Co,))
    ref = F.conv2d(q(x, dt), w, b, padding=1)
    for scalar in ("0", "1"):
        monkeypatch.setenv("SDMI_BOUNDARY_SCALAR", scalar)
        y = ops.conv_out(nhwc(x, dt), w.cuda(), b.cuda())
        torch.cuda.synchronize()
        assert tuple(y.shape) == (B, Co, H, W)
        # inputs are identical, products exact in fp32 up to the 2^-22 weight split: fp32-level agreement in every mode
        assert max_rel_err(y.cpu(), ref) < 2e-5, (name, dt, scalar)


# ---- LayerNorm folded into the Linear behind it --------------------------------------------------------
LNL_CASES = [   # name, rows, K, N, geglu, splits, bias
    ("qkv_small", 128, 320, 960, False, 1, False),           # 128x128 kernel
    ("q_small_edge", 200, 320, 320, False, 1, False),        # row tail inside a tile
    ("ff1_small", 128, 320, 2560, True, 1, True),
    ("qkv_large", 4096, 320, 960, False, 1, False),          # large-tile kernels
    ("ff1_large", 2048, 640, 5120, True, 1, True),
    ("q_d1280", 1024, 1280, 1280, False, None, False),       # the executor's own split choice
    ("splitk", 256, 1280, 1280, False, 4, True),             # slabs + reduction pass apply the fold
    ("ff1_splitk", 128, 1280, 10240, True, 2, True),
]


@pytest.mark.parametrize("dt", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("case", LNL_CASES, ids=[c[0] for c in LNL_CASES])
def test_layernorm_linear(ops, dt, case):
    """Linear(LayerNorm(x)) as one pass: against torch fp32 on the same (dtype-rounded) x.  The rows carry a common
    offset several times their spread, the case in which folding the mean into the epilogue cancels the most."""
    name, rows, K, N, geglu, splits, with_bias = case
    x = q(rnd(f"lnl.{name}.x", (rows, K), -1.5, 1.5) + rnd(f"lnl.{name}.off", (rows, 1), -4.0, 4.0), dt)
    w = rnd(f"lnl.{name}.w", (N, K), -0.05, 0.05)
    b = rnd(f"lnl.{name}.b", (N,)) if with_bias else None
    gamma, beta = rnd(f"lnl.{name}.g", (K,), 0.5, 1.5), rnd(f"lnl.{name}.be", (K,), -0.5, 0.5)
    ref = F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-5), w, b)
    if geglu:
        a, gate = ref.chunk(2, dim=-1)
        ref = a * F.gelu(gate)
    y = ops.layernorm_linear(x.cuda().to(tdt(dt)), w.cuda(), None if b is None else b.cuda(), gamma.cuda(),
                             beta.cuda(), geglu=geglu, splits=splits)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape)
    err = max_rel_err(y.float().cpu(), ref)
    # two-stage reference: the unfused device path (LayerNorm kernel, then the plain GEMM) for comparison
    ln = ops.layernorm(x.cuda().to(tdt(dt)), gamma.cuda(), beta.cuda())
    if geglu:
        wp, bp = ops.pack_geglu(w.cuda(), b.cuda(), dt)
        y2 = ops.linear(ln, wp, bp, geglu=True)
    else:
        y2 = ops.linear(ln, ops.pack_weight(w.cuda(), dt), None if b is None else b.cuda())
    err2 = max_rel_err(y2.float().cpu(), ref)
    print(f"layernorm_linear {name} {dt}: folded {err:.3e}, two-stage {err2:.3e}")
    assert err < TOL[dt] * (2 if geglu else 1), (name, dt, err)
    if dt != "fp32":      # the executor folds in the 16-bit modes only; there it must not cost accuracy
        assert err < 2 * err2 + 1e-6, "the fold must not be less accurate than LayerNorm -> GEMM"
