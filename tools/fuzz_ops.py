#!/usr/bin/env python3
"""Randomised shapes and options for the C-ABI operators against fp64 / fp32 torch arithmetic on the same GPU.
Usage: fuzz_ops.py [seed] [cases per operator].  Prints one line per failing case and a summary; exit code 1
when anything failed.  The planner's tile shape, K order and store path are forced at random through the
library's environment switches (read on every call), so every kernel instantiation gets shapes its own tests do
not list."""
import importlib
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("stable-diffusion_amd.ops")

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
CASES = int(sys.argv[2]) if len(sys.argv) > 2 else 200
TD = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}
TOL = {"fp32": 4e-5, "fp16": 6e-3, "bf16": 4e-2}
rng = random.Random(SEED)
gen = torch.Generator(device="cuda").manual_seed(SEED)
fails = []
seen = {}        # operator -> [cases, worst error / tolerance]


def rnd(shape, lo, hi, dt):
    """values representable in dt, returned as (dt tensor, fp64 tensor of the same values)"""
    x = (torch.rand(shape, device="cuda", generator=gen) * (hi - lo) + lo).to(TD[dt])
    return x, x.double()


def err(y, ref):
    return float((y.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def setenv(name, value):
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)


def check(op, desc, y, ref, tol):
    e = err(y, ref)
    st = seen.setdefault(op, [0, 0.0])
    st[0] += 1
    st[1] = max(st[1], e / tol)
    bad = not (e < tol) or bool(torch.isnan(y.float()).any())
    if bad:
        fails.append((op, desc, e))
        print(f"FAIL {op}: {desc}  err {e:.3e} (tol {tol:.1e})", flush=True)


def fuzz_attention():
    for _ in range(CASES):
        dt = rng.choice(["fp16", "fp16", "bf16", "fp32"])
        d = rng.choice([40, 40, 64, 80, 128, 160])
        heads = rng.choice([1, 2, 5, 8])
        B = rng.randint(1, 3)
        Nq = rng.choice([rng.randint(1, 140), rng.randint(1, 700), rng.randint(2000, 2600)])
        causal = rng.random() < 0.2
        Nk = Nq if causal else rng.choice([rng.randint(1, 140), rng.randint(1, 700), 77])
        amp = rng.choice([0.5, 2.0, 4.0])
        qpw = rng.choice([None, None, 2, 3, 7])
        C = heads * d
        q, qr = rnd((B, Nq, C), -amp, amp, dt)
        k, kr = rnd((B, Nk, C), -amp, amp, dt)
        v, vr = rnd((B, Nk, C), -1, 1, dt)
        sp = lambda t: t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)
        s = sp(qr) @ sp(kr).transpose(-1, -2) * d ** -0.5
        if causal:
            s = s + torch.full((Nq, Nq), float("-inf"), device="cuda", dtype=torch.float64).triu(1)
        ref = (s.softmax(-1) @ sp(vr)).permute(0, 2, 1, 3).reshape(B, Nq, C)
        setenv("SDMI_XATTN_QPW", qpw)
        y = ops.attention(q, k, v, heads, causal=causal)
        setenv("SDMI_XATTN_QPW", None)
        check("attention", f"{dt} d{d} h{heads} B{B} Nq{Nq} Nk{Nk} causal{int(causal)} amp{amp} qpw{qpw}", y, ref,
              TOL[dt])


def conv_ref(x, w, bias, rowadd, resid, taps, stride, up, pad_lo):
    """x (B,H,W,C) fp64, w (O,C,kh,kw) fp64 -> (B,Ho,Wo,O) fp64"""
    xc = x.permute(0, 3, 1, 2)
    if up:
        xc = xc.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    if taps == 9:
        xc = F.pad(xc, (pad_lo, 1, pad_lo, 1))
        cols = F.unfold(xc, 3, stride=stride)
        Ho = (xc.shape[2] - 3) // stride + 1
        Wo = (xc.shape[3] - 3) // stride + 1
    else:
        xc = xc[:, :, ::stride, ::stride]
        Ho, Wo = xc.shape[2], xc.shape[3]
        cols = xc.reshape(xc.shape[0], xc.shape[1], -1)
    out = (w.reshape(w.shape[0], -1) @ cols).reshape(x.shape[0], w.shape[0], Ho, Wo).permute(0, 2, 3, 1)
    if bias is not None:
        out = out + bias
    if rowadd is not None:
        out = out + rowadd[:, None, None, :]
    if resid is not None:
        out = out + resid
    return out


def fuzz_conv():
    for _ in range(CASES):
        dt = rng.choice(["fp16", "fp16", "bf16", "fp32"])
        bk = 32 if dt == "fp32" else 64
        taps = rng.choice([1, 9])
        C0 = bk * rng.randint(1, 5)
        C1 = bk * rng.randint(1, 3) if rng.random() < 0.3 else 0
        N = rng.choice([4 * rng.randint(1, 50), 160 * rng.randint(1, 4), 256 * rng.randint(1, 2), 320 * rng.randint(1, 2),
                        128 * rng.choice([1, 3])])
        B = rng.randint(1, 3)
        H, W = rng.randint(1, 36), rng.randint(1, 36)
        stride, up, pad_lo = 1, False, 1
        if taps == 9:
            mode = rng.choice(["plain", "plain", "s2", "s2pad0", "up"])
            if mode == "s2":
                stride = 2
            elif mode == "s2pad0":
                stride, pad_lo = 2, 0
                H, W = max(H, 2), max(W, 2)
            elif mode == "up":
                up = True
                H, W = min(H, 20), min(W, 20)
        splits = rng.choice([1, 1, 2, 3, 5, None])
        shape = rng.choice([None, None, 1, 2, 3, 4, 6, 7])
        tap_inner = rng.choice([None, 0, 1])
        direct = rng.choice([None, None, 1])
        x0, x0r = rnd((B, H, W, C0), -1, 1, dt)
        x1 = x1r = None
        if C1:
            x1, x1r = rnd((B, H, W, C1), -1, 1, dt)
        w, wr = rnd((N, C0 + C1, 3, 3) if taps == 9 else (N, C0 + C1), -0.1, 0.1, dt)
        wp = ops.pack_weight(w.float(), dt)
        bias = torch.rand(N, device="cuda", generator=gen) - 0.5 if rng.random() < 0.7 else None
        rowadd = torch.rand(B, N, device="cuda", generator=gen) - 0.5 if rng.random() < 0.3 else None
        xr = torch.cat([x0r, x1r], dim=3) if C1 else x0r
        ref = conv_ref(xr, wr if taps == 9 else wr[:, :, None, None], bias.double() if bias is not None else None,
                       rowadd.double() if rowadd is not None else None, None, taps, stride, up, pad_lo)
        resid = residr = None
        if rng.random() < 0.4:
            resid, residr = rnd(tuple(ref.shape), -1, 1, dt)
            ref = ref + residr
        setenv("SDMI_GEMM_SHAPE", shape)
        setenv("SDMI_GEMM_TAP_INNER", tap_inner)
        setenv("SDMI_GEMM_DIRECT_STORE", direct)
        desc = (f"{dt} taps{taps} B{B} {H}x{W} C{C0}+{C1} N{N} stride{stride} up{int(up)} pad{pad_lo} splits{splits} "
                f"shape{shape} ti{tap_inner} direct{direct} bias{int(bias is not None)} rowadd{int(rowadd is not None)} "
                f"resid{int(resid is not None)}")
        try:
            y = ops.conv2d(x0, wp, bias, taps=taps, stride=stride, up=up, x1=x1, rowadd=rowadd, resid=resid,
                           splits=splits, pad_lo=pad_lo)
            check("conv2d", desc, y, ref, TOL[dt])
        except RuntimeError as e:
            fails.append(("conv2d", desc, str(e)))
            print(f"ERROR conv2d: {desc}: {e}", flush=True)
        for name in ("SDMI_GEMM_SHAPE", "SDMI_GEMM_TAP_INNER", "SDMI_GEMM_DIRECT_STORE"):
            setenv(name, None)


def fuzz_norms():
    for _ in range(CASES):
        dt = rng.choice(["fp16", "fp16", "bf16", "fp32"])
        C = rng.choice([128, 256, 320, 384, 512, 640, 960, 1280, 1920, 2560])
        B = rng.randint(1, 17)
        hw = rng.choice([rng.randint(1, 80), rng.randint(1, 1200), rng.randint(3000, 5000)])
        if B * hw * C > 40_000_000:
            B = max(1, 40_000_000 // (hw * C))
        silu = rng.random() < 0.6
        c0 = C
        if rng.random() < 0.3:
            c0 = 64 * rng.randint(1, C // 64 - 1) if C > 64 else C
        x, xr = rnd((B, hw, 1, C), -2, 3, dt)
        xa = x[..., :c0].contiguous()
        xb = x[..., c0:].contiguous() if c0 < C else None
        g = torch.rand(C, device="cuda", generator=gen) + 0.5
        b = torch.rand(C, device="cuda", generator=gen) - 0.5
        xg = xr.reshape(B, hw, 32, C // 32)
        mean = xg.mean(dim=(1, 3), keepdim=True)
        var = xg.var(dim=(1, 3), unbiased=False, keepdim=True)
        ref = ((xg - mean) / (var + 1e-5).sqrt()).reshape(B, hw, 1, C) * g.double() + b.double()
        if silu:
            ref = ref * torch.sigmoid(ref)
        y = ops.groupnorm(xa, g, b, 1e-5, silu, x1=xb)
        check("groupnorm", f"{dt} B{B} HW{hw} C{c0}+{C - c0} silu{int(silu)}", y, ref, TOL[dt])
    for _ in range(CASES // 2):
        dt = rng.choice(["fp16", "bf16", "fp32"])
        d = rng.choice([320, 640, 768, 1280])
        rows = rng.choice([rng.randint(1, 100), rng.randint(1, 6000)])
        x, xr = rnd((rows, d), -2, 3, dt)
        g = torch.rand(d, device="cuda", generator=gen) + 0.5
        b = torch.rand(d, device="cuda", generator=gen) - 0.5
        ref = F.layer_norm(xr, (d,), g.double(), b.double(), 1e-5)
        check("layernorm", f"{dt} rows{rows} d{d}", ops.layernorm(x, g, b), ref, TOL[dt])


def fuzz_layernorm_linear():
    for _ in range(CASES // 2):
        dt = rng.choice(["fp16", "fp16", "bf16", "fp32"])
        K = rng.choice([320, 640, 1280])
        geglu = rng.random() < 0.3
        N = 8 * K if geglu else rng.choice([K, 3 * K, 160, 4 * rng.randint(1, 60)])
        rows = rng.choice([rng.randint(1, 300), rng.randint(1, 3000), 256 * rng.randint(1, 20)])
        if geglu:
            rows = min(rows, 1024)
        splits = rng.choice([1, 1, 2, None])
        shape = rng.choice([None, None, 1, 2, 3, 4, 7])
        x, xr = rnd((rows, K), -2, 3, dt)
        w = (torch.rand(N, K, device="cuda", generator=gen) - 0.5) * 0.1
        bias = torch.rand(N, device="cuda", generator=gen) - 0.5
        g = torch.rand(K, device="cuda", generator=gen) + 0.5
        b = torch.rand(K, device="cuda", generator=gen) - 0.5
        # the kernel multiplies dt-rounded (w * gamma): take that as the reference weight as well
        wq = (w * g).to(TD[dt]).double() / g.double()
        ref = F.layer_norm(xr, (K,), g.double(), b.double(), 1e-5) @ wq.t() + bias.double()
        if geglu:
            val, gate = ref.chunk(2, dim=-1)
            ref = val * F.gelu(gate)
        setenv("SDMI_GEMM_SHAPE", shape)
        desc = f"{dt} rows{rows} K{K} N{N} geglu{int(geglu)} splits{splits} shape{shape}"
        try:
            y = ops.layernorm_linear(x, w, bias, g, b, 1e-5, geglu=geglu, splits=splits)
            check("layernorm_linear", desc, y, ref, 2 * TOL[dt])
        except RuntimeError as e:
            fails.append(("layernorm_linear", desc, str(e)))
            print(f"ERROR layernorm_linear: {desc}: {e}", flush=True)
        setenv("SDMI_GEMM_SHAPE", None)


if __name__ == "__main__":
    for f in (fuzz_attention, fuzz_conv, fuzz_norms, fuzz_layernorm_linear):
        n0 = len(fails)
        f()
        torch.cuda.synchronize()
        print(f"{f.__name__}: {len(fails) - n0} failure(s)", flush=True)
    for op, (n, worst) in seen.items():
        print(f"  {op}: {n} cases, worst error = {worst:.2f} of its tolerance")
    print(f"seed {SEED}, {CASES} cases per operator: {len(fails)} failure(s)")
    sys.exit(1 if fails else 0)
