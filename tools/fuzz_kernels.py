#!/usr/bin/env python3
"""Random-shape sweep of the main kernels against torch fp32 (GPU box): conv3x3 (bf16 / f16, plain + fused epilogues), dense
GEMM (bf16 / f16), ViT self-attention (both entry points), generic attention (bf16 / f16), LayerNorm dtypes.
usage: fuzz_kernels.py [seed] [rounds]   -- prints one line per case, exits non-zero on the first mismatch."""
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
g = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
pick = lambda xs: xs[ri(0, len(xs) - 1)]
BF, H16 = torch.bfloat16, torch.float16
bad = 0


def check(name, got, ref, tol):
    global bad
    err = (got.float() - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    ok = err <= tol * scale and torch.isfinite(got.float()).all().item()
    print(f"{'ok  ' if ok else 'FAIL'} {name}: max err {err:.3g} (scale {scale:.3g}, tol {tol * scale:.3g})", flush=True)
    bad += 0 if ok else 1


for r in range(rounds):
    torch.manual_seed(seed * 1000 + r)
    # ---- conv3x3
    B, H, W = ri(1, 3), ri(5, 90), ri(5, 90)
    C, N = pick([64, 128, 192, 256, 384]), pick([64, 128, 192, 320, 384, 448, 768])
    x = torch.randn(B, C, H, W, device="cuda")
    w = torch.randn(N, C, 3, 3, device="cuda") / math.sqrt(9 * C)
    bias = torch.randn(N, device="cuda")
    for dt, tol in ((BF, 2e-2), (H16, 4e-3)):
        if dt == H16 and not ops.conv_takes_f16(N):
            continue
        xd, wd = x.to(dt), w.to(dt)
        ref = F.relu(F.conv2d(xd.float(), wd.float(), bias, padding=1))
        y = ops.conv3x3(xd.permute(0, 2, 3, 1).contiguous(), wd.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous(), bias, "relu")
        check(f"conv3x3 {dt} B{B} {H}x{W} C{C} N{N}", y.permute(0, 3, 1, 2), ref, tol)
        if N % 192 == 0 and (B * H * W) % 4 == 0:
            wc = torch.randn(N, device="cuda") / math.sqrt(N)
            z = ops.conv3x3_relu_classifier(xd.permute(0, 2, 3, 1).contiguous(), wd.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous(), bias, wc, 0.1)
            check(f"conv+classifier {dt} B{B} {H}x{W} C{C} N{N}", z, (ref * wc.view(1, N, 1, 1)).sum(1) + 0.1, tol)
    # ---- dense GEMM
    M, K, N = ri(1, 5000), 64 * ri(1, 12), 4 * ri(1, 200)
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / math.sqrt(K)
    bias = torch.randn(N, device="cuda")
    for dt, tol in ((BF, 2e-2), (H16, 4e-3)):
        for act, fn in ((None, lambda t: t), ("gelu", F.gelu)):
            y = ops.linear(a.to(dt), w.to(dt), bias, act)
            check(f"linear {dt} {act} M{M} K{K} N{N}", y, fn(a.to(dt).float() @ w.to(dt).float().t() + bias), tol)
        res = torch.randn(M, N, device="cuda").to(dt)
        y = ops.linear_axpy_res(a.to(dt), w.to(dt), bias, res, 0.7)
        check(f"axpy_res {dt} M{M} K{K} N{N}", y, res.float() + 0.7 * (a.to(dt).float() @ w.to(dt).float().t() + bias), tol * 1.5)
    # ---- ViT self-attention, packed qkv
    Bq, L, heads = ri(1, 3), ri(1, 1300), ri(1, 6)
    D = heads * 64
    qkv = torch.randn(Bq * L, 3 * D, device="cuda").to(BF)
    q, k, v = qkv.float().view(Bq, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = ((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(Bq * L, D)
    check(f"attention64 scale B{Bq} L{L} h{heads}", ops.attention_packed_qkv(qkv, Bq, L, heads, 0.125), ref, 2e-2)
    q2 = qkv.clone()
    q2[:, :D] = (qkv[:, :D].float() * ops.ATTENTION_LOGIT2_SCALE).to(BF)
    qq = q2[:, :D].float().view(Bq, L, heads, 64).permute(0, 2, 1, 3)
    ref2 = ((qq @ k.transpose(-2, -1)) * math.log(2.0)).softmax(-1) @ v
    check(f"attention64 logit2 B{Bq} L{L} h{heads}", ops.attention_packed_qkv(q2, Bq, L, heads, None, q_logit2=True),
          ref2.transpose(1, 2).reshape(Bq * L, D), 2e-2)
    # ---- generic attention (cross): hd 64 / 128 / 256
    hd = pick([64, 128, 256])
    Lq, Lk, Hh = ri(1, 900), ri(1, 400), ri(1, 3)
    q, k, v = (torch.randn(2, n, Hh, hd, device="cuda") for n in (Lq, Lk, Lk))
    for dt, tol in ((BF, 2e-2), (H16, 4e-3)):
        o = ops.attention(q.to(dt), k.to(dt), v.to(dt), hd ** -0.5)
        p = ((q.to(dt).float().permute(0, 2, 1, 3) @ k.to(dt).float().permute(0, 2, 3, 1)) * hd ** -0.5).softmax(-1)
        check(f"attention {dt} hd{hd} Lq{Lq} Lk{Lk} H{Hh}", o, (p @ v.to(dt).float().permute(0, 2, 1, 3)).permute(0, 2, 1, 3), tol)
    # ---- base-2-logit queries (deferred-maximum kernels at head_dim 64 / 128; generic kernel at 256), with a spike row
    hd2 = pick([64, 128, 256])
    q2l = q if hd2 == hd else torch.randn(2, Lq, Hh, hd2, device="cuda")
    k2l, v2l = (k, v) if hd2 == hd else (torch.randn(2, Lk, Hh, hd2, device="cuda"), torch.randn(2, Lk, Hh, hd2, device="cuda"))
    q2l = q2l * (hd2 ** -0.5 * math.log2(math.e))
    if Lq > 3:
        q2l[0, ri(0, Lq - 1)] *= 40.0  # logits far above the deferred-maximum threshold
    for dt, tol in ((BF, 2e-2), (H16, 4e-3)):
        o = ops.attention(q2l.to(dt), k2l.to(dt), v2l.to(dt), None, q_logit2=True)
        p = ((q2l.to(dt).float().permute(0, 2, 1, 3) @ k2l.to(dt).float().permute(0, 2, 3, 1)) * math.log(2.0)).softmax(-1)
        check(f"attention logit2 {dt} hd{hd2} Lq{Lq} Lk{Lk} H{Hh}", o, (p @ v2l.to(dt).float().permute(0, 2, 1, 3)).permute(0, 2, 1, 3), tol)
    # ---- LayerNorm folded into the next GEMM: residual GEMM with row statistics -> LN-folded GEMM (LoftUp's half stream)
    Dl = 4 * ri(8, 112)            # channels the LayerNorm runs over
    cpad = (Dl + 63) // 64 * 64    # padded row width (zero columns behind Dl)
    Ml, Kl, Nl = ri(1, 4000), 64 * ri(1, 8), 4 * ri(1, 128)
    al = torch.randn(Ml, Kl, device="cuda").to(H16)
    wo = torch.zeros(cpad, Kl, device="cuda"); wo[:Dl] = torch.randn(Dl, Kl, device="cuda") / math.sqrt(Kl)
    bo = torch.zeros(cpad, device="cuda"); bo[:Dl] = torch.randn(Dl, device="cuda")
    res = torch.zeros(Ml, cpad, device="cuda"); res[:, :Dl] = torch.randn(Ml, Dl, device="cuda") * 2 + 0.5
    res = res.to(H16)
    xo, st = ops.linear_axpy_res_stats(al, wo.to(H16), bo, res, 1.0)
    xr = res.float() + al.float() @ wo.to(H16).float().t() + bo
    check(f"axpy_res_stats M{Ml} K{Kl} N{cpad}", xo, xr, 4e-3 * 1.5)
    gl, bl = torch.randn(Dl, device="cuda"), torch.randn(Dl, device="cuda")
    w2 = torch.randn(Nl, Dl, device="cuda") / math.sqrt(Dl)
    c2 = torch.randn(Nl, device="cuda")
    wf = torch.zeros(Nl, cpad, device="cuda"); wf[:, :Dl] = w2 * gl[None, :]
    wf = wf.to(H16)
    for act, fn in ((None, lambda t: t), ("gelu", F.gelu)):
        y = ops.linear_lnfold(xo, st, wf, wf.float().sum(1), c2 + w2 @ bl, Dl, 1e-5, act)
        ref = fn(F.layer_norm(xo.float()[:, :Dl], (Dl,), gl, bl, 1e-5) @ w2.t() + c2)
        check(f"lnfold {act} M{Ml} D{Dl}/{cpad} N{Nl}", y, ref, 8e-3)
    # ---- LayerNorm dtypes
    rows, Dn = ri(1, 3000), 4 * ri(4, 300)
    xx = torch.randn(rows, Dn, device="cuda") * 3 + 1
    gg, bb = torch.randn(Dn, device="cuda"), torch.randn(Dn, device="cuda")
    for din in (torch.float32, BF, H16):
        for dout, tol in ((BF, 3e-2), (H16, 4e-3), (torch.float32, 1e-4)):
            if (din, dout) in ((H16, BF), (H16, torch.float32)):
                continue
            y = ops.layernorm(xx.to(din).contiguous(), gg, bb, 1e-5, out_dtype=dout)
            check(f"layernorm {din}->{dout} rows{rows} D{Dn}", y, F.layer_norm(xx.to(din).float(), (Dn,), gg, bb, 1e-5), tol)
    # ---- first head conv through the bilinear resize (random geometry with up-scaling >= 6: 5 x 5 footprint at most), exact-fp32 attention
    hb, wb = ri(2, 20), ri(2, 20)
    Hb, Wb = hb * ri(6, 15) + ri(0, 5), wb * ri(6, 15) + ri(0, 5)
    Cb, Nb = pick([64, 128, 384]), pick([64, 128, 192, 384])
    if ops.conv3x3_of_bilinear_supported(hb, wb, Hb, Wb, Nb):
        Bb = ri(1, 2)
        xb = torch.randn(Bb, hb, wb, Cb, device="cuda")
        wb4 = torch.randn(Nb, Cb, 3, 3, device="cuda") / math.sqrt(9 * Cb)
        bb_ = torch.randn(Nb, device="cuda") * 0.2
        yb = F.interpolate(xb.half().float().permute(0, 3, 1, 2), size=(Hb, Wb), mode="bilinear", align_corners=True)
        refb = F.relu(F.conv2d(yb, wb4.half().float(), bb_, padding=1)).permute(0, 2, 3, 1)
        wz = wb4.permute(2, 3, 0, 1).reshape(9 * Nb, Cb).half().contiguous()
        zb = ops.linear(xb.half().view(-1, Cb), wz)
        check(f"conv_of_bilinear f16 B{Bb} {hb}x{wb}->{Hb}x{Wb} C{Cb} N{Nb}", ops.conv3x3_of_bilinear_blend(zb, bb_, Bb, hb, wb, Hb, Wb, Nb), refb, 3e-3)
        z32 = (xb.view(-1, Cb) @ wb4.permute(2, 3, 0, 1).reshape(9 * Nb, Cb).t()).contiguous()
        y32 = F.interpolate(xb.permute(0, 3, 1, 2), size=(Hb, Wb), mode="bilinear", align_corners=True)
        check(f"conv_of_bilinear f32 {hb}x{wb}->{Hb}x{Wb}", ops.conv3x3_of_bilinear_blend(z32, bb_, Bb, hb, wb, Hb, Wb, Nb, out_dtype=torch.float32),
              F.relu(F.conv2d(y32, wb4, bb_, padding=1)).permute(0, 2, 3, 1), 3e-5)
    Ba, La, ha = ri(1, 3), ri(1, 700), ri(1, 6)
    qkv = torch.randn(Ba * La, 3 * ha * 64, device="cuda")
    q_, k_, v_ = (qkv.double().view(Ba, La, 3, ha, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    refa = (torch.softmax((q_ * 0.125) @ k_.transpose(-1, -2), dim=-1) @ v_).permute(0, 2, 1, 3).reshape(Ba * La, ha * 64).float()
    check(f"attention_f32 B{Ba} L{La} heads{ha}", ops.attention_packed_qkv_f32(qkv, Ba, La, ha, 0.125), refa, 3e-6)
print(f"{bad} mismatches")
sys.exit(1 if bad else 0)
