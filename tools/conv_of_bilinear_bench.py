"""A/B of the head's first convolution: resize + conv3x3 (the route of rounds 1-3) against the low-resolution GEMM + blend
(csrc/conv_bilinear.hip).  python tools/conv_of_bilinear_bench.py [B h w H W C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isegprobe_amd import hip_ops as ops  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    cases = [(32, 32, 32, 448, 448, 384), (2, 128, 128, 896, 896, 1024), (2, 64, 64, 448, 448, 384), (8, 32, 32, 448, 448, 768)]
    if len(sys.argv) == 7:
        cases = [tuple(int(v) for v in sys.argv[1:])]
    for B, h, w, H, W, C in cases:
        N = C
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, h, w, C, generator=g).cuda().half()
        wconv = (torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
        bias = (0.1 * torch.randn(N, generator=g)).cuda()
        wt = wconv.permute(0, 2, 3, 1).reshape(N, 9 * C).half().contiguous()
        wz = wconv.permute(2, 3, 0, 1).reshape(9 * N, C).half().contiguous()
        xb = x.to(torch.bfloat16)

        def old():
            y = ops.to_f16(ops.resize_nhwc(xb, H, W, "bilinear"))
            return ops.conv3x3(y, wt, bias, "relu")

        def gemm():
            return ops.linear(x.view(-1, C), wz)

        z = gemm()

        def blend():
            return ops.conv3x3_of_bilinear_blend(z, bias, B, h, w, H, W, N)

        def new():
            return ops.conv3x3_of_bilinear_blend(ops.linear(x.view(-1, C), wz), bias, B, h, w, H, W, N)

        a, b_ = old().float(), new().float()
        err = (a - b_).abs().max().item()
        t_old, t_gemm, t_blend, t_new = timed(old), timed(gemm), timed(blend), timed(new)
        out_gb = B * H * W * N * 2 / 1e9
        print(f"B{B} {h}x{w}->{H}x{W} C{C}: resize+conv {t_old:.2f} ms | gemm {t_gemm:.3f} + blend {t_blend:.3f} = {t_new:.3f} ms "
              f"({t_old / t_new:.1f}x)  blend writes {out_gb:.2f} GB -> {out_gb / t_blend * 1e3:.0f} GB/s   max|diff| {err:.3g}", flush=True)


if __name__ == "__main__":
    main()
