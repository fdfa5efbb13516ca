#!/usr/bin/env python3
"""The software-pipelined attention kernel (csrc/attention_pipe.hip) against the 32-query kernels it replaces, on the two
shapes of the dense-feature path: the ViT's self-attention (batch 32 x 6 heads x 1025 tokens, head_dim 64, packed qkv) and
LoftUp's cross-attention (batch 8 x 4 heads, 200 704 pixel queries x 1024 LR keys, head_dim 101 padded to 128).
usage: att_pipe_bench.py [iters]   -- prints microseconds per launch and TFLOP/s (algorithmic FLOPs: 4 Lq Lk hd per head)"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, ".")


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    from isegprobe_amd import hip_ops as ops
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    tag = "pipelined" if os.environ.get("ISEGPROBE_ATT_PIPE", "0") == "1" else "32-query"
    torch.manual_seed(0)
    for dt in (torch.bfloat16, torch.float16):
        B, L, heads = 32, 1025, 6
        qkv = (torch.randn(B * L, 3 * heads * 64, device="cuda") * 0.5).to(dt)
        us = timed(lambda: ops.attention_packed_qkv(qkv, B, L, heads, None, q_logit2=True), iters)
        print(f"[{tag}] ViT self-attention {dt}: B={B} L={L} heads={heads} hd=64: {us:.1f} us  {4.0 * B * heads * L * L * 64 / us / 1e6:.0f} TFLOP/s", flush=True)
        del qkv
        B, Lq, Lk, heads, hd, hdp = 8, 448 * 448, 1024, 4, 101, 128
        q = (torch.randn(B, Lq, heads, hdp, device="cuda") * 0.1).to(dt)
        k = torch.randn(B, Lk, heads, hdp, device="cuda").to(dt)
        v = torch.randn(B, Lk, heads, hdp, device="cuda").to(dt)
        us = timed(lambda: ops.attention(q, k, v, None, q_logit2=True), max(3, iters // 6))
        print(f"[{tag}] LoftUp cross-attention {dt}: B={B} Lq={Lq} Lk={Lk} heads={heads} hd={hd}->{hdp}: {us / 1e3:.3f} ms  "
              f"{4.0 * B * heads * Lq * Lk * hd / us / 1e6:.0f} TFLOP/s algorithmic ({4.0 * B * heads * Lq * Lk * hdp / us / 1e6:.0f} executed)", flush=True)
        del q, k, v


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[-1] == "--both":  # the two kernels in separate processes (the switch is read once per process)
        sys.argv.pop()
        for v in ("1", "0"):
            subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, ISEGPROBE_ATT_PIPE=v), check=True)
    else:
        main()
