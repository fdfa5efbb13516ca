#!/usr/bin/env python3
"""Shader-clock stamps of the pipelined attention kernel (ISP_PIPE_ABL & 128 builds): block 0 / wave 0."""
import glob, os, subprocess, sys
code = r'''
import sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
torch.manual_seed(0)
for (B, L, heads, hd) in ((32, 1024, 6, 64), (2, 1024, 4, 128)):
    if hd == 64:
        qkv = (torch.randn(B * L, 3 * heads * 64, device="cuda") * 0.5).to(torch.bfloat16)
        for _ in range(3): out = ops.attention_packed_qkv(qkv, B, L, heads, None, q_logit2=True)
    else:
        q = (torch.randn(B, 448 * 448, heads, hd, device="cuda") * 0.1).to(torch.bfloat16)
        k = torch.randn(B, L, heads, hd, device="cuda").to(torch.bfloat16)
        v = torch.randn(B, L, heads, hd, device="cuda").to(torch.bfloat16)
        for _ in range(2): out = ops.attention(q, k, v, None, q_logit2=True)
    torch.cuda.synchronize()
    st = out.reshape(-1)[:32].view(torch.int64).cpu().tolist()
    t0 = st[0]
    print(f"hd{hd}: start->prologue-wait {st[1]-t0}, QK(0)+rebase {st[2]-st[1]}, loop {st[3]-st[2]} (phase A sum {st[5]}, phase B sum {st[6]}), tail {st[4]-st[3]}, epilogue+stores {st[7]-st[4]}, total {st[7]-t0} cycles")
'''
for lib in sorted(glob.glob("build_variants/lib_abl*.so")):
    env = dict(os.environ, ISEGPROBE_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), "\n  ".join([""] + r.stdout.strip().splitlines()), r.stderr.strip().splitlines()[-1:] if r.returncode else "", flush=True)
