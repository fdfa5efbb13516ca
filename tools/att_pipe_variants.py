#!/usr/bin/env python3
"""Timing of the pipelined attention kernel's ablation builds (build_variants/lib_abl*.so, ISP_PIPE_ABL bit mask: 1 no exp/pack,
2 no max/sum, 4 no tile DMA, 8 no PV MFMAs, 16 no QK MFMAs, 32 no barrier/wait, 64 no fragment reads): one subprocess each."""
import glob, os, subprocess, sys
code = r'''
import sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
torch.manual_seed(0)
B, L, heads = 32, 1024, 6          # 1024: no straggler launch in the number
qkv = (torch.randn(B * L, 3 * heads * 64, device="cuda") * 0.5).to(torch.bfloat16)
a = timed(lambda: ops.attention_packed_qkv(qkv, B, L, heads, None, q_logit2=True), 30)
del qkv
B, Lq, Lk, heads, hdp = 2, 448 * 448, 1024, 4, 128
q = (torch.randn(B, Lq, heads, hdp, device="cuda") * 0.1).to(torch.bfloat16)
k = torch.randn(B, Lk, heads, hdp, device="cuda").to(torch.bfloat16)
v = torch.randn(B, Lk, heads, hdp, device="cuda").to(torch.bfloat16)
b = timed(lambda: ops.attention(q, k, v, None, q_logit2=True), 5)
print(f"hd64 B32 L1024: {a:.1f} us ({4.0*32*6*1024*1024*64/a/1e6:.0f} TF)   hd128 B2 448^2x1024: {b:.0f} us ({4.0*2*4*Lq*Lk*128/b/1e6:.0f} TF executed)")
'''
for lib in sorted(glob.glob("build_variants/lib_abl*.so")):
    env = dict(os.environ, ISEGPROBE_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), (r.stdout.strip().splitlines() or ["?"])[-1], r.stderr.strip().splitlines()[-1:] if r.returncode else "", flush=True)
