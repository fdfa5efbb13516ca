#!/usr/bin/env python3
"""A/B the conv tile configurations: one subprocess per variant .so (env ISEGPROBE_HIP_LIB)."""
import glob, os, subprocess, sys
code = r'''
import sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
B = 8
torch.manual_seed(0)
xs = torch.randn(2, 384, 37, 45, device="cuda").to(torch.bfloat16)
ws = (torch.randn(384, 384, 3, 3, device="cuda") / 60).to(torch.bfloat16)
bs = torch.randn(384, device="cuda")
refs = torch.relu(torch.nn.functional.conv2d(xs.float(), ws.float(), bs, padding=1))
ys = ops.conv3x3(xs.permute(0, 2, 3, 1).contiguous(), ws.permute(0, 2, 3, 1).reshape(384, -1).contiguous(), bs, "relu")
err = (ys.permute(0, 3, 1, 2).float() - refs).abs().max().item() / refs.abs().max().item()
x = torch.randn(B, 448, 448, 384, device="cuda").to(torch.bfloat16)
w = (torch.randn(384, 9 * 384, device="cuda") / 60).to(torch.bfloat16)
bias = torch.randn(384, device="cuda")
ref = None
for _ in range(3): y = ops.conv3x3(x, w, bias, "relu")
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): y = ops.conv3x3(x, w, bias, "relu")
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"{ms:.3f} ms  {2*B*448*448*384*9*384/ms/1e9:.1f} TFLOP/s  relerr {err:.2e}")
'''
for lib in sorted(glob.glob("build_variants/lib_*.so")):
    env = dict(os.environ, ISEGPROBE_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), (r.stdout.strip().splitlines() or ["?"])[-1], r.stderr.strip().splitlines()[-1:] if r.returncode else "")
