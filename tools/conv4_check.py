#!/usr/bin/env python3
"""8-wave (ISEGPROBE_CONV_ENGINE=8) vs 4-wave patch conv (the default): outputs must be bit-identical; times both."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
torch.manual_seed(0)
tag = os.environ.get("ISEGPROBE_CONV_ENGINE", "8")
res = []
for (B, H, W, C) in ((2, 37, 45, 384), (1, 16, 16, 128), (3, 50, 33, 192)):
    x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(384, 9 * C, device="cuda") / 60).to(torch.bfloat16)
    b = torch.randn(384, device="cuda")
    y = ops.conv3x3(x, w, b, "relu").cpu()
    f = "/tmp/conv4_ref_%d_%d_%d_%d.pt" % (B, H, W, C)
    if tag == "8": torch.save(y, f); res.append("ref")
    else: res.append(str(torch.equal(torch.load(f), y)))
for relu_in in (False, True):
    x = torch.randn(8, 448, 448, 384, device="cuda")
    x = (x.relu() if relu_in else x).to(torch.bfloat16)
    w = (torch.randn(384, 9 * 384, device="cuda") / 60).to(torch.bfloat16)
    b = torch.randn(384, device="cuda")
    for _ in range(3): y = ops.conv3x3(x, w, b, "relu")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): y = ops.conv3x3(x, w, b, "relu")
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    res.append(f"relu_in={relu_in}: {ms:.3f} ms {2*8*448*448*384*9*384/ms/1e9:.0f} TF")
    f = "/tmp/conv4_big_%d.pt" % relu_in
    if tag == "8": torch.save(y[0, :64].cpu(), f)
    else: res.append("big_same=" + str(torch.equal(torch.load(f), y[0, :64].cpu())))
print(tag, " | ".join(res))
'''
for eng in ("8", "4", "8", "4"):
    env = dict(os.environ, ISEGPROBE_CONV_ENGINE=eng)  # "8": the 8-wave patch kernel, "4": the one-wave-per-SIMD kernel (default)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print((r.stdout.strip().splitlines() or ["?"])[-1], r.stderr.strip().splitlines()[-2:] if r.returncode else "", flush=True)
