#!/usr/bin/env python3
"""8-wave (ISEGPROBE_CONV_ENGINE=8) vs 4-wave patch conv with 128-channel blocks (N = 448 ragged, N = 1024)."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
torch.manual_seed(0)
tag = os.environ.get("ISEGPROBE_CONV_ENGINE", "4")
res = []
for (B, H, W, C, N) in ((2, 37, 45, 448, 448), (1, 16, 16, 128, 128), (3, 50, 33, 192, 256), (1, 64, 64, 1024, 1024)):
    x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, 9 * C, device="cuda") / 60).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    y = ops.conv3x3(x, w, b, "relu").cpu()
    f = "/tmp/conv4b_ref_%d_%d_%d_%d_%d.pt" % (B, H, W, C, N)
    if tag == "8": torch.save(y, f); res.append("ref")
    else: res.append(str(torch.equal(torch.load(f), y)))
for (B, S, C, N) in ((8, 224, 448, 448), (2, 256, 1024, 1024)):
    x = torch.randn(B, S, S, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, 9 * C, device="cuda") / 60).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    for _ in range(3): y = ops.conv3x3(x, w, b, "relu")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): y = ops.conv3x3(x, w, b, "relu")
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    res.append(f"C={C} N={N}: {ms:.3f} ms {2*B*S*S*C*9*N/ms/1e9:.0f} TF")
print(tag, " | ".join(res))
'''
for eng in ("8", "4", "8", "4"):
    env = dict(os.environ, ISEGPROBE_CONV_ENGINE=eng)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print((r.stdout.strip().splitlines() or ["?"])[-1], r.stderr.strip().splitlines()[-2:] if r.returncode else "", flush=True)
