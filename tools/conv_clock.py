#!/usr/bin/env python3
"""Loop the head conv for a few seconds while sampling rocm-smi clocks/power from a child process."""
import subprocess
import sys
import time
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
B = 8
mode = sys.argv[1] if len(sys.argv) > 1 else "randn"
x = torch.randn(B, 448, 448, 384, device="cuda")
x = {"randn": x, "relu": x.relu(), "zero": torch.zeros_like(x)}[mode].to(torch.bfloat16)
w = (torch.randn(384, 9 * 384, device="cuda") / 60).to(torch.bfloat16)
bias = torch.randn(384, device="cuda")
for _ in range(3):
    ops.conv3x3(x, w, bias, "relu")
torch.cuda.synchronize()
t0 = time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 0
samples = []
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.conv3x3(x, w, bias, "relu")
    n += 50
    if len(samples) < 3 and time.time() - t0 > 1.0 + len(samples):
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        samples.append([l.strip() for l in r.stdout.splitlines() if "sclk" in l or "Power" in l or "mclk" in l])
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"mode={mode} {ms:.3f} ms {2.0*B*448*448*384*9*384/ms/1e9:.0f} TFLOP/s")
for smp in samples:
    print("  ", " | ".join(smp))
