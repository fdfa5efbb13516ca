#!/usr/bin/env python3
"""The head's two convolutions alone at the bench shape (batch 32, 448^2, 384 -> 384, IEEE half): conv + bias + ReLU with its
4.9 GB output map (the first conv's kind of epilogue) against conv + ReLU + classifier dot (the second's: 26 MB out).
HIP events over 5 launches; A/B builds (e.g. -DISP_ABLATE_NO_EPILOGUE) through ISEGPROBE_HIP_LIB."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isegprobe_amd import hip_ops as ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H16 = torch.float16
torch.manual_seed(0)
x = torch.relu(torch.randn(B, 448, 448, 384, device="cuda")).to(H16)
w = (torch.randn(384, 9 * 384, device="cuda") / math.sqrt(9 * 384)).to(H16)
bias, wc = torch.randn(384, device="cuda"), torch.randn(384, device="cuda") / 20


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


fl = 2.0 * B * 448 * 448 * 9 * 384 * 384
for name, fn in (("conv + bias + ReLU -> map ", lambda: ops.conv3x3(x, w, bias, "relu")),
                 ("conv + ReLU + classifier  ", lambda: ops.conv3x3_relu_classifier(x, w, bias, wc, 0.1))):
    ms = timeit(fn)
    print(f"{name}: {ms:7.3f} ms  {fl / ms / 1e9:6.0f} TFLOP/s")
