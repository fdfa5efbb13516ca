#!/usr/bin/env python3
"""Stage timing of the LoftUp upsampler at 448^2 (C=384)."""
import os, sys, logging
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import seeded_
from isegprobe_amd.core.model.upsamplers import LoftUpUpsampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
up = seeded_(LoftUpUpsampler(None, n_dim=384), 3).cuda().eval()
src = torch.randn(B, 384, 32, 32, device="cuda")
gd = torch.randn(B, 3, 448, 448, device="cuda")
with torch.no_grad():
    for _ in range(2): y = up(src, gd)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): y = up(src, gd)
    e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 3
print(f"LoftUp B={B} 448^2: {ms:.2f} ms/batch = {ms/B:.2f} ms/img, {2.12*B/ms*1e3:.0f} TFLOP/s algorithmic; out {tuple(y.shape)} mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
