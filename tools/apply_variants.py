#!/usr/bin/env python3
"""A/B jbu_apply builds: one subprocess per variant .so (env ISEGPROBE_HIP_LIB); outputs compared bit for bit with the
first variant's."""
import glob, os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
torch.manual_seed(0)
out = []
for (B, h, C) in ((2, 37, 192), (32, 256, 384), (32, 128, 384)):
    x = torch.randn(B, h, h, C, device="cuda").to(torch.bfloat16)
    kc = torch.rand(B, 2*h, 2*h, 8, 16, device="cuda") / 8
    xs = torch.arange(2*h, device="cuda")
    bx = ((xs - 4) >> 1) - 1
    slot = torch.arange(16, device="cuda")
    inwin = ((slot[None, :] - bx[:, None]) & 15) < 8          # the format's precondition: zero outside the window slots
    kc = (kc * inwin[None, None, :, None, :]).to(torch.float16)
    y = ops.jbu_apply(x, kc); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): y = ops.jbu_apply(x, kc)
    e.record(); torch.cuda.synchronize()
    ref = "/tmp/apply_ref_%d_%d_%d.pt" % (B, h, C)
    if os.path.exists(ref):
        r = torch.load(ref).float(); same = "maxdiff %.2e (ref rms %.2e)" % ((r - y.cpu().float()).abs().max().item(), r.pow(2).mean().sqrt().item())
    else:
        torch.save(y.cpu(), ref); same = "ref"
    out.append(f"B={B} {h}->{2*h} C={C}: {s.elapsed_time(e)/5:.3f} ms same={same}")
print(" | ".join(out))
'''
for f in glob.glob("/tmp/apply_ref_*.pt"):
    os.remove(f)  # references of an earlier run
for lib in sorted(glob.glob("build_variants/lib_*.so")):
    env = dict(os.environ, ISEGPROBE_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(os.path.basename(lib), (r.stdout.strip().splitlines() or ["?"])[-1], r.stderr.strip().splitlines()[-1:] if r.returncode else "", flush=True)
