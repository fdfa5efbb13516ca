import sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
torch.manual_seed(0)
for C in (384, 256):
    for (B, h) in ((8, 128), (32, 256)):
        x = torch.randn(B, h, h, C, device="cuda").to(torch.bfloat16)
        kc = (torch.rand(B, 2*h, 2*h, 8, 16, device="cuda") / 8).to(torch.float16)
        y = ops.jbu_apply(x, kc); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): y = ops.jbu_apply(x, kc)
        e.record(); torch.cuda.synchronize()
        print(f"C={C} B={B} {h}->{2*h}: {s.elapsed_time(e)/3:.3f} ms finite={bool(torch.isfinite(y.float()).all())}")
