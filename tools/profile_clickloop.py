#!/usr/bin/env python3
"""cProfile of the click loop (host side) -- where does per-click wall time go?"""
import cProfile, pstats, logging, sys, io
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import S14, build_model, seeded_
from isegprobe_amd.core.inference.evaluation import evaluate_sample
from isegprobe_amd.core.inference.predictors import get_predictor
S = 448
model = seeded_(build_model("bilinear", vit=S14, img=(S, S)), 1).cuda().eval()
rng = np.random.default_rng(0)
image = rng.integers(0, 255, (480, 640, 3), dtype=np.uint8)
yy, xx = np.mgrid[:480, :640]
gt = (((yy - 240) / 150) ** 2 + ((xx - 300) / 200) ** 2 <= 1).astype(np.int32)
predictor = get_predictor(model, "NoBRS", torch.device("cuda"), prob_thresh=0.5, zoom_in_params={"skip_clicks": -1, "target_size": (S, S)})
evaluate_sample(image, gt, predictor, max_iou_thr=1.01, pred_thr=0.5, max_clicks=20)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    evaluate_sample(image, gt, predictor, max_iou_thr=1.01, pred_thr=0.5, max_clicks=20)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
