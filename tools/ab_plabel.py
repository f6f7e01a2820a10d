"""A/B of plabel_pass1 (library via HIAST_LIB): logits of std `scale` at 64x128 -> 512x1024, B = 8"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hiast_amd import kernels as K  # noqa: E402
from ab_igemm import timeit  # noqa: E402

torch.manual_seed(0)
row = "%-24s" % os.environ.get("HIAST_LIB", "in-tree")[-24:]
for scale in (1.0, 3.0, 6.0, 12.0):
    z = torch.randn(8, 19, 64, 128, device="cuda") * scale
    t = timeit(lambda: K.plabel_pass1(z, 512, 1024), n=30)
    mp, am, hist = K.plabel_pass1(z, 512, 1024)
    conf = float((mp > 0.94).float().mean())
    row += " | std %4.1f: %6.1f us (%.0f %% above 0.94)" % (scale, t * 1e3, conf * 100)
for coarse in ((8, 16), (16, 32)):        # spatially smooth maps, as a network produces them: neighbouring lanes hold neighbouring bins
    z = torch.nn.functional.interpolate(torch.randn(8, 19, *coarse, device="cuda") * 6.0, size=(64, 128), mode="bilinear", align_corners=True)
    t = timeit(lambda: K.plabel_pass1(z, 512, 1024), n=30)
    mp, am, hist = K.plabel_pass1(z, 512, 1024)
    row += " | smooth %dx%d: %6.1f us (%.0f %% above 0.94)" % (coarse[0], coarse[1], t * 1e3, float((mp > 0.94).float().mean()) * 100)
print(row, flush=True)
