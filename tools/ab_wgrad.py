"""A/B of the weight-gradient kernel (K9d): python tools/ab_wgrad.py  (library via HIAST_LIB).  Times
hiast_conv_wgrad_nhwc (wgrad_tn_kernel + wgrad_reduce_kernel) on the six trunk shapes of the bench workload."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hiast_amd import kernels as K  # noqa: E402
from ab_igemm import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W = 8, 64, 128
dt = torch.float16 if os.environ.get("AB_DT", "fp16") == "fp16" else torch.bfloat16
row = "%-28s" % (os.environ.get("HIAST_LIB", "in-tree")[-28:])
for name, ci, co, k, dl in (("l3.c1", 1024, 256, 1, 1), ("l3.c2", 256, 256, 3, 2), ("l3.c3", 256, 1024, 1, 1),
                            ("l4.c1", 2048, 512, 1, 1), ("l4.c2", 512, 512, 3, 4), ("l4.c3", 512, 2048, 1, 1)):
    x = torch.randn(B, H, W, ci, device=dev).to(dt)
    dy = torch.randn(B, H, W, co, device=dev).to(dt)
    t = timeit(lambda: K.conv_wgrad_nhwc(dy, x, k, 1, dl), n=40)
    gf = 2.0 * B * H * W * ci * co * k * k / 1e9
    row += " | %s %6.1f us %4.0f TF" % (name, t * 1e3, gf / t)
print(row, flush=True)
