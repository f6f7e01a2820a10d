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

# round 4: the three weight gradients of one bottleneck one by one (6 launches) against the grouped launch (2 launches)
for name, cin, mid, dl in (("layer3 block", 1024, 256, 2), ("layer4 block", 2048, 512, 4)):
    x1 = torch.randn(B, H, W, cin, device=dev).to(dt)
    d1 = torch.randn(B, H, W, mid, device=dev).to(dt)
    x2 = torch.randn(B, H, W, mid, device=dev).to(dt)
    d2 = torch.randn(B, H, W, mid, device=dev).to(dt)
    x3 = torch.randn(B, H, W, mid, device=dev).to(dt)
    d3 = torch.randn(B, H, W, cin, device=dev).to(dt)
    jobs = [(d3, x3, 1, 1, 1), (d2, x2, 3, 1, dl), (d1, x1, 1, 1, 1)]
    t_one = timeit(lambda: [K.conv_wgrad_nhwc(*j) for j in jobs], n=30)
    t_grp = timeit(lambda: K.conv_wgrad_group(jobs), n=30)
    t_13 = timeit(lambda: K.conv_wgrad_group([jobs[0], jobs[2]]), n=30)
    gf = 2.0 * B * H * W * (2 * cin * mid + 9 * mid * mid) / 1e9
    print("%s: one by one %.1f us (%.0f TF) | grouped %.1f us (%.0f TF) | grouped 1x1 pair only %.1f us" %
          (name, t_one * 1e3, gf / t_one, t_grp * 1e3, gf / t_grp, t_13 * 1e3), flush=True)
