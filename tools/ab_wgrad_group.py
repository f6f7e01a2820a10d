"""parts of the grouped weight-gradient kernel: python tools/ab_wgrad_group.py  (HIAST_LIB = a tools/build_variant.sh build of
wgrad.hip with -DWG_ABL_NODMA / -DWG_ABL_NOMFMA / -DWG_ABL_NOREAD); prints the time of the layer3 / layer4 grouped launches"""
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
dt = torch.float16
row = "%-34s" % (os.environ.get("HIAST_LIB", "in-tree")[-34:])
for name, cin, mid, dl in (("layer3", 1024, 256, 2), ("layer4 1x1 pair", 2048, 512, 4)):
    mk = lambda c: torch.randn(B, H, W, c, device=dev).to(dt)
    x1, d1, x2, d2, x3, d3 = mk(cin), mk(mid), mk(mid), mk(mid), mk(mid), mk(cin)
    jobs = [(d3, x3, 1, 1, 1), (d2, x2, 3, 1, dl), (d1, x1, 1, 1, 1)]
    if name != "layer3":
        jobs = [jobs[0], jobs[2]]
    t = timeit(lambda: K.conv_wgrad_group(jobs), n=30)
    gf = sum(2.0 * B * H * W * j[0].shape[3] * j[1].shape[3] * j[2] ** 2 for j in jobs) / 1e9
    row += " | %s %6.1f us %4.0f TF" % (name, t * 1e3, gf / t)
print(row, flush=True)
