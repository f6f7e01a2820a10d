"""timing of the small-channel weight gradient (K9h) against the library on the layer1 / layer2 shapes of the bench workload
(B = 8, 512x1024 images): python tools/ab_wgrad_small.py"""
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
dt = torch.float16
tot_own = tot_lib = 0.0
for name, n, ci, co, H, W, k, stride in (("l1.0.conv1", 1, 64, 64, 128, 256, 1, 1), ("l1.x.conv1", 2, 256, 64, 128, 256, 1, 1),
                                         ("l1.x.conv2", 3, 64, 64, 128, 256, 3, 1), ("l1.x.conv3+ds", 4, 64, 256, 128, 256, 1, 1),
                                         ("l2.0.conv1", 1, 256, 128, 128, 256, 1, 2), ("l2.x.conv1", 3, 512, 128, 64, 128, 1, 1),
                                         ("l2.x.conv2", 4, 128, 128, 64, 128, 3, 1), ("l2.x.conv3", 4, 128, 512, 64, 128, 1, 1),
                                         ("l2.0.ds", 1, 256, 512, 128, 256, 1, 2)):
    B = 8
    x = torch.randn(B, H, W, ci, device=dev).to(dt)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randn(B, Ho, Wo, co, device=dev).to(dt)
    t = timeit(lambda: K.conv_wgrad_small_nhwc(dy, x, k, stride, 1), n=20)
    xl, dyl = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
    wl = torch.empty(co, ci, k, k, device=dev, dtype=dt)
    pad = 1 if k == 3 else 0
    t2 = timeit(lambda: torch.ops.aten.convolution_backward(dyl, xl, wl, None, (stride, stride), (pad, pad), (1, 1), False, (0, 0), 1,
                                                            (False, True, False))[1].float(), n=20)
    mb = (x.numel() / stride ** 2 + dy.numel()) * 2 / 1e6
    print("%-14s x%d  own %6.1f us (%4.2f TB/s of operands) | library + cast %6.1f us" % (name, n, t * 1e3, mb / t / 1e6, t2 * 1e3), flush=True)
    tot_own += n * t
    tot_lib += n * t2
print("per training step: own %.3f ms | library %.3f ms" % (tot_own, tot_lib))
