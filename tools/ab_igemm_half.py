"""A/B of the 128 x 128 / two-blocks-per-CU form of the tile kernel (HIAST_IGEMM_HALF=1) against the 256-row form on every 1x1
shape of the step (B = 8 and the B = 4 sub-batch of the inference forwards), the launch variants the step uses.
Event-timed back-to-back launches, median; the env switch is read per launch."""
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
# (name, H, W, Cin, Cout)
SHAPES = [("l3.conv1 1024->256", 64, 128, 1024, 256), ("l4.conv1 2048->512", 64, 128, 2048, 512),
          ("l4.0.conv1 1024->512", 64, 128, 1024, 512), ("l3.0.conv1 512->256", 64, 128, 512, 256),
          ("l2.conv1 512->128", 64, 128, 512, 128), ("l2.conv3 128->512", 64, 128, 128, 512),
          ("l4.conv3 512->2048", 64, 128, 512, 2048), ("l3.down 512->1024", 64, 128, 512, 1024),
          ("l1.conv1 256->64 (N%128!=0)", 128, 256, 256, 64), ("l1.conv3 64->256", 128, 256, 64, 256),
          ("l2.0.conv1 256->128", 128, 256, 256, 128)]


def both(fn):
    out = []
    for v in ("0", "1"):
        os.environ["HIAST_IGEMM_HALF"] = v
        out.append(timeit(fn, n=30) * 1e3)
    os.environ.pop("HIAST_IGEMM_HALF")
    return out


for B in (8, 4):
    for name, H, W, ci, co in SHAPES:
        w = torch.randn(co, ci, 1, 1, device=dev) * (2.0 / ci) ** 0.5
        bn = torch.nn.BatchNorm2d(co).to(dev).eval()
        x32 = torch.randn(B, H, W, ci, device=dev)
        row = "B=%d %-28s" % (B, name)
        xs = K.split_planes(x32.view(-1, ci)).view(B, H, W, 2 * ci)
        a, b = both(lambda: K.igemm_bn_act(xs, K.pack_conv_weight(w, 2) if False else wp2, 2, bn, None, True)) if False else (0, 0)
        wp2 = K.pack_conv_weight(w, 2)
        a, b = both(lambda: K.igemm_bn_act(xs, wp2, 2, bn, None, True))
        row += " | split bn+relu %6.1f -> %6.1f" % (a, b)
        xh = x32.half()
        wph = K.pack_conv_weight(w, K.FMT_FP16)
        a, b = both(lambda: K.igemm_bn_act(xh, wph, 1, bn, None, True))
        row += " | fp16 bn+relu %6.1f -> %6.1f" % (a, b)
        a, b = both(lambda: K.igemm_bn_act(xh, wph, 1, None, None, False, want_stats=True))
        row += " | fp16 stats %6.1f -> %6.1f" % (a, b)
        a, b = both(lambda: K.igemm_bn_act(xh, wph, 1, None, None, False))
        row += " | fp16 plain %6.1f -> %6.1f" % (a, b)
        # data gradient with the BatchNorm backward sums (dy has ci channels, da co)
        bx = torch.randn(B, H, W, co, device=dev).half()
        sm, si = torch.zeros(co, device=dev), torch.ones(co, device=dev)
        a, b = both(lambda: K.igemm_dgrad_bn_stats(xh, wph, 1, bx, None, None, sm, si))
        row += " | fp16 dgrad+sums %6.1f -> %6.1f" % (a, b)
        print(row, flush=True)
