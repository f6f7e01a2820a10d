"""A/B of the one-launch bottleneck tail (K9m, hiast_bottleneck_tail) against the two launches it replaces, stand-alone on the
layer3 shape of the bench (B = 8 and the B = 4 sub-batch of eval_forward_split), the three operand formats.
Event-timed back-to-back launches (median); a sequence conv1 -> tail as the eval forward issues it is timed too."""
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
H, W, C, Co, dil = 64, 128, 256, 1024, 2
for B in (8, 4):
    w1 = torch.randn(C, Co, 1, 1, device=dev) * (2.0 / Co) ** 0.5
    w2 = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    w3 = torch.randn(Co, C, 1, 1, device=dev) * (2.0 / C) ** 0.5
    bn1 = torch.nn.BatchNorm2d(C).to(dev).eval()
    bn2 = torch.nn.BatchNorm2d(C).to(dev).eval()
    bn3 = torch.nn.BatchNorm2d(Co).to(dev).eval()
    x32 = torch.randn(B, H, W, Co, device=dev).relu()
    for name, PL, fmt, dt in (("split", 2, K.FMT_SPLIT_BF16, None), ("fp16", 1, K.FMT_FP16, torch.float16),
                              ("bf16", 1, K.FMT_BF16, torch.bfloat16)):
        if PL == 2:
            xin = K.split_planes(x32.view(-1, Co)).view(B, H, W, 2 * Co)
        else:
            xin = x32.to(dt)
        w1p, w2p, w3p = (K.pack_conv_weight(w, fmt) for w in (w1, w2, w3))
        a1 = K.igemm_bn_act(xin, w1p, PL, bn1, None, True)

        def two():
            a2 = K.igemm_bn_act(a1, w2p, PL, bn2, None, True, 1, dil)
            return K.igemm_bn_act(a2, w3p, PL, bn3, xin, True)

        def one():
            return K.bottleneck_tail(a1, w2p, bn2, w3p, bn3, xin, PL, dil)

        def blk(tail):
            o = K.igemm_bn_act(xin, w1p, PL, bn1, None, True)
            if tail:
                return K.bottleneck_tail(o, w2p, bn2, w3p, bn3, xin, PL, dil)
            o = K.igemm_bn_act(o, w2p, PL, bn2, None, True, 1, dil)
            return K.igemm_bn_act(o, w3p, PL, bn3, xin, True)

        t3 = timeit(lambda: K.igemm_bn_act(a1, w2p, PL, bn2, None, True, 1, dil), n=40)
        t2, t1 = timeit(two, n=40), timeit(one, n=40)
        tb2, tb1 = timeit(lambda: blk(False), n=40), timeit(lambda: blk(True), n=40)
        eq = torch.equal(one(), two())
        print("B=%d %-5s 3x3 alone %6.1f us | conv2+conv3: two launches %6.1f us, one launch %6.1f us (%+.1f) | whole block: %6.1f -> %6.1f us (%+.1f) | bit-equal: %s"
              % (B, name, t3 * 1e3, t2 * 1e3, t1 * 1e3, (t1 - t2) * 1e3, tb2 * 1e3, tb1 * 1e3, (tb1 - tb2) * 1e3, eq), flush=True)
