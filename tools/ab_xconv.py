"""Timing of the register-resident-weight 1x1 launches (xconv.hip: 16-bit rows; xconv2.hip: split planes) — run once per
build / switch to A/B on one box:   HIAST_LIB=... python3 tools/ab_xconv.py ;  HIAST_XCONV2=0 python3 tools/ab_xconv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hiast_amd import kernels as K      # noqa: E402
from ab_igemm import timeit             # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, H, W, ci, co = 8, 64, 128, 256, 1024
    torch.manual_seed(0)
    print("library:", os.environ.get("HIAST_LIB", "in-tree"), "HIAST_XCONV2 =", os.environ.get("HIAST_XCONV2", "1"))
    wt = torch.randn(co, ci, 1, 1, device=dev) * (2.0 / ci) ** 0.5
    bn = torch.nn.BatchNorm2d(co).to(dev).eval()
    x32 = torch.randn(B, H, W, ci, device=dev)
    M = B * H * W
    for name, fmt, dt in (("fp16", K.FMT_FP16, torch.float16), ("bf16", K.FMT_BF16, torch.bfloat16)):
        xp = x32.to(dt)
        wp = K.pack_conv_weight(wt, fmt)
        res = torch.randn(B, H, W, co, device=dev).to(dt)
        bits = (torch.rand(M, co // 8, device=dev) * 256).to(torch.uint8)
        rows = []
        for tag, fn, mb in (("bn+res+relu", lambda: K.igemm_bn_act(xp, wp, 1, bn, res, True), 302),
                            ("gated dgrad", lambda: K.igemm_bn_act(xp, wp, 1, None, res, False, res_gate=bits), 310),
                            ("stats", lambda: K.igemm_bn_act(xp, wp, 1, None, None, False, want_stats=True), 168),
                            ("plain", lambda: K.igemm_bn_act(xp, wp, 1, None, None, False), 168)):
            t = timeit(fn)
            rows.append("%s %6.1f us = %.2f TB/s" % (tag, t * 1e3, mb / t / 1e3))
        print("xconv %s 256->1024 B=8: " % name + " | ".join(rows), flush=True)
    xp2 = K.split_planes(x32.view(-1, ci)).view(B, H, W, 2 * ci)
    wp2 = K.pack_conv_weight(wt, K.FMT_SPLIT_BF16)
    res2 = K.split_planes(torch.randn(M, co, device=dev)).view(B, H, W, 2 * co)
    t = timeit(lambda: K.igemm_bn_act(xp2, wp2, 2, bn, res2, True))
    print("split planes 256->1024 bn+res+relu: %6.1f us = %.2f TB/s (605 MB)" % (t * 1e3, 605 / t / 1e3), flush=True)
    t = timeit(lambda: K.igemm_bn_act(xp2, wp2, 2, bn, None, True))
    print("split planes 256->1024 bn+relu:     %6.1f us = %.2f TB/s (336 MB)" % (t * 1e3, 336 / t / 1e3), flush=True)


if __name__ == "__main__":
    main()
