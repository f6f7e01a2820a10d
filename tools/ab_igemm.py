"""Timing of the igemm launches of the trunk (one line per shape) — run once per build to A/B two libraries on one box:
    HIAST_LIB=/path/to/libhiast_hip_base.so python3 tools/ab_igemm.py ; python3 tools/ab_igemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hiast_amd import kernels as K      # noqa: E402


def timeit(fn, n=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    B, H, W = 8, 64, 128
    torch.manual_seed(0)
    print("library:", os.environ.get("HIAST_LIB", "in-tree"))
    cases = [("l3.conv2 3x3 d2", 256, 256, 9, 2, False), ("l3.conv1 1x1", 1024, 256, 1, 1, False),
             ("l3.conv3 1x1+res", 256, 1024, 1, 1, True), ("l4.conv2 3x3 d4", 512, 512, 9, 4, False),
             ("l4.conv1 1x1", 2048, 512, 1, 1, False), ("l4.conv3 1x1+res", 512, 2048, 1, 1, True),
             ("l2.conv2 3x3", 128, 128, 9, 1, False), ("l2.conv3 1x1+res", 128, 512, 1, 1, True)]
    for name, ci, co, taps, dl, has_res in cases:
        kk = 3 if taps == 9 else 1
        wt = torch.randn(co, ci, kk, kk, device=dev) * (2.0 / (ci * taps)) ** 0.5
        bn = torch.nn.BatchNorm2d(co).to(dev).eval()
        x32 = torch.randn(B, H, W, ci, device=dev)
        gf = 2.0 * B * H * W * ci * co * taps / 1e9
        row = "%-18s %6.1f GF |" % (name, gf)
        for PL in (2, 1):
            xp = K.split_planes(x32.view(-1, ci)).view(B, H, W, 2 * ci) if PL == 2 else x32.bfloat16()
            wp = K.pack_conv_weight(wt, PL)
            res = torch.randn(B, H, W, PL * co, device=dev).bfloat16() if has_res else None
            t = timeit(lambda: K.igemm_bn_act(xp, wp, PL, bn, res, True, 1, dl))
            row += " PL%d bn+relu %6.1f us %5.0f TF/s |" % (PL, t * 1e3, gf / t * (3 if PL == 2 else 1))
            if PL == 1:
                t = timeit(lambda: K.igemm_bn_act(xp, wp, 1, None, None, False, 1, dl))
                row += " plain %6.1f |" % (t * 1e3)
                t = timeit(lambda: K.igemm_bn_act(xp, wp, 1, None, None, False, 1, dl, want_stats=True))
                row += " stats %6.1f |" % (t * 1e3)
        print(row, flush=True)


if __name__ == "__main__":
    main()
