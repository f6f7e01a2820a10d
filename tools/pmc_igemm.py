"""Launch ONE trunk-conv shape of the LDS-DMA implicit-GEMM kernel a few times (for rocprofv3 --pmc passes).
    python tools/pmc_igemm.py <name>     name in SHAPES"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hiast_amd import kernels as K  # noqa: E402

# name: (B, H, W, Cin, Cout, taps, dil, PL, has_res)
SHAPES = {
    "l3conv2_pl2": (8, 64, 128, 256, 256, 9, 2, 2, False),
    "l4conv2_pl2": (8, 64, 128, 512, 512, 9, 4, 2, False),
    "l3conv3_pl2": (8, 64, 128, 256, 1024, 1, 1, 2, True),
    "l3conv1_pl2": (8, 64, 128, 1024, 256, 1, 1, 2, False),
    "l3conv2_pl1": (8, 64, 128, 256, 256, 9, 2, 1, False),
    "l3conv3_pl1": (8, 64, 128, 256, 1024, 1, 1, 1, True),       # K9e xconv: teacher conv3 + BN + residual + ReLU
    "wgrad_l3conv3": (8, 64, 128, 256, 1024, 1, 1, 1, False),    # K9d: dW of the 256 -> 1024 1x1
    "wgrad_l3conv2": (8, 64, 128, 256, 256, 9, 2, 1, False),     # K9d: dW of the 3x3, dilation 2
    "wgrad_group_l3": (8, 64, 128, 256, 1024, 0, 2, 1, False),   # K9d (round 4): the three dW of a layer3 bottleneck, one launch
}


def main():
    name = sys.argv[1]
    B, H, W, Cin, Cout, taps, dil, PL, has_res = SHAPES[name]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    if name == "wgrad_group_l3":
        mk = lambda c: torch.randn(B, H, W, c, device=dev).half()
        x1, d1, x2, d2, x3, d3 = mk(Cout), mk(Cin), mk(Cin), mk(Cin), mk(Cin), mk(Cout)
        jobs = [(d3, x3, 1, 1, 1), (d2, x2, 3, 1, dil), (d1, x1, 1, 1, 1)]
        for _ in range(6):
            K.conv_wgrad_group(jobs)
        torch.cuda.synchronize()
        alg = sum((j[0].numel() + j[1].numel()) * 2 + j[0].shape[3] * j[1].shape[3] * j[2] ** 2 * 4 for j in jobs)
        flop = sum(2 * B * H * W * j[0].shape[3] * j[1].shape[3] * j[2] ** 2 for j in jobs)
        print("shape %s algorithmic_bytes %d flop %d (+ 255 fp32 partial tiles of 256 KiB written and re-read)" % (name, alg, flop))
        return
    if name.startswith("wgrad"):
        x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
        dy = torch.randn(B, H, W, Cout, device=dev).bfloat16()
        for _ in range(6):
            K.conv_wgrad_nhwc(dy, x, 3 if taps == 9 else 1, 1, dil)
        torch.cuda.synchronize()
        nsplit_bytes = K._lib.load().hiast_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, taps)
        alg = (x.numel() + dy.numel()) * 2 + Cout * Cin * taps * 4
        print("shape %s algorithmic_bytes %d flop %d (+ split partials written and re-read: %d bytes each way)"
              % (name, alg, 2 * B * H * W * taps * Cin * Cout, nsplit_bytes))
        return
    kk = 3 if taps == 9 else 1
    w = torch.randn(Cout, Cin, kk, kk, device=dev) * (2.0 / (Cin * taps)) ** 0.5
    bn = torch.nn.BatchNorm2d(Cout).to(dev).eval()
    x32 = torch.randn(B, H, W, Cin, device=dev)
    xp = K.split_planes(x32.view(-1, Cin)).view(B, H, W, 2 * Cin) if PL == 2 else x32.bfloat16()
    wp = K.pack_conv_weight(w, PL)
    res = torch.randn(B, H, W, PL * Cout, device=dev).bfloat16() if has_res else None
    for _ in range(6):
        K.igemm_bn_act(xp, wp, PL, bn, res, True, 1, dil)
    torch.cuda.synchronize()
    alg = (B * H * W * PL * Cin + Cout * taps * PL * Cin) * 2 + B * H * W * Cout * 2 * PL * (2 if has_res else 1)
    print("shape %s algorithmic_bytes %d flop %d" % (name, alg, 2 * B * H * W * taps * Cin * Cout))


if __name__ == "__main__":
    main()
