"""which elements of hiast_bottleneck_tail's output differ between repeated launches / from the two-launch form"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hiast_amd import kernels as K
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, H, W, C, Co, dil = int(os.environ.get("BB_B", 1)), 64, 128, 256, 1024, 2
w2 = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
w3 = torch.randn(Co, C, 1, 1, device=dev) * (2.0 / C) ** 0.5
bn2 = torch.nn.BatchNorm2d(C).to(dev).eval(); bn3 = torch.nn.BatchNorm2d(Co).to(dev).eval()
x32 = torch.randn(B, H, W, C, device=dev).relu(); r32 = torch.randn(B, H, W, Co, device=dev)
for name, PL, fmt, dt in (("split", 2, K.FMT_SPLIT_BF16, None), ("fp16", 1, K.FMT_FP16, torch.float16), ("bf16", 1, K.FMT_BF16, torch.bfloat16)):
    if PL == 2:
        xin = K.split_planes(x32.view(-1, C)).view(B, H, W, 2 * C); res = K.split_planes(r32.view(-1, Co)).view(B, H, W, 2 * Co)
    else:
        xin, res = x32.to(dt), r32.to(dt)
    w2p, w3p = K.pack_conv_weight(w2, fmt), K.pack_conv_weight(w3, fmt)
    a2 = K.igemm_bn_act(xin, w2p, PL, bn2, None, True, 1, dil)
    ref = K.igemm_bn_act(a2, w3p, PL, bn3, res, True)
    nbad = 0
    for rep in range(int(os.environ.get("BB_REPS", 30))):
        y = K.bottleneck_tail(xin, w2p, bn2, w3p, bn3, res, PL, dil)
        bad = (y.view(B * H * W, -1).view(torch.int16) != ref.view(B * H * W, -1).view(torch.int16))
        n = int(bad.sum())
        if n:
            nbad += 1
            rows = bad.any(1).nonzero().flatten().cpu().numpy(); cols = bad.any(0).nonzero().flatten().cpu().numpy()
            if nbad <= 4:
                print("  %s rep %d: %d elements differ; rows %s ... (n=%d; mod 256: %s) cols %s ... (n=%d; //32: %s)" % (
                    name, rep, n, rows[:8], len(rows), sorted(set((rows % 256) // 32))[:8], cols[:8], len(cols),
                    sorted(set(cols // (32 * PL)))[:12]))
    print("%s: %d of the launches differ from the two-launch form" % (name, nbad), flush=True)
