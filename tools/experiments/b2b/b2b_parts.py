"""time of hiast_bottleneck_tail (whatever build HIAST_LIB selects) on the layer3 shape, B = 8, three formats"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from hiast_amd import kernels as K
from ab_igemm import timeit
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, H, W, C, Co, dil = int(os.environ.get("BB_B", 8)), 64, 128, 256, 1024, 2
w2 = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
w3 = torch.randn(Co, C, 1, 1, device=dev) * (2.0 / C) ** 0.5
bn2 = torch.nn.BatchNorm2d(C).to(dev).eval(); bn3 = torch.nn.BatchNorm2d(Co).to(dev).eval()
x32 = torch.randn(B, H, W, C, device=dev).relu(); r32 = torch.randn(B, H, W, Co, device=dev)
out = []
for name, PL, fmt, dt in (("split", 2, K.FMT_SPLIT_BF16, None), ("fp16", 1, K.FMT_FP16, torch.float16)):
    if PL == 2:
        xin = K.split_planes(x32.view(-1, C)).view(B, H, W, 2 * C); res = K.split_planes(r32.view(-1, Co)).view(B, H, W, 2 * Co)
    else:
        xin, res = x32.to(dt), r32.to(dt)
    w2p, w3p = K.pack_conv_weight(w2, fmt), K.pack_conv_weight(w3, fmt)
    t = timeit(lambda: K.bottleneck_tail(xin, w2p, bn2, w3p, bn3, res, PL, dil), n=40)
    t3 = timeit(lambda: K.igemm_bn_act(xin, w2p, PL, bn2, None, True, 1, dil), n=40)
    out.append("%s %.1f us (tile-kernel 3x3 %.1f)" % (name, t * 1e3, t3 * 1e3))
print("%-28s %s" % (os.path.basename(os.environ.get("HIAST_LIB", "in-tree")), " | ".join(out)), flush=True)
