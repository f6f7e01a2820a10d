"""compare the tile kernel's output under HIAST_LIB (e.g. the early-barrier build) with reference values saved by a run of the
in-tree library: python tools/dbg/eb_check.py save|check"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hiast_amd import kernels as K
mode = sys.argv[1]
dev = torch.device("cuda:0")
os.environ["HIAST_XCONV"] = "0"; os.environ["HIAST_XCONV2"] = "0"
cases = []
for (B, H, W) in ((2, 9, 17), (1, 64, 64)):
    for ci, co, taps in ((64, 64, 1), (128, 64, 1), (192, 64, 1), (64, 128, 1), (64, 256, 1), (256, 256, 1), (64, 64, 9), (32, 64, 9), (128, 256, 9)):
        for PL in (1, 2):
            if PL == 1 and ci % 64:
                continue
            cases.append((B, H, W, ci, co, taps, PL))
out = {}
for c in cases:
    B, H, W, ci, co, taps, PL = c
    torch.manual_seed(hash(c) % 1000)
    kk = 3 if taps == 9 else 1
    w = torch.randn(co, ci, kk, kk, device=dev) * (2.0 / (ci * taps)) ** 0.5
    bn = torch.nn.BatchNorm2d(co).to(dev).eval()
    x32 = torch.randn(B, H, W, ci, device=dev)
    xp = K.split_planes(x32.view(-1, ci)).view(B, H, W, 2 * ci) if PL == 2 else x32.bfloat16()
    wp = K.pack_conv_weight(w, PL)
    y = K.igemm_bn_act(xp, wp, PL, bn, None, True, 1, 1)
    y2 = K.igemm_bn_act(xp, wp, PL, bn, None, True, 1, 1)
    out[c] = y.cpu()
    if not torch.equal(y, y2):
        print("NOT REPEATABLE", c)
path = "/tmp/eb_ref.pt"
if mode == "save":
    torch.save(out, path); print("saved", len(out))
else:
    ref = torch.load(path)
    for c in cases:
        same = torch.equal(ref[c], out[c])
        if not same:
            d = (ref[c].float() - out[c].float()).abs()
            bad = (d > 0).view(-1, d.shape[-1])
            rows = bad.any(1).nonzero().flatten()[:6].tolist(); cols = bad.any(0).nonzero().flatten()[:8].tolist()
            print("DIFF", c, "nk=%d" % (taps_ := (c[5] * c[3] * c[6] // 64)), "max", float(d.max()), "n", int(bad.sum()), "rows", rows, "cols", cols)
    print("checked", len(cases))
