import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from hiast_amd.sseg.models.modules.resnet import Bottleneck
torch.manual_seed(0)
blk = Bottleneck(256, 64, 1, 2).cuda().train()
cl = lambda a: torch.from_numpy(a).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
x0 = cl(synth.normal_f32(950, (2, 256, 24, 40))); gy = cl(synth.normal_f32(951, (2, 256, 24, 40)))
outs = []
for off in ("1", "0"):
    from hiast_amd import switches as SW; SW.SWITCHES["HIAST_NO_IDT_HANDOFF"] = off == "1"
    blk.zero_grad()
    src = x0.clone().requires_grad_(True)
    xin = src * 1.0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(xin)
    y.backward(gy)
    outs.append((y.detach().float(), src.grad.float()))
(y0, g0), (y1, g1) = outs
print("y equal", torch.equal(y0, y1))
d = (g0 - g1).abs()
print("grad max", float(g0.abs().max()), "diff max", float(d.max()), "mean", float(d.mean()), "frac>1e-2", float((d > 1e-2 * g0.abs().max()).float().mean()))
idx = d.flatten().argmax(); print(float(g0.flatten()[idx]), float(g1.flatten()[idx]))
# expected masked residual
mask = (y0 > 0).float()
print("g0 - g1 vs masked gy:", float(((g0 - g1) - 0).abs().max()), float((gy.float() * mask).abs().max()))
