import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import synth
from hiast_amd.sseg.models.modules.resnet import build_resnet101
torch.manual_seed(5)
m = build_resnet101(False, 8).cuda().eval()
x = torch.from_numpy(synth.normal_f32(3710, (2, 3, 64, 96))).cuda()
with torch.no_grad():
    a = m(x)
    os.environ["HIAST_NO_STEM_FUSED"] = "1"
    b = m(x)
    os.environ["HIAST_NO_FAST_EVAL"] = "1"
    c = m(x)          # module path (library, fp32)
    print("fused vs tail:", float((a - b).abs().max()) / float(b.abs().max()), " fused vs module fp32:", float((a - c).abs().max()) / float(c.abs().max()),
          " tail vs module:", float((b - c).abs().max()) / float(c.abs().max()))
os.environ.pop("HIAST_NO_STEM_FUSED"); os.environ.pop("HIAST_NO_FAST_EVAL")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    a16 = m(x).float()
    os.environ["HIAST_NO_STEM_FUSED"] = "1"
    b16 = m(x).float()
s = float(c.abs().max())
print("fp16: fused vs fp32 %.3e | tail vs fp32 %.3e | fused vs tail %.3e ; finite %s %s" % (float((a16 - c).abs().max()) / s, float((b16 - c).abs().max()) / s,
      float((a16 - b16).abs().max()) / s, bool(torch.isfinite(a16).all()), bool(torch.isfinite(b16).all())))
