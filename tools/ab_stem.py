"""fused inference stem (K9j, hiast_stem_eval) against library convolution + hiast_stem_tail: agreement on a calibrated random-init
ResNet-101 and kernel time at B = 8, 512x1024: python tools/ab_stem.py  (library via HIAST_LIB)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from hiast_amd import kernels as K
from hiast_amd.tools import synth_data
from hiast_amd.sseg.models.modules.resnet import build_resnet101
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab_igemm import timeit
torch.manual_seed(5)
x = torch.from_numpy(synth.normal_f32(3710, (2, 3, 64, 96))).cuda()
m = synth_data.calibrate_bn(build_resnet101(False, 8).cuda(), x).eval()
with torch.no_grad():
    o1 = K.merge_planes(K.stem_eval(x, m.conv1.weight.detach(), m.bn1, 2).view(-1, 128))
    c = m.conv1(x.contiguous(memory_format=torch.channels_last)).contiguous(memory_format=torch.channels_last)
    o2 = K.merge_planes(K.stem_tail(c, m.bn1, 2).view(-1, 128))
    ref = torch.nn.functional.max_pool2d(torch.relu(m.bn1(m.conv1(x))), 3, 2, 1).permute(0, 2, 3, 1).reshape(-1, 64)
    print("stem only: fused vs tail %.3e, fused vs torch %.3e, tail vs torch %.3e (max %.3f)" % (float((o1 - o2).abs().max()), float((o1 - ref).abs().max()),
          float((o2 - ref).abs().max()), float(ref.abs().max())))
    print("bn1 var min", float(m.bn1.running_var.min()), "w dtype", m.conv1.weight.dtype)
    xb = torch.randn(8, 3, 512, 1024, device="cuda")
    for fmt in (2, 3):
        t = timeit(lambda: K.stem_eval(xb, m.conv1.weight.detach(), m.bn1, fmt), n=20)
        print("stem_eval fmt %d: %.1f us" % (fmt, t * 1e3))
