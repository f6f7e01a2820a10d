import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hiast_amd.utils.registry import register  # noqa
from hiast_amd.utils.registry.registries import SEG_MODEL
import hiast_amd.sseg.models.modules.resnet as R
from hiast_amd import kernels as K
def t(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
m = SEG_MODEL["DeepLab_V2"](19, 256).cuda().eval()
x = torch.randn(8, 3, 512, 1024, device="cuda")
def f32():
    with torch.no_grad(): m.backbone(x)
def b16():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16): m.backbone(x)
print("fast fp32 trunk %.1f ms | fast bf16 trunk %.1f ms" % (t(f32), t(b16)))
orig = R.ResNet._fast_eval_ok
R.ResNet._fast_eval_ok = lambda self, x: False
print("module fp32 trunk %.1f ms | module bf16 trunk %.1f ms" % (t(f32), t(b16)))
R.ResNet._fast_eval_ok = orig
for (M, Kd, N, res) in [(65536, 1024, 256, False), (65536, 256, 1024, True), (65536, 2048, 512, False), (65536, 512, 2048, True)]:
    for dt in (torch.float32, torch.bfloat16):
        xx = torch.randn(M, Kd, device="cuda").to(dt); w = torch.randn(N, Kd, device="cuda") * 0.05
        r = torch.randn(M, N, device="cuda").to(dt) if res else None
        bn = torch.nn.BatchNorm2d(N).cuda().eval()
        ms = t(lambda: K.conv1x1_bn_act_nhwc(xx, w, bn, r, True), n=20, warm=5)
        print("conv1x1 %s M%d K%d N%d res=%d %.3f ms" % (str(dt)[6:], M, Kd, N, res, ms))
for dt in (torch.float32, torch.bfloat16):
    xx = torch.randn(8, 64, 128, 256, device="cuda").to(dt); w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
    bn = torch.nn.BatchNorm2d(256).cuda().eval()
    print("conv3x3 %s 256ch d2 %.3f ms" % (str(dt)[6:], t(lambda: K.conv3x3_bn_act_nhwc(xx, w, bn, 1, 2, True), n=20, warm=5)))
