"""Experiment: ResNet-101 (output stride 8) trunk timing on PyTorch-ROCm/MIOpen by memory format and dtype."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hiast_amd.sseg.models.modules.resnet import build_resnet101

def t(fn, n=3, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.backends.cudnn.benchmark = bool(int(os.environ.get("FIND", "0")))
for cl in (False, True):
    net = build_resnet101(output_stride=8).cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            for p in m.parameters(): p.requires_grad = False
    x = torch.randn(B, 3, 512, 1024, device="cuda")
    if cl:
        net = net.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    net.eval()
    def fwd32():
        with torch.no_grad(): net(x)
    def fwd16():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16): net(x)
    print("channels_last=%s eval fwd fp32  %.1f ms" % (cl, t(fwd32)), flush=True)
    print("channels_last=%s eval fwd bf16  %.1f ms" % (cl, t(fwd16)), flush=True)
    net.train()
    def fb16():
        with torch.autocast("cuda", dtype=torch.bfloat16): y = net(x)
        y.float().mean().backward()
    print("channels_last=%s train fwd+bwd bf16 %.1f ms" % (cl, t(fb16)), flush=True)
    def fbh():
        with torch.autocast("cuda", dtype=torch.float16): y = net(x)
        y.float().mean().backward()
    print("channels_last=%s train fwd+bwd fp16 %.1f ms" % (cl, t(fbh)), flush=True)
    del net, x
