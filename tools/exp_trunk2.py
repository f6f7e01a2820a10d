"""Experiment: conv-only cost of the trunk (BN/ReLU replaced by identity) in NCHW vs channels_last, bf16 autocast."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hiast_amd.sseg.models.modules.resnet as R

def t(fn, n=4, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

R.bn_act = lambda bn, x, res=None, relu=True: (x if res is None else x + res)
B = 8
for cl in (False, True):
    net = R.build_resnet101(output_stride=8).cuda()
    x = torch.randn(B, 3, 512, 1024, device="cuda")
    if cl:
        net = net.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    net.train()
    def f16():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16): net(x)
    def fb16():
        with torch.autocast("cuda", dtype=torch.bfloat16): y = net(x)
        y.float().mean().backward()
    def f32():
        with torch.no_grad(): net(x)
    print("conv-only channels_last=%s fwd bf16 %.1f ms | fwd+bwd bf16 %.1f ms | fwd fp32 %.1f ms" % (cl, t(f16), t(fb16), t(f32)), flush=True)
