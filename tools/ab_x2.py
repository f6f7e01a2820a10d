import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from hiast_amd import kernels as K
from ab_igemm import timeit
dev = torch.device("cuda:0"); torch.manual_seed(0)
B,H,W,ci,co = 8,64,128,256,1024
wt = torch.randn(co, ci, 1, 1, device=dev) * (2.0/ci)**0.5
bn = torch.nn.BatchNorm2d(co).to(dev).eval()
x32 = torch.randn(B,H,W,ci, device=dev)
xp2 = K.split_planes(x32.view(-1, ci)).view(B,H,W,2*ci); wp2 = K.pack_conv_weight(wt, 2)
res2 = K.split_planes(torch.randn(B*H*W, co, device=dev)).view(B,H,W,2*co)
t = timeit(lambda: K.igemm_bn_act(xp2, wp2, 2, bn, res2, True), n=60)
t2 = timeit(lambda: K.igemm_bn_act(xp2, wp2, 2, bn, None, True), n=60)
print("%-60s res %.1f us = %.2f TB/s | no-res %.1f us" % (os.environ.get("HIAST_LIB","in-tree")[-40:] + " X2=" + os.environ.get("HIAST_XCONV2","1"), t*1e3, 605/t/1e3, t2*1e3))
