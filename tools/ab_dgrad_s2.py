import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from hiast_amd import kernels as K
from ab_igemm import timeit
dev = torch.device("cuda:0")
B, H, W, C = 8, 128, 256, 128
dy = torch.randn(B, 64, 128, C, device=dev).half()
w = torch.randn(C, C, 3, 3, device=dev) * 0.03
wpt = K.pack_conv_weight(w, K.fmt_of(dy), transpose=True)
x = torch.randn(B, H, W, C, device=dev).half().permute(0, 3, 1, 2)
t_own = timeit(lambda: K.igemm_dgrad_s2(dy, wpt, H, W), n=30)
wl = w.half()
t_lib = timeit(lambda: torch.ops.aten.convolution_backward(dy.permute(0, 3, 1, 2), x, wl, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)), n=30)
print("layer2.0.conv2 data gradient (3x3 s2, 128 ch, B=8 128x256): own %.1f us | library %.1f us" % (t_own * 1e3, t_lib * 1e3))
