"""TIMING experiment: the K = 1024 / 2048 1x1 launches of the tile kernel when the A operand is stored slab-major inside each
256-row tile (a k-step's operand tile = one contiguous 32 KiB run instead of 256 pieces of 128 B at the row pitch).  Run once with
the in-tree library and once with HIAST_LIB=hiast_amd/csrc/_ab/libhiast_slabmajor.so (built with -DIG_SLABMAJOR_A: the results are
meaningless, the memory traffic is the same bytes in another order)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from hiast_amd import kernels as K
from ab_igemm import timeit
dev = torch.device("cuda:0"); torch.manual_seed(0)
os.environ["HIAST_IGEMM_HALF"] = "0"
row = os.path.basename(os.environ.get("HIAST_LIB", "in-tree"))[:28].ljust(28)
for B in (8,):
    for name, H, W, ci, co in (("1024->256", 64, 128, 1024, 256), ("2048->512", 64, 128, 2048, 512), ("512->128 l2", 128, 256, 512, 128)):
        w = torch.randn(co, ci, 1, 1, device=dev) * (2.0 / ci) ** 0.5
        bn = torch.nn.BatchNorm2d(co).to(dev).eval()
        x32 = torch.randn(B, H, W, ci, device=dev)
        xs = K.split_planes(x32.view(-1, ci)).view(B, H, W, 2 * ci); wp2 = K.pack_conv_weight(w, 2)
        xh = x32.half(); wph = K.pack_conv_weight(w, K.FMT_FP16)
        t2 = timeit(lambda: K.igemm_bn_act(xs, wp2, 2, bn, None, True), n=40) * 1e3
        t1 = timeit(lambda: K.igemm_bn_act(xh, wph, 1, bn, None, True), n=40) * 1e3
        ts = timeit(lambda: K.igemm_bn_act(xh, wph, 1, None, None, False, want_stats=True), n=40) * 1e3
        row += " | %s: split %6.1f fp16 %6.1f stats %6.1f" % (name, t2, t1, ts)
print(row, flush=True)
