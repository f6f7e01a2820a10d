import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from hiast_amd import kernels as K
from test_gpu_kernels import _igemm_ref, _mk_bn, _bf16r, dev
for case in [(2, 9, 17, 64, 64, 1, 1, 1, False), (2, 9, 17, 128, 64, 1, 1, 1, False), (2, 9, 17, 64, 128, 1, 1, 1, False), (2, 9, 17, 64, 256, 1, 1, 1, False), (1, 40, 40, 64, 64, 1, 1, 1, False)]:
    B, H, W, Cin, Cout, taps, stride, dil, has_res = case
    x = _bf16r(synth.normal_f32(320, (B, H, W, Cin)))
    w = synth.normal_f32(321, (Cout, Cin, 1, 1), (2.0 / Cin) ** 0.5)
    bn, bnref = _mk_bn(322, Cout)
    xp = dev(x).bfloat16(); wp = K.pack_conv_weight(dev(w), 1)
    want = _igemm_ref(x, _bf16r(w), bnref, None, True, 1, 1, 1)
    for rep in range(3):
        y = K.igemm_bn_act(xp, wp, 1, bn, None, True, 1, 1).float().cpu().numpy()
        bad = ~(np.abs(y - want) <= 2.0 ** -8 * np.abs(want) + 3e-5 * np.abs(want).max())
        b2 = bad.reshape(-1, Cout)
        print(case[:6], "rep", rep, "bad", int(bad.sum()), "rows", np.nonzero(b2.any(1))[0][:10], "cols", np.nonzero(b2.any(0))[0][:16])
        if bad.sum():
            i = np.argwhere(b2)[0]; print("   first bad", i, "got", y.reshape(-1, Cout)[i[0], i[1]], "want", want.reshape(-1, Cout)[i[0], i[1]])
