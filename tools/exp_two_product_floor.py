"""CPU experiment for VERDICT r4 item 7 (2-product split for the 3x3 launches of the pseudo-label forward):
the floor of that scheme is the rounding of the 3x3 WEIGHTS to fp16 (x_hi*w_hi + x_lo*w_hi keeps x at ~22 bits).
Oracle forward in float64 with exact weights vs. the same with conv2 (3x3) weights rounded to fp16."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "tests", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import synth
from oracle import deeplab_ref
from hiast_amd.utils.registry import register  # noqa
from hiast_amd.utils.registry.registries import SEG_MODEL
from make_golden import seeded_state_dict
torch.set_num_threads(8)
H, W = int(sys.argv[1]), int(sys.argv[2])
m = SEG_MODEL["DeepLab_V2"](19, 256)
sd = seeded_state_dict(m, 9100)
x = torch.from_numpy(synth.normal_f32(9200 + H, (1, 3, H, W)))
sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
with torch.no_grad():
    ref = deeplab_ref.deeplab_v2(x.double(), sd64)[0]
    for name, sel in (("3x3 weights -> fp16", lambda k, v: v.dim() == 4 and v.shape[-1] == 3 and "backbone" in k),
                      ("3x3 weights -> bf16", None),
                      ("all trunk conv weights -> fp16", lambda k, v: v.dim() == 4 and "backbone" in k)):
        sdq = dict(sd64)
        for k, v in sd64.items():
            if sel is None:
                if v.dim() == 4 and v.shape[-1] == 3 and "backbone" in k:
                    sdq[k] = v.float().bfloat16().double()
            elif sel(k, v):
                sdq[k] = v.float().half().double()
        got = deeplab_ref.deeplab_v2(x.double(), sdq)[0]
        err = (got - ref).abs()
        scale = float(ref.abs().max()); rms = float(ref.pow(2).mean().sqrt())
        print("%dx%d %-34s max|d|/max|ref| = %.2e   max rel(|ref|+rms) = %.2e   argmax differs %.2e" % (
            H, W, name, float(err.max()) / scale, float((err / (ref.abs() + rms)).max()),
            float((got.argmax(1) != ref.argmax(1)).float().mean())), flush=True)
