"""Is the fp32-class inference forward bit-reproducible — run to run, alone vs beside another forward on a second stream,
B = 2 alone vs as half of a split batch of 4?  Prints the number of differing logits per comparison."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth  # noqa: E402
from hiast_amd import functional as HF  # noqa: E402
from hiast_amd.utils.registry import register  # noqa: E402,F401
from hiast_amd.utils.registry.registries import MODEL  # noqa: E402
from hiast_amd.utils.default_config import get_default_cfg  # noqa: E402
from make_golden import seeded_state_dict  # noqa: E402

cfg = get_default_cfg()
cfg.model.type = "SelfTrainingSegmentor"
net = MODEL["SelfTrainingSegmentor"](cfg)
net.load_state_dict({"seg_model." + k: v for k, v in seeded_state_dict(net.seg_model, 782).items()})
net = net.cuda().eval()
x = torch.from_numpy(synth.normal_f32(950, (4, 3, 128, 256))).cuda()


def fwd(t):
    with torch.no_grad():
        return net(t, lowres=True)["logits_lowres"].float().clone()


def diff(a, b):
    return int((a != b).sum()), float((a - b).abs().max())


ref = fwd(x[:2])
torch.cuda.synchronize()
for i in range(3):
    print("repeat %d of B=2 alone:" % i, diff(fwd(x[:2]), ref))
with torch.no_grad():
    sp = HF.eval_forward_split(net, x, 2)["logits_lowres"].float().clone()
torch.cuda.synchronize()
print("first half of a split batch of 4 (second half on a side stream):", diff(sp[:2], ref))
print("second half vs B=2 alone:", diff(sp[2:], fwd(x[2:])))
side = torch.cuda.Stream()
for i in range(3):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        other = fwd(x[2:])
    mine = fwd(x[:2])
    torch.cuda.synchronize()
    print("B=2 beside another forward on a second stream, repeat %d:" % i, diff(mine, ref), diff(other, fwd(x[2:])))
y4 = fwd(x)
print("B=4 in one launch sequence, first half vs B=2 alone:", diff(y4[:2], ref))
# stage by stage: where does a difference start?  (stem = library convolution)
bb = net.seg_model.backbone
with torch.no_grad():
    s2 = bb.conv1(x[:2].contiguous(memory_format=torch.channels_last))
    s4 = bb.conv1(x.contiguous(memory_format=torch.channels_last))
    print("library stem convolution, B=4 vs B=2:", diff(s4[:2], s2), "| repeat B=2:", diff(bb.conv1(x[:2].contiguous(memory_format=torch.channels_last)), s2))
