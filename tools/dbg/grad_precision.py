"""cosine between the gradients of ONE self-training step in fp32 (O0), fp16 (O1, library) and bf16 (O1, own kernels),
same calibrated checkpoint and batch: how much of the 16-bit gradient is signal, layer by layer"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_precision as T  # noqa: E402
import synth  # noqa: E402


class _F:
    def mktemp(self, n):
        import tempfile
        return tempfile.mkdtemp()


def grads(root, opt, dt):
    tr = T._trainer(root, opt, dt)
    dev = tr.device
    weak = synth.normal_f32(800, (T.B, 3, T.H, T.W))
    losses = tr.train_on(torch.from_numpy(weak).to(dev), torch.from_numpy((weak * 1.05 + 0.02).astype(np.float32)).to(dev),
                         torch.from_numpy(synth.pseudo_labels(810, T.B, T.H, T.W, T.C, 0.4)).to(dev))
    g = sum(torch.mean(v) for v in losses.values())
    tr.g_optimizer.zero_grad(set_to_none=True)
    (g * (1024.0 if dt == "fp16" and opt != "O0" else 1.0)).backward()
    sc = 1024.0 if dt == "fp16" and opt != "O0" else 1.0
    out = {k: (p.grad.detach().double().cpu() / sc) for k, p in tr.model.module.named_parameters() if p.grad is not None}
    del tr
    torch.cuda.empty_cache()
    return out


def main():
    root = T.make_checkpoint(_F())
    g32, g32b, g16, gb = grads(root, "O0", "bf16"), grads(root, "O0", "bf16"), grads(root, "O1", "fp16"), grads(root, "O1", "bf16")
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
    keys = [k for k in g32 if any(s in k for s in ("aspp.conv2d_list.0.weight", "layer4.2.conv3", "layer4.0.conv1", "layer3.20.conv2",
                                                    "layer3.10.conv1", "layer3.0.conv1", "layer2.0.conv1", "layer1.0.conv1", "backbone.conv1.weight"))]
    print("%-55s %10s %10s %10s   |g32|" % ("parameter", "fp32again", "fp16", "bf16"))
    for k in keys:
        print("%-55s %10.4f %10.4f %10.4f   %.3e" % (k[10:], cos(g32b[k], g32[k]), cos(g16[k], g32[k]), cos(gb[k], g32[k]), float(g32[k].norm())))
    allc = lambda g: float(np.mean([cos(g[k], g32[k]) for k in g32]))
    print("mean over all 112 tensors: fp32again %.4f fp16 %.4f bf16 %.4f" % (allc(g32b), allc(g16), allc(gb)))


if __name__ == "__main__":
    main()
