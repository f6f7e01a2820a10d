"""Adversarial warm-up step on one MI355X: images/s of AdversarialWarmupTrainer's iteration
(source fwd + target fwd, generator backward + Adam, discriminator backward + Adam) at bs B of 1024x512 crops,
data resident on the device, bf16 autocast; plus the K15 kernels against their HBM roofline.

    python tools/bench_warmup.py [B=4] [steps=10] [warmup=3]        -> one JSON line
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    from hiast_amd import functional as HF, kernels as K
    from hiast_amd.utils import utils
    from hiast_amd.utils.default_config import get_default_cfg
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import MODEL
    dev = torch.device("cuda")
    C, H, W, h, w = 19, 512, 1024, 64, 128

    # --- K15 against the HBM roofline (algorithmic bytes: fwd = low-res read + map write;
    #     bwd = low-res read + map-gradient read + scratch write + scratch read + low-res write)
    z = torch.randn(B, C, h, w, device=dev) * 3
    g = torch.randn(B, C, H, W, device=dev)
    lr_bytes, map_bytes = B * C * h * w * 4, B * C * H * W * 4
    kern = {}
    for ent in (False, True):
        tf = timeit(lambda: K.dinput_fwd(z, H, W, ent))
        tb = timeit(lambda: K.dinput_bwd(z, g, ent))
        kern["dinput_fwd_%s" % ("entropy" if ent else "softmax")] = {
            "ms": round(tf, 4), "GBps": round((lr_bytes + map_bytes) / tf / 1e6, 1)}
        kern["dinput_bwd_%s" % ("entropy" if ent else "softmax")] = {
            "ms": round(tb, 4), "GBps": round((2 * lr_bytes + 3 * map_bytes) / tb / 1e6, 1)}
    del z, g

    # --- the training iteration
    cfg = get_default_cfg()
    cfg.dataset.num_classes = C
    cfg.model.type = "AdversarialWarmupSegmentor"
    cfg.model.discriminator.is_enabled = True
    cfg.train.optimizer = "Adam"
    cfg.train.lr = 2.5e-4
    torch.manual_seed(888)
    model = MODEL[cfg.model.type](cfg).to(dev).train()
    g_opt, d_opt = utils.init_optimizers(cfg, model)
    s_img = torch.randn(B, 3, H, W, device=dev)
    t_img = torch.randn(B, 3, H, W, device=dev)
    s_lbl = torch.randint(0, C, (B, H, W), device=dev)
    s_lbl[torch.rand(B, H, W, device=dev) < 0.05] = 255

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses = model(s_img, t_img, s_lbl)
        g_loss = sum(torch.mean(v) for k, v in losses.items() if "D_" not in k)
        g_opt.zero_grad(set_to_none=True)
        HF.enable_wgrad_overlap(True)
        try:
            g_loss.backward()
        finally:
            HF.enable_wgrad_overlap(False)
        HF.wgrad_stream_join()
        g_opt.step()
        d_opt.zero_grad(set_to_none=True)
        torch.mean(losses["D_loss"]).backward()
        d_opt.step()
        return losses

    for _ in range(warm):
        losses = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"metric": "adversarial warm-up images/s (source+target pair per image slot)", "value": round(B / dt, 2),
                      "unit": "image pairs/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": round(dt * 1e3, 2),
                      "dtype": "bf16", "data": "synthetic", "config": {"workload": "AdversarialWarmupTrainer step, bs %d, "
                                                                       "1024x512, MSE discriminator loss, MinEnt 3.0" % B},
                      "losses": {k: round(float(v), 5) for k, v in losses.items()}, "kernels": kern,
                      "hbm_peak_GBps": 8000}))


if __name__ == "__main__":
    main()
