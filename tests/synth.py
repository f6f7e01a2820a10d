"""Seeded synthetic inputs shared by the golden generator, the parity tests, smoke() and bench.py.

Everything is drawn from numpy's PCG64 (bit-stable across platforms for a given numpy), so
fixtures only need to hold the reference's OUTPUTS; inputs are regenerated from the seed.
"""
import numpy as np


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def logits_lr(seed, B, C, h, w, sigma=3.0):
    """low-res head logits; sigma 3 spreads the max-probs over (1/C, 1)"""
    return (rng(seed).standard_normal((B, C, h, w)) * sigma).astype(np.float32)


def smooth_logits_lr(seed, B, C, h, w, sigma=4.0):
    """spatially smooth logits (blobby argmax maps, like a trained segmentor's output)"""
    g = rng(seed)
    ch, cw = max(2, h // 8 + 1), max(2, w // 8 + 1)
    coarse = g.standard_normal((B, C, ch, cw)).astype(np.float32) * sigma
    ys = np.linspace(0, ch - 1, h)
    xs = np.linspace(0, cw - 1, w)
    y0 = np.floor(ys).astype(int).clip(0, ch - 2)
    x0 = np.floor(xs).astype(int).clip(0, cw - 2)
    fy = (ys - y0).astype(np.float32)[None, None, :, None]
    fx = (xs - x0).astype(np.float32)[None, None, None, :]
    a = coarse[:, :, y0][:, :, :, x0]
    b = coarse[:, :, y0][:, :, :, x0 + 1]
    c = coarse[:, :, y0 + 1][:, :, :, x0]
    d = coarse[:, :, y0 + 1][:, :, :, x0 + 1]
    out = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    out += g.standard_normal((B, C, h, w)).astype(np.float32) * 0.3
    return np.ascontiguousarray(out, dtype=np.float32)


def probs_and_labels(seed, B, H, W, C):
    """direct Stage-B inputs: max-prob float32 in (1/C, 1], label in [0, C)"""
    g = rng(seed)
    u = g.random((B, H, W), dtype=np.float32)
    p = (1.0 / C + (1 - 1.0 / C) * np.sqrt(u)).astype(np.float32)
    # a block of exact duplicates / bin-edge values to stress the quantile interpolation
    p[:, : H // 8, : W // 8] = np.float32(0.90039062)
    lbl = g.integers(0, C, size=(B, H, W), dtype=np.int64)
    lbl[:, :, : W // 16] = 0          # an over-represented class
    if C > 3:
        lbl[lbl == C - 2] = C - 3     # and an absent one (class C-2 never predicted)
    return p, lbl


def pseudo_labels(seed, B, H, W, C, p_ignore=0.4, dtype=np.uint8):
    g = rng(seed)
    l = g.integers(0, C, size=(B, H, W))
    l[g.random((B, H, W)) < p_ignore] = 255
    return l.astype(dtype)


def images_u8(seed, B, H, W):
    g = rng(seed)
    return g.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)


def normal_f32(seed, shape, sigma=1.0):
    return (rng(seed).standard_normal(shape) * sigma).astype(np.float32)
