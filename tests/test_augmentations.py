"""The data path of SURVEY §8 f-2 on the host: the reference's aug-string tables, the draw order of the random streams
(so a seeded run walks the same decisions as the albumentations pipeline it restates: augmentations.py:50-134 of the
reference), and the integer definitions of the colour transforms.  albumentations / OpenCV are not installable here, so
these pin the DEFINITIONS, not outputs of the reference (stated in the module docstring)."""
import json
import os
import random

import numpy as np
import pytest

import synth
from hiast_amd.sseg.datasets import augmentations as A


def _img(seed, h=40, w=64):
    return synth.images_u8(seed, 1, h, w)[0]


def test_random_stream_of_flip_crop_resize():
    """Compose: 1 draw; HorizontalFlip: 1 draw; RandomSizedCrop: 1 draw, then randint, random, random"""
    img, lbl = _img(1, 400, 800), synth.pseudo_labels(2, 1, 400, 800, 19)[0]
    f = A.flip_crop_resize(64, 128, (100, 300), 2)
    random.seed(5)
    out = f(image=img, mask=lbl)
    random.seed(5)
    random.random()                       # Compose p
    flip = random.random() < 0.5
    random.random()                       # RandomSizedCrop p
    ch = random.randint(100, 300)
    hs, ws = random.random(), random.random()
    cw = int(ch * 2)
    y1, x1 = int((400 - ch) * hs), int((800 - cw) * ws)
    src_i = img[:, ::-1] if flip else img
    src_l = lbl[:, ::-1] if flip else lbl
    want_i = A._resize_img(np.ascontiguousarray(src_i[y1:y1 + ch, x1:x1 + cw]), 64, 128)
    want_l = A._resize_mask(np.ascontiguousarray(src_l[y1:y1 + ch, x1:x1 + cw]), 64, 128)
    assert np.array_equal(out["image"], want_i) and np.array_equal(out["mask"], want_l)
    assert set(np.unique(out["mask"])) <= set(np.unique(lbl))          # nearest: no new label values


def test_some_of_is_the_references_rewrite():
    """uniform choice WITHOUT replacement of n = 3 through RandomState(random.randint(0, 2**32 - 1)); each chosen
    transform then applies itself with its own p = 0.5"""
    calls = []

    class Probe(A._Aug):
        image_only = True

        def __init__(self, k):
            self.k, self.p = k, 0.5

        def apply(self, image, masks):
            calls.append(self.k)
            return image, masks

    so = A.SomeOf([Probe(k) for k in range(8)], n=3)
    img = _img(3)
    random.seed(11)
    so(image=img, mask=None)
    random.seed(11)
    random.random()
    picked = np.random.RandomState(random.randint(0, 2 ** 32 - 1)).choice(8, size=3, replace=False)
    want = [int(k) for k in picked if random.random() < 0.5]
    assert calls == want and len(set(picked.tolist())) == 3
    pool = A.complex_color_aug().transforms
    assert [type(t).__name__ for t in pool] == ["ColorJitter", "GaussianBlur", "RandomContrast", "RandomBrightness",
                                                "Posterize", "Equalize", "Solarize", "ToGray"]
    assert all(t.p == 0.5 for t in pool) and A.complex_color_aug().n == 3
    assert pool[1].blur == (3, 41) and pool[2].limit == (0, 3) and pool[3].limit == (-0.5, 0.5) and pool[4].bits == (4, 4)


def test_integer_colour_transforms():
    img = _img(4)
    # RandomContrast / RandomBrightness: one LUT, alpha * v (+ beta * 255), clipped
    assert np.array_equal(A.RandomContrast((0, 3)).apply(img, [], alpha=2.5)[0],
                          np.clip(np.arange(256, dtype=np.float32) * 2.5, 0, 255).astype(np.uint8)[img])
    assert np.array_equal(A.RandomBrightness(0.5).apply(img, [], beta=-0.25)[0],
                          np.clip(np.arange(256, dtype=np.float32) - 63.75, 0, 255).astype(np.uint8)[img])
    # Posterize(4): the four high bits; Solarize(128): v -> 255 - v from the threshold on
    assert np.array_equal(A.Posterize(4).apply(img, [], bits=4)[0], img & 0xF0)
    sol = A.Solarize(128).apply(img, [], threshold=128.0)[0]
    assert np.array_equal(sol, np.where(img < 128, img, 255 - img))
    # ToGray: OpenCV's fixed-point luma on all three channels
    g = A.ToGray().apply(img, [])[0]
    want = ((img[..., 0].astype(np.int64) * 4899 + img[..., 1].astype(np.int64) * 9617 + img[..., 2].astype(np.int64) * 1868
             + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(g[..., 0], want) and np.array_equal(g[..., 1], want) and np.array_equal(g[..., 2], want)
    # Equalize: per channel, monotone LUT, full range
    e = A.Equalize().apply(img, [])[0]
    for c in range(3):
        order = np.argsort(img[..., c].ravel(), kind="stable")
        assert (np.diff(e[..., c].ravel()[order].astype(np.int32)) >= 0).all() and e[..., c].max() == 255
    const = np.full((8, 8, 3), 77, np.uint8)
    assert np.array_equal(A.Equalize().apply(const, [])[0], const)
    # GaussianBlur: odd kernel from the limits, constant image unchanged, mass preserved
    random.seed(3)
    ks = [A.GaussianBlur((3, 41)).params(img)["ksize"] for _ in range(200)]
    assert all(k % 2 == 1 and 3 <= k <= 41 for k in ks) and len(set(ks)) > 10
    assert np.array_equal(A.GaussianBlur((3, 41)).apply(const, [], ksize=21, sigma=0)[0], const)
    k = A._gaussian_kernel_cv(5, 0)
    assert abs(k.sum() - 1) < 1e-6 and k[2] == k.max() and np.allclose(k, k[::-1])
    # ColorJitter: neutral factors are the identity (hue through the HSV round trip: within rounding)
    out = A.ColorJitter().apply(img, [], factors=(1.0, 1.0, 1.0, 0.0), order=[0, 1, 2, 3])[0]
    assert np.array_equal(out, img)
    rt = A._hsv_to_rgb_u8(A._rgb_to_hsv_u8(img))
    assert np.abs(rt.astype(np.int32) - img.astype(np.int32)).max() <= 4


def test_fda_swaps_low_frequency_amplitudes():
    src, trg = _img(5, 32, 48), _img(6, 32, 48)
    out = A.fourier_domain_adaptation(src, trg, beta=0.0)           # only the DC term: the target's mean brightness
    for c in range(3):
        assert abs(out[..., c].astype(np.float64).mean() - trg[..., c].astype(np.float64).mean()) < 1.5
    same = A.fourier_domain_adaptation(src, src, beta=0.1)
    assert np.abs(same.astype(np.int32) - src.astype(np.int32)).max() <= 1
    full = A.fourier_domain_adaptation(src, trg, beta=0.5)          # every amplitude from the target, phases from the source
    assert np.abs(np.abs(np.fft.fft2(full.astype(np.float32), axes=(0, 1)))[1:8, 1:8]
                  - np.abs(np.fft.fft2(trg.astype(np.float32), axes=(0, 1)))[1:8, 1:8]).mean() < \
        0.2 * np.abs(np.fft.fft2(trg.astype(np.float32), axes=(0, 1)))[1:8, 1:8].mean()


def test_dataset_aug_tables(tmp_path):
    """the per-dataset aug-string tables of the reference (cityscapes / gtav / synthia / oxford _dataset.py)"""
    from hiast_amd.tools import synth_data
    from hiast_amd.utils.registry import register  # noqa: F401
    from hiast_amd.utils.registry.registries import DATASET
    c = synth_data.synthetic_cfg(str(tmp_path), n_train=2, n_val=1, h=32, w=64)
    c.dataset.source.json_path, c.dataset.source.image_dir = c.dataset.target.json_path, c.dataset.target.image_dir
    mk = lambda name, aug: DATASET[name](c, c.dataset.target.json_path, c.dataset.target.image_dir, aug_type=[aug])
    crop = lambda ds: ds.aug_fun.transforms[1]
    assert crop(mk("Cityscapes", "MS")).mm == (341, 1000) and crop(mk("GTAV", "MS")).mm == (341, 950)
    assert crop(mk("SYNTHIA", "MS")).mm == (341, 640)
    assert crop(mk("Cityscapes", "OMS")).mm == (341, 1000) and crop(mk("Oxford", "OMS")).mm == (341, 900)
    assert abs(crop(mk("Oxford", "OMS")).ratio - 1280 / 960) < 1e-12 and (crop(mk("Oxford", "OMS")).h, crop(mk("Oxford", "OMS")).w) == (768, 1024)
    d = mk("GTAV", "DACS").aug_fun.transforms
    assert (d[0].h, d[0].w, d[1].h, d[1].w) == (720, 1280, 512, 512)
    assert (mk("SYNTHIA", "DACS").aug_fun.transforms[0].h, mk("Cityscapes", "DACS").aug_fun.transforms[0].w) == (760, 1024)
    assert isinstance(mk("Cityscapes", "CCA").aug_fun, A.SomeOf) and isinstance(mk("Oxford", "SCA").aug_fun, A.Compose)
    for name, bad in (("GTAV", "CCA"), ("SYNTHIA", "OMS"), ("Oxford", "MS"), ("GTAV", "FDA-Source")):
        with pytest.raises(ValueError):
            mk(name, bad)
    f = mk("GTAV", "FDA-Target").aug_fun
    assert isinstance(f, A.FDA) and f.p == 1.0 and f.beta == (0, 0.001) and len(f.refs) == 2
    img = _img(7, 32, 64)
    random.seed(1)
    out = f(image=img, mask=np.zeros((32, 64), np.uint8))["image"]
    assert out.shape == img.shape and out.dtype == np.uint8
    with pytest.raises(AssertionError):
        mk("Cityscapes", "FDA-Target")        # only valid for Cityscapes -> Oxford (cityscapes_dataset.py:42)


def test_serial_multi_view_and_index_seeding():
    img, lbl = _img(8), synth.pseudo_labels(9, 1, 40, 64, 19)[0]
    views = [A.resize(20, 32), A.complex_color_aug()]
    a = A.aug(views, img, lbl, index=7)
    b = A.aug(views, img, lbl, index=7)
    assert np.array_equal(a[0][1], b[0][1]) and a[0][0].shape == (20, 32, 3)
    assert np.array_equal(a[1][0], a[1][1])           # colour transforms leave the label alone


def test_random_draws_per_transform_follow_albumentations_1_0_3():
    """number and ORDER of the random.random() calls behind each transform's parameters (albumentations 1.0.3):
    RandomContrast / RandomBrightness are RandomBrightnessContrast subclasses whose get_params draws alpha THEN beta, the
    degenerate uniform(0, 0) included (2 draws each); FDA draws beta (get_params) BEFORE the reference image
    (get_params_dependent_on_targets).  A transform that drew one value less would shift the stream every later
    transform of a 'CCA' SomeOf sees."""
    img = _img(21)

    class Counting(random.Random):
        def __init__(self, seed):
            super().__init__(seed)
            self.n = 0

        def random(self):
            self.n += 1
            return super().random()

    def draws(fn, seed=31):
        """-> (result of fn, number of random.random() calls, the values drawn) with the module RNG replaced"""
        rng = Counting(seed)
        saved = {k: getattr(random, k) for k in ("random", "uniform", "randint", "choice")}
        random.random, random.uniform, random.randint, random.choice = rng.random, rng.uniform, rng.randint, rng.choice
        try:
            out = fn()
        finally:
            for k, v in saved.items():
                setattr(random, k, v)
        ref = random.Random(seed)
        return out, rng.n, [ref.random() for _ in range(rng.n)]

    p, n, u = draws(lambda: A.RandomContrast(limit=(0, 3)).params(img))
    assert n == 2 and p == {"alpha": 1.0 + (0 + 3 * u[0])}
    p, n, u = draws(lambda: A.RandomBrightness(limit=0.5).params(img))
    assert n == 2 and p == {"beta": 0.0 + (-0.5 + 1.0 * u[1])}          # the SECOND draw: alpha's dummy draw comes first
    refs = ["a", "b", "c", "d", "e"]
    seen = []

    def read(path):
        seen.append(path)
        return _img(22)
    p, n, u = draws(lambda: A.FDA(refs, beta_limit=0.1, read_fn=read).params(img))
    assert abs(p["beta"] - 0.1 * u[0]) < 1e-15                           # beta first ...
    again = Counting(31)             # (a Random subclass that overrides random() draws choice() through random() too)
    again.random()
    assert seen == [again.choice(refs)]                                  # ... then the reference image
