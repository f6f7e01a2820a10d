"""CPU image augmentation for the self-training configs (reference: sseg/datasets/augmentations.py,
a thin layer over albumentations 1.0.3, which is not available offline).  Own numpy/PIL versions of
the pieces the HIAST configs name: 'PRS-h-w' resize, 'MS' (flip + random sized crop -> 512x1024) and
a reduced 'CCA' colour pool.  Each aug is a callable `f(image=, mask=|masks=) -> dict` like an
albumentations transform; randomness comes from Python's `random`, seeded per index on the plain
path exactly as the reference does (augmentations.py:11-17)."""
import random

import numpy as np
from PIL import Image


def _resize_img(img, h, w):
    return np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))


def _resize_mask(m, h, w):
    return np.asarray(Image.fromarray(m).resize((w, h), Image.NEAREST))


class _Aug:
    def apply(self, image, masks):
        raise NotImplementedError

    def __call__(self, image, mask=None, masks=None):
        if masks is not None:
            img, ms = self.apply(image, list(masks))
            return {"image": img, "masks": ms}
        img, ms = self.apply(image, [mask])
        return {"image": img, "mask": ms[0]}


class Resize(_Aug):
    def __init__(self, h, w):
        self.h, self.w = h, w

    def apply(self, image, masks):
        if image.shape[:2] == (self.h, self.w):
            return image, masks
        return _resize_img(image, self.h, self.w), [_resize_mask(m, self.h, self.w) for m in masks]


class FlipCropResize(_Aug):
    """HorizontalFlip(p=.5) then RandomSizedCrop(min_max_height, height, width, w2h_ratio)"""

    def __init__(self, height, width, min_max_height, w2h_ratio):
        self.h, self.w, self.mm, self.ratio = height, width, min_max_height, w2h_ratio

    def apply(self, image, masks):
        if random.random() < 0.5:
            image = image[:, ::-1]
            masks = [m[:, ::-1] for m in masks]
        H, W = image.shape[:2]
        ch = random.randint(min(self.mm[0], H), min(self.mm[1], H))
        cw = min(int(ch * self.ratio), W)
        y0 = int(random.random() * (H - ch + 1))
        x0 = int(random.random() * (W - cw + 1))
        image = np.ascontiguousarray(image[y0:y0 + ch, x0:x0 + cw])
        masks = [np.ascontiguousarray(m[y0:y0 + ch, x0:x0 + cw]) for m in masks]
        return _resize_img(image, self.h, self.w), [_resize_mask(m, self.h, self.w) for m in masks]


class ColorAug(_Aug):
    """pick `n` of a pool of photometric ops (labels untouched)"""

    def __init__(self, n=2, strong=True):
        self.n, self.strong = n, strong

    @staticmethod
    def _brightness_contrast(img):
        a = 1.0 + random.uniform(-0.3, 0.3)
        b = random.uniform(-0.2, 0.2) * 255
        return np.clip(img.astype(np.float32) * a + b, 0, 255).astype(np.uint8)

    @staticmethod
    def _gamma(img):
        g = random.uniform(0.7, 1.5)
        return (255.0 * (img.astype(np.float32) / 255.0) ** g).astype(np.uint8)

    @staticmethod
    def _channel_gain(img):
        gain = np.array([random.uniform(0.8, 1.2) for _ in range(3)], np.float32)
        return np.clip(img.astype(np.float32) * gain, 0, 255).astype(np.uint8)

    @staticmethod
    def _gray(img):
        g = img.astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)
        return np.repeat(g[..., None], 3, axis=2).astype(np.uint8)

    @staticmethod
    def _noise(img):
        sigma = random.uniform(3, 12)
        g = np.random.Generator(np.random.SFC64(np.random.randint(0, 2 ** 31 - 1)))     # child of the global stream
        n = g.standard_normal(img.shape, dtype=np.float32)                               # float32 draws: 3x cheaper
        n *= sigma
        n += img
        return np.clip(n, 0, 255, out=n).astype(np.uint8)

    def apply(self, image, masks):
        pool = [self._brightness_contrast, self._gamma, self._channel_gain, self._noise]
        if self.strong:
            pool.append(self._gray)
        for f in random.sample(pool, self.n):
            if random.random() < 0.8:
                image = f(image)
        return image, masks


def resize(h, w):
    return Resize(h, w)


def flip_crop_resize(height, width, min_max_height, w2h_ratio):
    return FlipCropResize(height, width, min_max_height, w2h_ratio)


def simple_color_aug():
    return ColorAug(n=1, strong=False)


def complex_color_aug():
    return ColorAug(n=2, strong=True)


def _apply(fun, img, lbl):
    if fun is None:
        return img, lbl
    if isinstance(lbl, (list, tuple)):
        r = fun(image=img, masks=lbl)
        return r["image"], r["masks"]
    r = fun(image=img, mask=lbl)
    return r["image"], r["mask"]


def aug(aug_fun, img, lbl, index=None):
    """one aug -> (img, lbl); a list of augs -> serial multi-view ([img_0, img_1], [lbl_0, lbl_1]):
    view k is aug_k applied to view k-1's output (augmentations.py:31-47 of the reference)."""
    if index is not None:
        random.seed(index)
    if isinstance(aug_fun, (list, tuple)):
        imgs, lbls = [], []
        cur_i, cur_l = img, lbl
        for f in aug_fun:
            cur_i, cur_l = _apply(f, cur_i, cur_l)
            imgs.append(cur_i)
            lbls.append(cur_l)
        return imgs, lbls
    return _apply(aug_fun, img, lbl)
