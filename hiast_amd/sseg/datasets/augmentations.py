"""CPU image augmentation for every aug string of the reference's datasets (reference: sseg/datasets/augmentations.py,
a thin layer over albumentations 1.0.3 + OpenCV, neither of which is available offline).

Own numpy / PIL restatement of the albumentations transforms the reference composes, with the SAME stream of
`random` / `np.random` draws in the same order (every transform: one `random.random()` for its own probability, then
its parameters), the reference's rewritten `SomeOf` (uniform choice WITHOUT replacement through
`np.random.RandomState(random.randint(0, 2**32 - 1))`, each chosen transform then applied with its own p), the serial
multi-view rule and the per-index seeding of `aug()` (augmentations.py:11-47,106-134):

    'PRS-h-w'     resize(h, w)                                          A.Resize
    'MS' / 'OMS'  flip_crop_resize(h, w, min_max_height, w2h_ratio)     A.HorizontalFlip(.5) + A.RandomSizedCrop
    'DACS'        resize_crop(h, w, hc, wc)                             A.Resize + A.RandomCrop
    'SCA'         simple_color_aug()                                    ColorJitter(.5), GaussianBlur((3, 41), .5)
    'CCA'         complex_color_aug()                                   SomeOf(n = 3) of the 8-transform pool
    'FDA-*'       fda(json, image_dir, beta_limit)                      A.FDA (Fourier domain adaptation)

Integer look-up-table transforms (RandomContrast, RandomBrightness, Posterize, Solarize, ToGray's fixed-point weights,
Equalize's histogram LUT) follow OpenCV's integer arithmetic; resampling and blurring go through PIL / numpy, so pixel
values of those can differ from OpenCV's in the last bit — parity of the colour operations is NOT pinned by a fixture
(the reference's dependency cannot run here); the geometry and the RNG stream are tested against their definitions.
Each aug is a callable `f(image=, mask=|masks=) -> dict` like an albumentations transform."""
import json
import os
import random

import numpy as np
from PIL import Image, ImageFilter


def _resize_img(img, h, w):
    return np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))


def _resize_mask(m, h, w):
    return np.asarray(Image.fromarray(m).resize((w, h), Image.NEAREST))


class _Aug:
    """albumentations' BasicTransform.__call__: `if force_apply or random.random() < p`, then parameters, then apply"""
    p = 1.0
    image_only = False

    def params(self, image):
        return {}

    def apply(self, image, masks, **params):
        raise NotImplementedError

    def __call__(self, image=None, mask=None, masks=None, force_apply=False):
        ms = list(masks) if masks is not None else [mask]
        if force_apply or random.random() < self.p:
            image, ms = self.apply(image, ms, **self.params(image))
        if masks is not None:
            return {"image": image, "masks": ms}
        return {"image": image, "mask": ms[0]}


class Compose(_Aug):
    """A.Compose(p = 1.0): one probability draw, then the transforms in order"""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, image=None, mask=None, masks=None, force_apply=False):
        ms = list(masks) if masks is not None else [mask]
        if force_apply or random.random() < self.p:
            for t in self.transforms:
                r = t(image=image, masks=ms)
                image, ms = r["image"], r["masks"]
        if masks is not None:
            return {"image": image, "masks": ms}
        return {"image": image, "mask": ms[0]}


class SomeOf(Compose):
    """the reference's rewritten SomeOf (augmentations.py:106-134)"""

    def __init__(self, transforms, n, replace=False, p=1.0):
        super().__init__(transforms)
        self.n, self.replace, self.p = n, replace, p

    def __call__(self, image=None, mask=None, masks=None, force_apply=False):
        ms = list(masks) if masks is not None else [mask]
        if force_apply or random.random() < self.p:
            rs = np.random.RandomState(random.randint(0, 2 ** 32 - 1))
            for i in rs.choice(len(self.transforms), size=self.n, replace=self.replace):
                r = self.transforms[int(i)](image=image, masks=ms)
                image, ms = r["image"], r["masks"]
        if masks is not None:
            return {"image": image, "masks": ms}
        return {"image": image, "mask": ms[0]}


# ------------------------------------------------------------------------------------------------- geometry
class Resize(_Aug):
    def __init__(self, h, w, p=1.0):
        self.h, self.w, self.p = h, w, p

    def apply(self, image, masks):
        if image.shape[:2] == (self.h, self.w):
            return image, masks
        return _resize_img(image, self.h, self.w), [_resize_mask(m, self.h, self.w) for m in masks]


class HorizontalFlip(_Aug):
    def __init__(self, p=0.5):
        self.p = p

    def apply(self, image, masks):
        return np.ascontiguousarray(image[:, ::-1]), [np.ascontiguousarray(m[:, ::-1]) for m in masks]


def _crop_coords(H, W, ch, cw, h_start, w_start):
    """albumentations.functional.get_random_crop_coords"""
    y1 = int((H - ch) * h_start)
    x1 = int((W - cw) * w_start)
    return y1, y1 + ch, x1, x1 + cw


class RandomSizedCrop(_Aug):
    """crop_height = random.randint(min, max), h_start, w_start = random.random() x 2, crop width =
    int(crop_height * w2h_ratio); crop, then resize to (height, width)"""

    def __init__(self, min_max_height, height, width, w2h_ratio=1.0, p=1.0):
        self.mm, self.h, self.w, self.ratio, self.p = tuple(min_max_height), height, width, w2h_ratio, p

    def params(self, image):
        ch = random.randint(self.mm[0], self.mm[1])
        return {"h_start": random.random(), "w_start": random.random(), "ch": ch, "cw": int(ch * self.ratio)}

    def apply(self, image, masks, h_start, w_start, ch, cw):
        H, W = image.shape[:2]
        ch, cw = min(ch, H), min(cw, W)      # (albumentations raises on a crop larger than the image; small synthetic
                                             # frames are cropped whole instead)
        y1, y2, x1, x2 = _crop_coords(H, W, ch, cw, h_start, w_start)
        image = np.ascontiguousarray(image[y1:y2, x1:x2])
        masks = [np.ascontiguousarray(m[y1:y2, x1:x2]) for m in masks]
        return _resize_img(image, self.h, self.w), [_resize_mask(m, self.h, self.w) for m in masks]


class RandomCrop(_Aug):
    def __init__(self, height, width, p=1.0):
        self.h, self.w, self.p = height, width, p

    def params(self, image):
        return {"h_start": random.random(), "w_start": random.random()}

    def apply(self, image, masks, h_start, w_start):
        H, W = image.shape[:2]
        if self.h > H or self.w > W:
            raise ValueError("RandomCrop: crop %dx%d is larger than the image %dx%d" % (self.h, self.w, H, W))
        y1, y2, x1, x2 = _crop_coords(H, W, self.h, self.w, h_start, w_start)
        return np.ascontiguousarray(image[y1:y2, x1:x2]), [np.ascontiguousarray(m[y1:y2, x1:x2]) for m in masks]


# ------------------------------------------------------------------------------------------------- colour
def _lut(img, lut):
    return np.asarray(lut, np.uint8)[img]


def _gray_cv(img):
    """cv2.cvtColor(RGB2GRAY) on uint8: fixed point (R*4899 + G*9617 + B*1868 + 8192) >> 14"""
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _brightness_contrast_lut(alpha, beta, beta_by_max=True, img=None):
    """albumentations.functional._brightness_contrast_adjust_uint: one 256-entry LUT"""
    lut = np.arange(0, 256, dtype=np.float32)
    if alpha != 1:
        lut *= alpha
    if beta != 0:
        lut += beta * (255.0 if beta_by_max else float(np.mean(img)))
    return np.clip(lut, 0, 255).astype(np.uint8)


class RandomContrast(_Aug):
    image_only = True

    def __init__(self, limit=0.2, p=0.5):
        self.limit = (-limit, limit) if np.isscalar(limit) else tuple(limit)
        self.p = p

    def params(self, image):
        # albumentations 1.0.3: a RandomBrightnessContrast subclass with brightness_limit = (0, 0) — get_params draws
        # alpha, then beta; the degenerate uniform(0, 0) still consumes one random.random()
        alpha = 1.0 + random.uniform(self.limit[0], self.limit[1])
        random.uniform(0.0, 0.0)
        return {"alpha": alpha}

    def apply(self, image, masks, alpha):
        return _lut(image, _brightness_contrast_lut(alpha, 0.0)), masks


class RandomBrightness(_Aug):
    image_only = True

    def __init__(self, limit=0.2, p=0.5):
        self.limit = (-limit, limit) if np.isscalar(limit) else tuple(limit)
        self.p = p

    def params(self, image):
        # RandomBrightnessContrast with contrast_limit = (0, 0): the (degenerate) alpha draw comes first, then beta
        random.uniform(0.0, 0.0)
        return {"beta": 0.0 + random.uniform(self.limit[0], self.limit[1])}

    def apply(self, image, masks, beta):
        return _lut(image, _brightness_contrast_lut(1.0, beta)), masks


class Posterize(_Aug):
    image_only = True

    def __init__(self, num_bits=4, p=0.5):
        self.bits, self.p = (num_bits, num_bits) if np.isscalar(num_bits) else tuple(num_bits), p

    def params(self, image):
        return {"bits": random.randint(self.bits[0], self.bits[1])}

    def apply(self, image, masks, bits):
        if bits == 0:
            return np.zeros_like(image), masks
        if bits == 8:
            return image, masks
        keep = np.uint8(~np.uint8(2 ** (8 - bits) - 1))
        return image & keep, masks


class Solarize(_Aug):
    image_only = True

    def __init__(self, threshold=128, p=0.5):
        self.thr, self.p = (threshold, threshold) if np.isscalar(threshold) else tuple(threshold), p

    def params(self, image):
        return {"threshold": random.uniform(self.thr[0], self.thr[1])}

    def apply(self, image, masks, threshold):
        lut = np.array([i if i < threshold else 255 - i for i in range(256)], np.uint8)
        return _lut(image, lut), masks


class ToGray(_Aug):
    image_only = True

    def __init__(self, p=0.5):
        self.p = p

    def apply(self, image, masks):
        g = _gray_cv(image)
        return np.repeat(g[..., None], 3, axis=2), masks


def _equalize_cv_channel(ch):
    """cv2.equalizeHist on one uint8 channel"""
    hist = np.bincount(ch.ravel(), minlength=256)
    nz = np.nonzero(hist)[0]
    total = int(ch.size)
    if len(nz) == 0 or hist[nz[0]] == total:
        return np.full_like(ch, nz[0] if len(nz) else 0)
    first = nz[0]
    scale = 255.0 / (total - hist[first])
    lut = np.zeros(256, np.uint8)
    s = 0
    for i in range(first + 1, 256):
        s += int(hist[i])
        lut[i] = min(255, max(0, int(round(s * scale))))
    return lut[ch]


class Equalize(_Aug):
    """A.Equalize(mode='cv', by_channels=True)"""
    image_only = True

    def __init__(self, p=0.5):
        self.p = p

    def apply(self, image, masks):
        return np.stack([_equalize_cv_channel(image[..., c]) for c in range(image.shape[2])], axis=2), masks


def _gaussian_kernel_cv(ksize, sigma):
    """cv2.getGaussianKernel; sigma <= 0 -> 0.3*((ksize-1)*0.5 - 1) + 0.8"""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def _blur_separable(img, kernel):
    """separable convolution with BORDER_REFLECT_101 ('mirror'), float32 accumulation, round to uint8
    (scipy's C loop: 41 taps on a 1024x512 view in ~12 ms, the numpy slice-sum took 65)"""
    from scipy.ndimage import correlate1d
    x = img.astype(np.float32)
    k = np.asarray(kernel, np.float32)
    x = correlate1d(x, k, axis=0, mode="mirror")
    x = correlate1d(x, k, axis=1, mode="mirror")
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


class GaussianBlur(_Aug):
    """A.GaussianBlur(blur_limit=(lo, hi), sigma_limit=0): odd ksize in [lo, hi]"""
    image_only = True

    def __init__(self, blur_limit=(3, 7), sigma_limit=0, p=0.5):
        self.blur, self.sigma, self.p = tuple(blur_limit), (0, sigma_limit) if np.isscalar(sigma_limit) else tuple(sigma_limit), p

    def params(self, image):
        ksize = random.randrange(self.blur[0], self.blur[1] + 1)
        if ksize != 0 and ksize % 2 != 1:
            ksize = (ksize + 1) % (self.blur[1] + 1)
        return {"ksize": ksize, "sigma": random.uniform(self.sigma[0], self.sigma[1])}

    def apply(self, image, masks, ksize, sigma):
        if ksize <= 1:
            return image, masks
        return _blur_separable(image, _gaussian_kernel_cv(ksize, sigma)), masks


def _rgb_to_hsv_u8(img):
    """cv2 COLOR_RGB2HSV on uint8: H in [0, 180), S, V in [0, 255]"""
    a = img.astype(np.float32)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    v = a.max(-1)
    mn = a.min(-1)
    d = v - mn
    s = np.where(v > 0, d / np.maximum(v, 1e-12) * 255.0, 0.0)
    dd = np.maximum(d, 1e-12)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd)) * 60.0
    h = np.where(d == 0, 0.0, h)
    h = np.where(h < 0, h + 360.0, h) / 2.0
    return np.stack([np.rint(h) % 180, np.rint(s), v], -1).astype(np.uint8)


def _hsv_to_rgb_u8(hsv):
    h = hsv[..., 0].astype(np.float32) * 2.0 / 60.0
    s = hsv[..., 1].astype(np.float32) / 255.0
    v = hsv[..., 2].astype(np.float32)
    i = np.floor(h).astype(np.int32) % 6
    f = h - np.floor(h)
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([r, g, b], -1)), 0, 255).astype(np.uint8)


class ColorJitter(_Aug):
    """A.ColorJitter(brightness=.2, contrast=.2, saturation=.2, hue=.2): four torchvision-style adjustments in a random
    order"""
    image_only = True

    def __init__(self, brightness=0.2, contrast=0.2, saturation=0.2, hue=0.2, p=0.5):
        self.b, self.c, self.s = (max(0, 1 - brightness), 1 + brightness), (max(0, 1 - contrast), 1 + contrast), \
            (max(0, 1 - saturation), 1 + saturation)
        self.h, self.p = (-hue, hue), p

    def params(self, image):
        b, c, s, h = (random.uniform(*self.b), random.uniform(*self.c), random.uniform(*self.s), random.uniform(*self.h))
        order = [0, 1, 2, 3]
        random.shuffle(order)
        return {"factors": (b, c, s, h), "order": order}

    @staticmethod
    def _brightness(img, f):
        return _lut(img, np.clip(np.arange(256, dtype=np.float32) * f, 0, 255).astype(np.uint8))

    @staticmethod
    def _contrast(img, f):
        mean = float(_gray_cv(img).mean())
        return _lut(img, np.clip(np.arange(256, dtype=np.float32) * f + mean * (1 - f), 0, 255).astype(np.uint8))

    @staticmethod
    def _saturation(img, f):
        g = _gray_cv(img).astype(np.float32)
        g *= (1 - f)
        out = img.astype(np.float32)
        out *= f
        out += g[..., None]
        return np.clip(out, 0, 255, out=out).astype(np.uint8)

    @staticmethod
    def _hue(img, f):
        """hue rotation by f turns through an 8-bit HSV image (PIL's C conversion: H in 256 steps where OpenCV has 180 —
        the numpy conversion of _rgb_to_hsv_u8 reproduces OpenCV's scale but costs 90 ms per 1024x512 view in a worker)"""
        if f == 0:
            return img
        hsv = np.array(Image.fromarray(img).convert("HSV"))
        lut = np.mod(np.arange(256, dtype=np.int32) + int(round(256 * f)), 256).astype(np.uint8)
        hsv[..., 0] = lut[hsv[..., 0]]
        return np.asarray(Image.fromarray(hsv, "HSV").convert("RGB"))

    def apply(self, image, masks, factors, order):
        fns = (self._brightness, self._contrast, self._saturation, self._hue)
        for i in order:
            image = fns[i](image, factors[i])
        return image, masks


# ------------------------------------------------------------------------------------------------- FDA
def fourier_domain_adaptation(img, target_img, beta):
    """albumentations.augmentations.domain_adaptation.fourier_domain_adaptation (arXiv:2004.05498): the low-frequency
    amplitudes (centre square of half-width floor(min(h, w) * beta)) of `img`'s spectrum are replaced by `target_img`'s"""
    img = np.squeeze(img)
    target_img = np.squeeze(target_img)
    if target_img.shape != img.shape:
        target_img = _resize_img(target_img, img.shape[0], img.shape[1])
    fft_src = np.fft.fft2(img.astype(np.float32), axes=(0, 1))
    fft_trg = np.fft.fft2(target_img.astype(np.float32), axes=(0, 1))
    amp_src, pha_src = np.fft.fftshift(np.abs(fft_src), axes=(0, 1)), np.angle(fft_src)
    amp_trg = np.fft.fftshift(np.abs(fft_trg), axes=(0, 1))
    h, w = amp_src.shape[:2]
    border = int(np.floor(min(h, w) * beta))
    cy, cx = int(np.floor(h / 2.0)), int(np.floor(w / 2.0))
    y1, y2, x1, x2 = cy - border, cy + border + 1, cx - border, cx + border + 1
    amp_src[y1:y2, x1:x2] = amp_trg[y1:y2, x1:x2]
    amp_src = np.fft.ifftshift(amp_src, axes=(0, 1))
    out = np.fft.ifft2(amp_src * np.exp(1j * pha_src), axes=(0, 1))
    return np.clip(np.real(out), 0, 255).astype(np.uint8)


class FDA(_Aug):
    """A.FDA(reference_images, beta_limit, read_fn, p): one reference image path drawn with random.choice, beta uniform
    in (0, beta_limit)"""
    image_only = True

    def __init__(self, reference_images, beta_limit=0.1, read_fn=None, p=0.5):
        self.refs, self.beta, self.p = list(reference_images), (0, beta_limit) if np.isscalar(beta_limit) else tuple(beta_limit), p
        self.read_fn = read_fn or (lambda path: np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8))

    def params(self, image):
        # draw order of albumentations 1.0.3: get_params() (beta) runs before get_params_dependent_on_targets() (the
        # reference image)
        beta = random.uniform(self.beta[0], self.beta[1])
        target = self.read_fn(random.choice(self.refs))
        target = _resize_img(target, image.shape[0], image.shape[1])
        return {"target": target, "beta": beta}

    def apply(self, image, masks, target, beta):
        return fourier_domain_adaptation(image, target, beta), masks


# ------------------------------------------------------------------------------------------------- the reference's builders
def resize(h, w):
    return Resize(h, w, p=1.0)


def flip_crop_resize(h, w, min_max_height, w2h_ratio):
    return Compose([HorizontalFlip(p=0.5), RandomSizedCrop(min_max_height, h, w, w2h_ratio)])


def resize_crop(h, w, h_c, w_c):
    return Compose([Resize(h, w, p=1.0), RandomCrop(h_c, w_c, p=1.0)])


def simple_color_aug():
    return Compose([ColorJitter(p=0.5), GaussianBlur(blur_limit=(3, 41), p=0.5)])


def complex_color_aug(selected_num=3):
    pool = [ColorJitter(p=0.5), GaussianBlur(blur_limit=(3, 41), p=0.5), RandomContrast(limit=(0, 3), p=0.5),
            RandomBrightness(limit=0.5, p=0.5), Posterize(num_bits=4, p=0.5), Equalize(p=0.5), Solarize(p=0.5), ToGray(p=0.5)]
    return SomeOf(pool, n=selected_num)


def fda(target_json_path, target_image_dir, beta_limit=0.001):
    with open(target_json_path) as f:
        entries = json.load(f)
    return FDA([os.path.join(target_image_dir, e["image_name"]) for e in entries], beta_limit=beta_limit, p=1.0)


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
