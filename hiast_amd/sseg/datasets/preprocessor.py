"""PREPROCESSOR['CopyPaste'] — HIAST's hard-aware pseudo-label augmentation
(reference: sseg/datasets/preprocessor.py:12-122).

Hard classes = the `selected_num_classes` classes with the lowest mean confidence (from
class_mean_probabilities.npy written by the generator); a class is drawn with probability
∝ (1 - value)^2, one pseudo-labelled image containing it is drawn, and every hard-class pixel of that
image is pasted (image bytes + label) onto the current sample.  Runs on the host inside DataLoader
workers on uint8 arrays (byte copies; nothing for a GPU to win).  The np.random call sequence is the
reference's, so a seeded run reproduces its outputs (tests/golden/copy_paste.npz).

Deviation (documented in DESIGN.md): for SYNTHIA the reference sets the three absent classes'
value to inf, which turns the sampling probabilities into NaN and makes np.random.choice raise
(preprocessor.py:18-19,31-40); here those classes get probability 0 instead."""
import numpy as np
from PIL import Image

from hiast_amd.utils.registry.registries import PREPROCESSOR


@PREPROCESSOR.register("CopyPaste")
class CopyPaste:

    def __init__(self, cfg, dataset_copy_from, init_class_value):
        self.cfg = cfg
        self.dataset_copy_from = dataset_copy_from
        self.ignored_classes = [9, 14, 16] if cfg.dataset.source.type == "SYNTHIA" else None
        print("%% Init copy paste with class_value: {}".format(init_class_value))
        self.class_value, self.hard_classes = self.get_hard_classes(np.array(init_class_value, dtype=np.float64))
        self.samples_with_class = self.dataset_copy_from.get_samples_with_class()
        self.class_probs = self.calculate_class_probs()

    def calculate_class_probs(self):
        v = np.asarray(self.class_value, np.float64)
        p = np.where(np.isinf(v), 0.0, (1 - np.where(np.isinf(v), 1.0, v)) ** 2)
        if p.sum() <= 0:                   # every class at confidence 1.0: fall back to uniform over finite classes
            p = np.where(np.isinf(v), 0.0, 1.0)
        return p / p.sum()

    def get_hard_classes(self, class_value):
        if self.ignored_classes is not None:
            for c in self.ignored_classes:
                class_value[c] = np.inf
        hard = np.argsort(class_value)[:self.cfg.preprocessor.copy_paste.selected_num_classes]
        return class_value, hard

    @staticmethod
    def resize(img, lbl, shape):
        h, w = shape[0], shape[1]
        img = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        lbl = np.asarray(Image.fromarray(lbl).resize((w, h), Image.NEAREST))
        return img, lbl

    def run(self, img, lbl):
        if self.cfg.preprocessor.copy_paste.mode != "original":
            raise NotImplementedError("copy_paste.mode %r" % self.cfg.preprocessor.copy_paste.mode)
        return self.run_original(img, lbl)

    def random_select(self, selected_classes):
        """rejection-sample a hard class (preprocessor.py:70-77).  Deviations: a class that no pseudo-labelled
        image contains is rejected too (the reference would raise in np.random.choice on its empty file list), and
        after 64 rejections the class is drawn directly from the distribution renormalised over the usable classes
        (the reference's loop is unbounded: with nearly all probability mass on unusable classes it spins for
        ~1/P(usable) draws).  With every hard class populated the draw sequence is identical to the reference's."""
        ids = [i for i in range(self.cfg.dataset.num_classes)]
        usable = [c for c in selected_classes if len(self.samples_with_class.get(int(c), [])) > 0]
        if not usable:
            return None
        p_usable = self.class_probs[np.asarray(usable, dtype=int)]
        if float(np.sum(p_usable)) <= 0.0:
            return np.random.choice(usable)     # degenerate: all usable classes have sampling probability 0
        for _ in range(64):
            c = np.random.choice(ids, size=1, replace=False, p=self.class_probs)[0]
            if c in usable:
                return c
        return np.random.choice(usable, p=p_usable / p_usable.sum())

    def run_original(self, img, lbl):
        """Returns (img, lbl, copy_paste_mask).  The reference's retry loop (up to 3 source images)
        always ends after the first paste, because the first pass marks every hard class as
        present (preprocessor.py:104-118); one paste is therefore the whole behaviour."""
        mask = np.full(lbl.shape, 255, dtype=np.uint8)
        c = self.random_select(self.hard_classes)
        if c is None:                      # nothing to paste from
            return img, lbl, mask
        name = np.random.choice(self.samples_with_class[c])
        img_, lbl_, _ = self.dataset_copy_from.load_data(self.dataset_copy_from.get_file_to_idx(name))
        if img.shape != img_.shape:
            img_, lbl_ = self.resize(img_, lbl_, lbl.shape)
        lut = np.zeros(256, dtype=bool)          # one table lookup instead of a compare per hard class
        lut[np.asarray(self.hard_classes, dtype=np.int64)] = True
        sel = lut[lbl_]
        rows = np.flatnonzero(sel.any(axis=1))
        if rows.size:            # img[mask] = img_[mask] as masked copies restricted to the bounding box of the pasted
            r0, r1 = int(rows[0]), int(rows[-1]) + 1         # pixels (labels: np.copyto(where=), a vectorised select; boolean
            cols = np.flatnonzero(sel[r0:r1].any(axis=0))    # fancy indexing measured 3x slower on dense masks)
            c0, c1 = int(cols[0]), int(cols[-1]) + 1
            sub = sel[r0:r1, c0:c1]
            np.copyto(mask[r0:r1, c0:c1], lbl_[r0:r1, c0:c1], where=sub)
            # RGB: byte-wise select with a mask expanded to the three channels, (a & ~m) | (b & m) — three SIMD passes
            # over contiguous bytes instead of numpy's broadcast-mask copy (12 vs 35 ms on a 2048x1024 frame)
            m3 = np.repeat(sub.view(np.uint8) * np.uint8(255), 3, axis=1).reshape(r1 - r0, c1 - c0, 3)
            dst = img[r0:r1, c0:c1]
            keep = np.bitwise_and(dst, ~m3)
            np.bitwise_and(img_[r0:r1, c0:c1], m3, out=m3)
            np.bitwise_or(keep, m3, out=keep)
            dst[...] = keep
            np.copyto(lbl[r0:r1, c0:c1], lbl_[r0:r1, c0:c1], where=sub)
        return img, lbl, mask
