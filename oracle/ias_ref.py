"""Numpy restatement of the reference's IAS pseudo-label post-processing.

TEST INFRASTRUCTURE ONLY (parity oracle) — never imported by hiast_amd/.
Parity: PINNED — tests/test_oracle_golden.py checks this file against tests/golden/ias_*.npz,
which hold outputs of the reference's own IASPseudoGenerator methods run in the build
container (tests/golden/make_golden.py).

Follows workflows/pseudo_label_generator.py of bupt-ai-cz/HIAST in the reference's own
formulation (per-class value lists + np.quantile + per-pixel threshold map), i.e. the slow
host path the HIP kernels replace; the product's histogram formulation lives in
hiast_amd/workflows/ias_math.py and is tested against this.
"""
import numpy as np


def ias_threshold(class_values, num_classes, alpha, old_thresholds, gamma):
    """IASPseudoGenerator.get_ias_threshold (pseudo_label_generator.py:171-179).

    class_values[c]: 1-D float64 array = [old_thr_c] + fp16 max-probs of pixels predicted c.
    Returns float32 [C] (the reference stores the float64 quantile into a float32 array)."""
    out = np.ones(num_classes, dtype=np.float32)
    for c in range(num_classes):
        q = 1 - alpha * old_thresholds[c] ** gamma
        out[c] = np.quantile(class_values[c], q)
    return out


def class_value_lists(probs_pred, lbls_pred, class_threshold, num_classes):
    """pseudo_label_generator.py:198-201: list seeded with the current threshold (np.float64),
    extended with the fp16-rounded max-probs of the batch's pixels of that class; np.quantile
    then sees them as one float64 array."""
    p16 = probs_pred.astype(np.float16)
    vals = []
    for c in range(num_classes):
        sel = p16[lbls_pred == c].astype(np.float64)
        vals.append(np.concatenate([np.array([class_threshold[c]], np.float64), sel]))
    return vals


def select_confident(probs_pred, lbls_pred, class_threshold):
    """pseudo_label_generator.py:72-80 for a whole batch: float32 prob < float64 thr[label]."""
    if class_threshold is None:
        return lbls_pred.copy()
    thr_map = np.asarray(class_threshold, np.float64)[lbls_pred]
    plbl = lbls_pred.copy()
    plbl[probs_pred < thr_map] = 255
    return plbl


class IASState:
    """Running state of BasePseudoGenerator / IASPseudoGenerator (pseudo_label_generator.py:16-23,185)."""

    def __init__(self, num_classes, alpha, beta, gamma, cp_gamma=0.99, init_threshold=0.9):
        self.C = num_classes
        self.alpha, self.beta, self.gamma, self.cp_gamma = alpha, beta, gamma, cp_gamma
        self.class_threshold = init_threshold * np.ones(num_classes)
        self.statics_class = np.zeros(num_classes, np.int64)
        self.class_mean_probs = np.zeros(num_classes)
        self.sample_stats = []
        self.samples_class = {i: [] for i in range(num_classes)}
        self.temp_history = []

    def step(self, probs_pred, lbls_pred, img_paths):
        """One loader iteration of IASPseudoGenerator.run (pseudo_label_generator.py:192-211).

        probs_pred float32 [B,H,W], lbls_pred int [B,H,W].  Returns plbl uint8 [B,H,W]."""
        C = self.C
        vals = class_value_lists(probs_pred, lbls_pred, self.class_threshold, C)
        temp = ias_threshold(vals, C, self.alpha, self.class_threshold, self.gamma)
        self.temp_history.append(temp.copy())
        self.class_threshold = self.beta * self.class_threshold + (1 - self.beta) * temp
        self.class_threshold[self.class_threshold >= 1] = 0.999
        return self.select_and_record(probs_pred, lbls_pred, img_paths)

    def select_and_record(self, probs_pred, lbls_pred, img_paths):
        """select_and_save_confident_label (pseudo_label_generator.py:67-105) minus the PNG write."""
        C = self.C
        plbl = select_confident(probs_pred, lbls_pred, self.class_threshold)
        for b, path in enumerate(img_paths):
            stats = {}
            for i in range(C):
                n = int(np.count_nonzero(plbl[b] == i))
                if n != 0:
                    stats[i] = n
                    self.samples_class[i].append([path, n])
                    self.statics_class[i] += n
            stats['file'] = path
            self.sample_stats.append(stats)
        for c in range(C):
            sel = probs_pred[plbl == c]
            if sel.size == 0:
                continue                      # np.mean([]) = nan -> skipped by the reference
            mean_value = np.mean(sel)
            if np.isnan(mean_value) or np.isinf(mean_value):
                continue
            if self.class_mean_probs[c] == 0:
                self.class_mean_probs[c] = mean_value
            else:
                self.class_mean_probs[c] = self.class_mean_probs[c] * self.cp_gamma + \
                    mean_value * (1 - self.cp_gamma)
        return plbl.astype(np.uint8)


def cbst_lists(batches, num_classes, sample_interval):
    """pseudo_label_generator.py:146-158: per class the fp16 max-probs of every `sample_interval`-th pixel of that
    class IN EACH BATCH (raster order over the batch), pooled over the whole target set.
    batches: iterable of (probs_pred f32 [B,H,W], lbls_pred)."""
    lists = {c: [] for c in range(num_classes)}
    for probs_pred, lbls_pred in batches:
        for c in range(num_classes):
            tmp = probs_pred[lbls_pred == c].astype(np.float16)
            lists[c].extend(tmp[0:len(tmp):sample_interval])
    return lists


def cbst_threshold(batches, num_classes, p, sample_interval, as_float64=False):
    """CBSTPseudoGenerator.get_constant_threshold (pseudo_label_generator.py:142-165): np.quantile(list_c, 1 - p).
    The lists hold np.float16 scalars: numpy >= 2 (this image: what the fixtures record) evaluates the quantile in
    float16, numpy 1.19.2 (the reference's pin) in float64 — `as_float64` converts the sample first."""
    lists = cbst_lists(batches, num_classes, sample_interval)
    thr = np.ones(num_classes)
    for c in range(num_classes):
        if not len(lists[c]):
            thr[c] = np.nan                    # numpy 1.19: np.quantile([]) = nan; numpy >= 1.22 raises
            continue
        thr[c] = np.quantile(np.asarray(lists[c], np.float64) if as_float64 else lists[c], 1 - p)
    return thr


class ConstantPolicyState(IASState):
    """'CT' / 'NT' / 'CBST' (pseudo_label_generator.py:109-140): one threshold vector (or None) for the whole set."""

    def __init__(self, num_classes, class_threshold, cp_gamma=0.99):
        super().__init__(num_classes, 0.0, 0.0, 0.0, cp_gamma)
        self.class_threshold = None if class_threshold is None else np.asarray(class_threshold, np.float64)

    def step(self, probs_pred, lbls_pred, img_paths):
        return self.select_and_record(probs_pred, lbls_pred, img_paths)
