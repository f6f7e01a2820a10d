"""Host-side scalar math of the instance-adaptive selector (IAS), driven by the integer
histogram that hiast_plabel_pass1 produces on the GPU.

Reference: IASPseudoGenerator.run / get_ias_threshold, workflows/pseudo_label_generator.py:171-179,
198-209: per class, `np.quantile([thr_c] + fp16(max-probs of the batch), 1 - alpha*thr_c**gamma)`
stored as float32, then an EMA in float64.  The multiset of fp16 values IS the histogram over fp16
bit patterns, so the quantile (numpy's 'linear' method: virtual index (n-1)q, lerp a+(b-a)g, or
b-(b-a)(1-g) for g >= 0.5) is reproduced bit-exactly from 19 x 15361 integers; this is ~1e5 integer
ops per batch on the host instead of O(pixels) Python list work, and it is what makes the sharded
(all-reduced histogram) generator identical to the single-process one.
"""
import numpy as np

NBINS = 15361
# value of fp16 bit pattern b, as float64 (monotone in b for b <= 0x3C00)
BIN_VALUE = np.arange(NBINS, dtype=np.uint16).view(np.float16).astype(np.float64)


def roundup_f32(t64):
    """smallest float32 >= t (so that `p32 < t64`  <=>  `p32 < roundup_f32(t64)`)"""
    t64 = np.asarray(t64, np.float64)
    t32 = t64.astype(np.float32)
    low = t32.astype(np.float64) < t64
    return np.where(low, np.nextafter(t32, np.float32(np.inf)), t32).astype(np.float32)


def _kth(cum, thr, k, r):
    """k-th smallest (0-based) of {thr} U histogram multiset; cum = inclusive cumsum of the
    histogram, r = number of histogram values strictly below thr."""
    if k == r:
        return thr
    kk = k if k < r else k - 1
    b = int(np.searchsorted(cum, kk, side="right"))
    return BIN_VALUE[b]


def hist_quantile(hist_c, thr, q):
    """np.quantile(np.array([thr] + values), q) where `values` has hist_c[b] copies of BIN_VALUE[b]."""
    cum = np.cumsum(hist_c.astype(np.int64))
    n = int(cum[-1]) + 1
    r = int(cum[np.searchsorted(BIN_VALUE, thr, side="left") - 1]) if thr > BIN_VALUE[0] else 0
    vi = (n - 1) * q
    lo = np.floor(vi)
    g = vi - lo
    lo = int(lo)
    hi = lo + 1
    if vi >= n - 1:
        lo = hi = n - 1
    if vi < 0:
        lo = hi = 0
    a = _kth(cum, thr, lo, r)
    b = _kth(cum, thr, hi, r) if hi != lo else a
    d = b - a
    return (b - d * (1 - g)) if g >= 0.5 else (a + d * g)


def ias_threshold(hist, thr_prev, alpha, gamma):
    """get_ias_threshold on histograms -> float32 [C].  Same arithmetic as hist_quantile per class, with the
    cumulative sums of all classes taken in one pass (this runs between the two GPU passes of every batch)."""
    C = hist.shape[0]
    # max-probs are >= 1/C, so the low bins are empty: start the cumulative sums at the first used bin
    used = hist.any(axis=0)
    first = int(np.argmax(used)) if used.any() else hist.shape[1] - 1
    cum = np.cumsum(hist[:, first:], axis=1, dtype=np.int64)
    bins = BIN_VALUE[first:]
    thr_prev = np.asarray(thr_prev, np.float64)
    out = np.ones(C, dtype=np.float32)
    idx = np.searchsorted(bins, thr_prev, side="left")
    for c in range(C):
        cc = cum[c]
        n = int(cc[-1]) + 1
        thr = thr_prev[c]
        r = int(cc[idx[c] - 1]) if idx[c] > 0 else 0
        q = 1 - alpha * thr ** gamma
        vi = (n - 1) * q
        lo = np.floor(vi)
        g = vi - lo
        lo = int(lo)
        hi = lo + 1
        if vi >= n - 1:
            lo = hi = n - 1
        if vi < 0:
            lo = hi = 0
        a = thr if lo == r else bins[cc.searchsorted(lo if lo < r else lo - 1, side="right")]
        b = a if hi == lo else (thr if hi == r else bins[cc.searchsorted(hi if hi < r else hi - 1, side="right")])
        d = b - a
        out[c] = (b - d * (1 - g)) if g >= 0.5 else (a + d * g)
    return out


def ias_update(hist, thr_prev, alpha, beta, gamma):
    """one IAS step: (temp float32 [C], new class_threshold float64 [C])
    (pseudo_label_generator.py:204-209)"""
    hist = np.asarray(hist)
    assert hist.ndim == 2 and hist.shape[1] == NBINS
    thr_prev = np.asarray(thr_prev, np.float64)
    temp = ias_threshold(hist, thr_prev, alpha, gamma)
    thr = beta * thr_prev + (1 - beta) * temp
    thr[thr >= 1] = 0.999
    return temp, thr


def cbst_threshold(hist, p, arithmetic=None):
    """CBST policy (pseudo_label_generator.py:160-163): np.quantile(values_c, 1 - p) of the pooled fp16 confidences
    (no seeded element); classes never predicted get NaN (numpy 1.19's np.quantile([]); numpy >= 1.22 raises there).

    `values_c` is a list of np.float16 scalars, and what np.quantile does with it depends on the numpy version:
      * arithmetic="float64" (default): index (n-1)q, gamma and the lerp in float64 — numpy 1.19.2, the reference's
        pin, converts a float16 sample to float64 on the way; the lerp form a+(b-a)g / b-(b-a)(1-g) is numpy 2.2's,
        like the IAS quantile above;
      * arithmetic="float16": numpy >= 2.0 keeps the SAMPLE's dtype — q, the virtual index (n-1)q and the lerp are
        all rounded to float16 (the index of a sample of more than 2048 values is no longer exact, and beyond
        65504/q it overflows to inf: both neighbours become the class maximum and the lerp weight inf, so the result
        is 0 * inf = NaN — numpy 2.2 returns exactly that, checked on 200k samples; a warning names such classes).
        This is what the reference computes under the numpy of this image, reproduced operation by operation;
        tests/golden/policies.npz holds such values.  Only usable for small pools: real target sets need "float64".
    HIAST_CBST_QUANTILE=float16|float64 selects it when `arithmetic` is None."""
    import os
    if arithmetic is None:
        arithmetic = os.environ.get("HIAST_CBST_QUANTILE", "float64")
    if arithmetic not in ("float64", "float16"):
        raise ValueError("cbst_threshold: arithmetic must be 'float64' or 'float16'")
    C = hist.shape[0]
    out = np.ones(C)
    for c in range(C):
        cum = np.cumsum(hist[c].astype(np.int64))
        n = int(cum[-1])
        if n == 0:
            out[c] = np.nan
            continue
        kth = lambda k: BIN_VALUE[int(np.searchsorted(cum, k % n, side="right"))]      # k = -1: the maximum
        if arithmetic == "float64":
            vi = (n - 1) * (1 - p)
            lo = int(np.floor(vi))
            g = vi - lo
            hi = min(lo + 1, n - 1)
            a, b = kth(lo), kth(hi)
            d = b - a
            out[c] = (b - d * (1 - g)) if g >= 0.5 else (a + d * g)
            continue
        # numpy 2.x on a float16 sample (numpy/lib/_function_base_impl.py: quantile, _quantile, _get_indexes, _lerp)
        q16 = np.asanyarray(1 - p, dtype=np.float16)
        with np.errstate(over="ignore", invalid="ignore"):
            vi = np.asanyarray((n - 1) * q16)                  # float16, rounded; inf beyond 65504
            prev = np.floor(vi)
            nxt = prev + 1
            if vi >= n - 1 or np.isnan(vi):
                lo = hi = -1
            elif vi < 0:
                lo = hi = 0
            else:
                lo, hi = int(prev), int(nxt)
            gamma = np.asanyarray(np.asanyarray(vi - np.intp(lo)), dtype=np.float16)
            a, b = np.float16(kth(lo)), np.float16(kth(hi))
            diff = np.subtract(b, a)
            r = np.add(a, diff * gamma)
            if gamma >= 0.5:
                r = np.subtract(b, diff * (1 - gamma)).astype(np.float16)
        out[c] = float(r)
        if np.isnan(out[c]):
            import warnings
            warnings.warn("cbst_threshold(arithmetic='float16'): class %d has %d pooled confidences, its float16 virtual "
                          "index overflows and numpy >= 2.0 yields NaN (no pixel of the class will be kept); use "
                          "arithmetic='float64' (numpy 1.19.2, the reference's pin)" % (c, n))
    return out


def update_class_mean_probs(class_mean_probs, count_c, sumprob_fx, cp_gamma):
    """pseudo_label_generator.py:96-105 from exact integer sums: mean = Σprob / count per class
    (Σprob accumulated as prob*2^30 integers on the GPU), EMA with copy_paste.gamma, classes
    without confident pixels skipped.  In place; returns the array."""
    for c in range(len(class_mean_probs)):
        n = int(count_c[c])
        if n == 0:
            continue
        mean_value = (float(int(sumprob_fx[c])) / float(1 << 30)) / n
        if class_mean_probs[c] == 0:
            class_mean_probs[c] = mean_value
        else:
            class_mean_probs[c] = class_mean_probs[c] * cp_gamma + mean_value * (1 - cp_gamma)
    return class_mean_probs
