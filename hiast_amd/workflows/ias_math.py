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


def cbst_threshold(hist, p):
    """CBST policy (pseudo_label_generator.py:160-163): np.quantile(values_c, 1-p) without the
    seeded element; classes never predicted keep the reference's NaN (np.quantile of [])."""
    C = hist.shape[0]
    out = np.ones(C)
    for c in range(C):
        cum = np.cumsum(hist[c].astype(np.int64))
        n = int(cum[-1])
        if n == 0:
            out[c] = np.nan
            continue
        vi = (n - 1) * (1 - p)
        lo = int(np.floor(vi))
        g = vi - lo
        hi = min(lo + 1, n - 1)
        a = BIN_VALUE[int(np.searchsorted(cum, lo, side="right"))]
        b = BIN_VALUE[int(np.searchsorted(cum, hi, side="right"))]
        d = b - a
        out[c] = (b - d * (1 - g)) if g >= 0.5 else (a + d * g)
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
