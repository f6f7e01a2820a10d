"""Host-side IAS math (histogram formulation, product code) against numpy's quantile and against
the oracle's list formulation / the reference goldens: bit-exact."""
import numpy as np
import pytest

import synth
from oracle import cref, ias_ref
from hiast_amd.workflows import ias_math


def test_roundup_f32_equivalence():
    g = synth.rng(3)
    t = g.random(20000)
    p = g.random(20000, dtype=np.float32)
    p[:5000] = t[:5000].astype(np.float32)         # force near-equal pairs
    up = ias_math.roundup_f32(t)
    assert np.array_equal(p.astype(np.float64) < t, p < up)


def test_hist_quantile_matches_numpy():
    g = synth.rng(4)
    for trial in range(400):
        n = int(g.integers(0, 400))
        vals = g.random(n, dtype=np.float32).astype(np.float16)
        if trial % 5 == 0 and n:
            vals[: n // 2] = vals[0]                # heavy duplicates
        vals = vals[vals.view(np.uint16) < ias_math.NBINS]
        thr = float(g.random()) if trial % 7 else float(np.float16(0.5))
        q = float(g.random()) if trial % 11 else 1.0
        hist = np.bincount(vals.view(np.uint16), minlength=ias_math.NBINS).astype(np.uint32)
        want = np.quantile(np.concatenate([[thr], vals.astype(np.float64)]), q)
        got = ias_math.hist_quantile(hist, thr, q)
        assert np.float64(got).view(np.uint64) == np.float64(want).view(np.uint64), (trial, got, want)


@pytest.mark.parametrize("bs", [2, 4])
def test_ias_steps_match_reference_golden(golden, bs):
    """product math fed by the oracle's histogram == the reference's thresholds, bit for bit"""
    g = golden("ias_stage_b")
    N, H, W, C = [int(v) for v in g["shape"]]
    tag = "b%d" % bs
    imgs = [synth.probs_and_labels(500 + i, 1, H, W, C) for i in range(N)]
    thr = 0.9 * np.ones(C)
    means = np.zeros(C)
    for k, s in enumerate(range(0, N, bs)):
        p = np.concatenate([imgs[i][0] for i in range(s, s + bs)])
        l = np.concatenate([imgs[i][1] for i in range(s, s + bs)]).astype(np.uint8)
        hist = cref.plabel_hist(p, l, C)
        temp, thr = ias_math.ias_update(hist, thr, 0.5, 0.9, 8.0)
        assert np.array_equal(temp.view(np.uint32), g["temp_" + tag][k].view(np.uint32))
        assert np.array_equal(thr.view(np.uint64), g["thr_" + tag][k].view(np.uint64))
        plbl, count, sfx = cref.plabel_select(p, l, thr, C)
        assert np.array_equal(plbl, g["plbl_" + tag][s:s + bs])
        ias_math.update_class_mean_probs(means, count.sum(0), sfx, 0.99)
        assert np.allclose(means, g["mean_" + tag][k], rtol=1e-6, atol=0)


def test_sharded_histograms_sum_to_unsharded():
    """§8e: summing per-rank histograms == the histogram of the pooled batch, hence identical thresholds"""
    C = 19
    p, l = synth.probs_and_labels(77, 4, 64, 128, C)
    l = l.astype(np.uint8)
    whole = cref.plabel_hist(p, l, C)
    parts = sum(cref.plabel_hist(p[i:i + 1], l[i:i + 1], C).astype(np.int64) for i in range(4))
    assert np.array_equal(whole.astype(np.int64), parts)
    t0 = 0.9 * np.ones(C)
    a = ias_math.ias_update(whole, t0, 0.5, 0.9, 8.0)[1]
    b = ias_math.ias_update(parts, t0, 0.5, 0.9, 8.0)[1]
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_empty_class_and_saturation():
    C = 4
    hist = np.zeros((C, ias_math.NBINS), np.uint32)
    hist[0, 0x3C00] = 1000                       # every pixel at prob 1.0
    temp, thr = ias_math.ias_update(hist, np.array([0.9, 0.9, 0.9995, 0.2]), 0.5, 0.0, 8.0)
    assert temp[0] == 1.0 and thr[0] == 0.999     # >= 1 is clamped (pseudo_label_generator.py:209)
    assert temp[1] == np.float32(0.9)             # empty class: quantile of [thr] is thr
    st = ias_ref.IASState(C, 0.5, 0.0, 8.0)
    st.class_threshold = np.array([0.9, 0.9, 0.9995, 0.2])
    p = np.ones((1, 10, 100), np.float32)
    st.step(p, np.zeros((1, 10, 100), np.int64), ["x"])
    assert np.array_equal(st.class_threshold.view(np.uint64), thr.view(np.uint64))
