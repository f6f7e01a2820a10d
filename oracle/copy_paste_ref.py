"""Numpy restatement of the hard-aware CopyPaste augmentation.  TEST INFRASTRUCTURE ONLY.
Parity: PINNED by tests/golden/copy_paste.npz (outputs of the reference class under np.random.seed(888)).

Follows sseg/datasets/preprocessor.py: hard classes = the `selected_num_classes` smallest class
values (:36-44), sampling probabilities (1-v)^2 normalised, computed through torch in the
reference (float64 tensor, :29-34), class draw by rejection (:70-77), one source file drawn
uniformly from samples_with_class[c] (:95), union mask of ALL hard classes present in that file
(:103-108), byte copy (:111-112); the loop always stops after the first paste because every hard
class was marked on the first pass (:104-106,116-118).
"""
import numpy as np


def hard_classes(class_value, k):
    return np.argsort(class_value)[:k]


def class_probs(class_value):
    p = (1 - np.asarray(class_value, np.float64)) ** 2
    return p / p.sum()


def run(img, lbl, hard, probs, samples_with_class, load_by_name, num_classes):
    mask = np.full(lbl.shape, 255, np.uint8)
    hard_set = set(int(c) for c in hard)
    while True:
        c = np.random.choice([i for i in range(num_classes)], size=1, replace=False, p=probs)[0]
        if int(c) in hard_set:
            break
    name = np.random.choice(samples_with_class[int(c)])
    img_, lbl_ = load_by_name(name)
    sel = np.zeros(lbl.shape, bool)
    for h in hard:
        sel[lbl_ == h] = True
        mask[lbl_ == h] = h
    img[sel] = img_[sel]
    lbl[sel] = lbl_[sel]
    return img, lbl, mask
