"""Functional torch-CPU fp32 restatement of the reference DeepLab_V2 forward (floating-point oracle).

TEST INFRASTRUCTURE ONLY.  Parity: PINNED by tests/golden/deeplab.npz (outputs of the reference's
own DeepLab_V2 class on seeded weights).

Follows sseg/models/modules/resnet.py:78-98 (Bottleneck), :176-190 (ResNet.forward) and
sseg/models/modules/seg_models/deeplab_v2.py:20-24,34-35,42-56,58-64 (output-stride-8 dilation
surgery, ASPP sum, dead `representation` branch), driven directly by a state dict with the
reference's key names.
"""
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)


def _bn(x, sd, pre, train, eps=1e-5, stats_out=None):
    if train:   # frozen-affine BN still normalises with batch statistics in model.train()
        if stats_out is not None:      # ... and moves its running statistics (momentum 0.1, unbiased variance)
            rm, rv = sd[pre + ".running_mean"].detach().clone(), sd[pre + ".running_var"].detach().clone()
            y = F.batch_norm(x, rm, rv, sd[pre + ".weight"], sd[pre + ".bias"], True, 0.1, eps)
            stats_out[pre + ".running_mean"], stats_out[pre + ".running_var"] = rm, rv
            return y
        return F.batch_norm(x, None, None, sd[pre + ".weight"], sd[pre + ".bias"], True, 0.1, eps)
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], sd[pre + ".weight"],
                        sd[pre + ".bias"], False, 0.1, eps)


def conv_geometry(layer, block):
    """(stride, dilation) of conv2 and stride of the downsample conv after _nostride_dilate
    (deeplab_v2.py:42-56): layer3/4 lose their stride; block 0 keeps half the dilation."""
    if layer == 1:
        return 1, 1
    if layer == 2:
        return (2 if block == 0 else 1), 1
    full = 2 if layer == 3 else 4
    return 1, (full // 2 if block == 0 else full)


def _pool_by_codes(x, codes):
    """MaxPool2d(3, 2, 1) with the window position of every output GIVEN (codes [B,C,Ho,Wo] in 0..8 = row * 3 + column of the
    window, row-major as in ATen): a gather — the linearisation of the pooling around a fixed selection"""
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    patches = F.unfold(x, 3, padding=1, stride=2).view(B, C, 9, Ho * Wo)
    return patches.gather(2, codes.reshape(B, C, 1, Ho * Wo).long()).view(B, C, Ho, Wo)


def backbone(x, sd, pre="backbone.", train=False, stats_out=None, relu_masks=None, pool_codes=None):
    """stats_out: dict that receives the running statistics every BatchNorm holds AFTER this train-mode forward.
    relu_masks / pool_codes (analysis mode of tests/test_gpu_trainstep_oracle.py, not a reference behaviour): every ReLU is
    replaced by a multiplication with the given 0/1 mask (keys: the name of the BatchNorm in front of it) and the stem's
    pooling by a gather at the given window positions — the network linearised around a GIVEN set of gates, so that an
    implementation whose gates are exported can be compared without the 100 % per-element effect of a flipped gate."""
    g_bn = globals()["_bn"]

    def _bn(x, sd, name, train):
        return g_bn(x, sd, name, train, stats_out=stats_out)

    def relu(z, name):
        if relu_masks is None:
            return F.relu(z)
        return z * relu_masks[name].to(z.dtype)
    x = F.conv2d(x, sd[pre + "conv1.weight"], None, 2, 3)
    x = relu(_bn(x, sd, pre + "bn1", train), pre + "bn1")
    x = F.max_pool2d(x, 3, 2, 1) if pool_codes is None else _pool_by_codes(x, pool_codes)
    for li, nblocks in enumerate(LAYERS, start=1):
        for b in range(nblocks):
            p = "%slayer%d.%d." % (pre, li, b)
            stride, dil = conv_geometry(li, b)
            idt = x
            o = relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1", train), p + "bn1")
            o = F.conv2d(o, sd[p + "conv2.weight"], None, stride, dil, dil)
            o = relu(_bn(o, sd, p + "bn2", train), p + "bn2")
            o = _bn(F.conv2d(o, sd[p + "conv3.weight"]), sd, p + "bn3", train)
            if (p + "downsample.0.weight") in sd:
                idt = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride), sd,
                          p + "downsample.1", train)
            x = relu(o + idt, p + "bn3")
    return x


def aspp(feat, sd, pre="aspp.conv2d_list.", dil=(6, 12, 18, 24)):
    out = None
    for i, d in enumerate(dil):
        y = F.conv2d(feat, sd["%s%d.weight" % (pre, i)], sd["%s%d.bias" % (pre, i)], 1, d, d)
        out = y if out is None else out + y
    return out


def deeplab_v2(x, sd, train=False, stats_out=None, relu_masks=None, pool_codes=None):
    """-> (prediction [B,C,H/8,W/8], feature [B,2048,H/8,W/8]); `representation` is computed and
    dropped by the reference (deeplab_v2.py:63), so it is skipped here."""
    feat = backbone(x, sd, train=train, stats_out=stats_out, relu_masks=relu_masks, pool_codes=pool_codes)
    return aspp(feat, sd), feat


def segmentor_logits(x, sd, train=False, prefix="seg_model."):
    """SelfTrainingSegmentor.forward (self_training_segmentor.py:25-28)."""
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    pred, feat = deeplab_v2(x, sub, train)
    return F.interpolate(pred, size=x.shape[2:], mode="bilinear", align_corners=True), pred, feat
