"""ResNet-101 trunk without avgpool/fc, parameter names identical to the reference
(sseg/models/modules/resnet.py:58-98,101-190 — torchvision naming: conv1/bn1/layer{1..4}.{i}.
conv{1,2,3}/bn{1,2,3}/downsample.{0,1}) so released checkpoints load key-for-key.
Every convolution of the trunk — inference (pseudo-label pass on fp32-class split planes, teacher forward in 16 bits: the
stem is hiast_stem_eval) and the mixed-precision training forward / data gradient / weight gradient (the stem:
hiast_stem_train_fwd / hiast_stem_wgrad; the bottlenecks: hiast_igemm_bn_act incl. the transposed form of the strided
3x3, hiast_conv_wgrad_group_nhwc / hiast_conv_wgrad_nhwc / hiast_conv_wgrad_small_nhwc) — runs on the hand-written
kernels between the fused BatchNorm kernels (hiast_bn_nhwc_*).  PyTorch-ROCm (MIOpen) carries the trunk only in fp32
(apex_opt O0) training and on CPU tensors (BASELINE configs[0])."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from hiast_amd import functional as HF


def bn_act(bn, x, res=None, relu=True, partial=None):
    """relu?(bn(x) [+ res]) — one fused HIP op on the device (hiast_bn_*), plain torch on CPU tensors.
    partial: per-block (Σx, Σx²) from the producing kernel's epilogue (device only)"""
    if x.is_cuda:
        return HF.bn_act(x, bn, res, relu, partial=partial)
    y = bn(x)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


def conv(m, x):
    """a trunk nn.Conv2d: on channels-last bf16 device activations (the mixed-precision training step) it runs on
    the hand-written implicit-GEMM kernel, otherwise as the plain torch module"""
    if x.is_cuda and x.dtype in HF.H16 and HF.conv_nhwc_ok(x, m):
        return HF.conv_nhwc(x, m)
    return m(x)


def conv_bn_act(cv, bn, x, res=None, relu=True, conv_box=None, bn_box=None, in_bn=None, stat_box=None, wgroup=None, xsum=None,
                in_bn3=None):
    """relu?(bn(conv(x)) [+ res]).  On the channels-last bf16 training path the convolution's epilogue also
    delivers the per-block sums the BatchNorm needs (one pass over the activation less); stat_box / in_bn: the same for
    the BACKWARD sums — this BatchNorm registers itself in stat_box, the next conv_bn_act gets that dict as in_bn."""
    if x.is_cuda and x.dtype in HF.H16 and bn.training and HF.conv_nhwc_ok(x, cv):
        y, partial = HF.conv_nhwc(x, cv, want_stats=True, box=conv_box, in_bn=in_bn, wgroup=wgroup, xsum=xsum, in_bn3=in_bn3)
        return HF.bn_act(y, bn, res, relu, partial=partial, box=bn_box, stat_box=stat_box)
    return bn_act(bn, conv(cv, x), res, relu)


def _nhwc2d(t):
    """logical NCHW tensor with channels-last strides -> zero-copy [B*H*W, C] view"""
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C)


def _from2d(y2d, B, H, W):
    """[B*H*W, C] -> logical NCHW tensor with channels-last strides (zero-copy)"""
    return y2d.view(B, H, W, y2d.shape[1]).permute(0, 3, 1, 2)


def packed_weight(conv, fmt):
    """the conv's weight in the LDS-DMA kernels' operand format `fmt` (K.FMT_*; hiast_pack_conv_weight), cached on the
    module until the parameter is modified (its version counter or storage changes)"""
    from hiast_amd import kernels as K
    w = conv.weight
    cache = conv.__dict__.setdefault("_hiast_packed", {})
    ent = cache.get(fmt)
    if ent is not None and ent[0] == w._version and ent[1] == w.data_ptr():
        return ent[2]
    wp = K.pack_conv_weight(w, fmt)
    cache[fmt] = (w._version, w.data_ptr(), wp)
    return wp


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # stage-entry block with a stride-1 downsample: conv1 and the downsample convolution read the same x; their two data
        # gradients become one (HF._ConvNhwcFn xsum: conv1 hands its gradient to the downsample's data-gradient epilogue)
        xs = None
        if (self.downsample is not None and self.downsample[0].stride == (1, 1) and x.is_cuda and x.dtype in HF.H16
                and self.bn1.training and x.requires_grad and not HF.SW.on("HIAST_NO_XSUM")):
            xs = {}
        idt = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False,
                                                            xsum=None if xs is None else (xs, "take"))
        # identity blocks on the channels-last training path: conv1's data-gradient epilogue adds the ReLU-masked
        # gradient of the identity branch itself (no masked copy written by bn3's backward, no separate add kernel);
        # `box` is the hand-off between the two autograd nodes of this call
        box = {} if (self.downsample is None and not HF.SW.on("HIAST_NO_IDT_HANDOFF")) else None
        sb1, sb2 = {}, {}       # bn1 -> conv2's data gradient, bn2 -> conv3's: backward statistics from the dgrad epilogue
        # layer3 / layer4 on the 16-bit training path: the three weight gradients of the block are ONE grouped launch at the
        # end of its backward (HF._WGroupFn) instead of three launches that each fill the chip with partial tiles
        grp, wv = (None, None)
        if x.is_cuda and x.dtype in HF.H16 and self.bn1.training:
            grp, wv = HF.wgroup_weights((self.conv1, self.conv2, self.conv3), x)
        g = (lambda i: (grp, grp["slot"][i], wv[i]) if wv[i] is not None else None) if grp is not None else (lambda i: None)
        # x may carry the registration of the previous block's bn3 (below): conv1's data gradient of an identity block then also
        # delivers that BatchNorm's backward sums (HF._ConvNhwcFn in_bn3)
        o = conv_bn_act(self.conv1, self.bn1, x, conv_box=box, stat_box=sb1, wgroup=g(0),
                        xsum=None if xs is None else (xs, "give"), in_bn3=getattr(x, "_hiast_bn3", None) if box is not None else None)
        o = conv_bn_act(self.conv2, self.bn2, o, in_bn=sb1, stat_box=sb2, wgroup=g(1))
        sb3 = {}
        y = conv_bn_act(self.conv3, self.bn3, o, res=idt, bn_box=box, in_bn=sb2, stat_box=sb3, wgroup=g(2))   # += identity, ReLU
        if "bn3" in sb3:
            y._hiast_bn3 = sb3         # read by the next block (the one consumer of y besides its identity path)
        return y

    def forward_eval_planes(self, x, fmt):
        """inference on channels-last 16-bit activations [B,H,W,planes*C] in operand format `fmt` (K.FMT_SPLIT_BF16: split
        planes, fp32-class — the pseudo-label forward; K.FMT_BF16 / K.FMT_FP16: the teacher forward under autocast): every
        conv runs as one LDS-DMA implicit-GEMM kernel fused with BN(eval) + (residual) + ReLU (hiast_igemm_bn_act)."""
        from hiast_amd import kernels as K
        PL = 2 if fmt == K.FMT_SPLIT_BF16 else 1
        o = K.igemm_bn_act(x, packed_weight(self.conv1, fmt), PL, self.bn1, None, True)
        if self.downsample is None:
            idt = x
        else:
            dconv, dbn = self.downsample[0], self.downsample[1]
            xs = x if dconv.stride == (1, 1) else x[:, ::dconv.stride[0], ::dconv.stride[1]].contiguous()
            idt = K.igemm_bn_act(xs, packed_weight(dconv, fmt), PL, dbn, None, False)
        o = K.igemm_bn_act(o, packed_weight(self.conv2, fmt), PL, self.bn2, None, True, self.conv2.stride[0],
                           self.conv2.dilation[0])
        return K.igemm_bn_act(o, packed_weight(self.conv3, fmt), PL, self.bn3, idt, True)


class ResNet(nn.Module):
    """`strides`/`dilations` give, per stage, the stride of its first block and the dilation of
    (first block, remaining blocks).  The default is the plain torchvision geometry; DeepLab_V2
    builds the output-stride-8 variant directly instead of patching conv attributes afterwards
    (deeplab_v2.py:34-35,42-56 of the reference does the latter; the resulting convs are the same)."""

    def __init__(self, layers=(3, 4, 23, 3), strides=(1, 2, 2, 2), dilations=((1, 1), (1, 1), (1, 1), (1, 1))):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            setattr(self, "layer%d" % (i + 1), self._stage(planes, n, strides[i], dilations[i]))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")   # resnet.py:135-137
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, planes, n, stride, dil):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        blocks = [Bottleneck(self.inplanes, planes, stride, dil[0], down)]
        self.inplanes = planes * 4
        blocks += [Bottleneck(self.inplanes, planes, 1, dil[1]) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def prepack(self, PL, adjoint=False):
        """bring the packed (kernel-format) copies of every bottleneck convolution weight up to date with ONE launch
        (hiast_pack_conv_weight_multi) — they go stale whenever the optimiser or the EMA update has run — and publish
        them in the per-module caches packed_weight() / conv_nhwc() look at.  PL: the operand format (K.FMT_BF16 = 1,
        K.FMT_SPLIT_BF16 = 2, K.FMT_FP16 = 3); adjoint: also the data-gradient forms."""
        from hiast_amd import kernels as K
        plans = self.__dict__.setdefault("_hiast_plans", {})
        ent = plans.get((PL, adjoint))
        if ent is None or not ent[0].still_valid():
            convs = []
            for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in stage:
                    convs += [blk.conv1, blk.conv2, blk.conv3] + ([blk.downsample[0]] if blk.downsample is not None else [])
            convs = [c for c in convs if c.in_channels % 64 == 0 and c.out_channels % 64 == 0 and c.weight.is_cuda]
            if not convs:
                return
            plan = K.PackPlan([c.weight for c in convs], PL,
                              [adjoint and (c.stride == (1, 1) or c.kernel_size == (1, 1)
                                            or (c.kernel_size == (3, 3) and c.stride == (2, 2) and c.dilation == (1, 1)))
                               for c in convs])
            ent = (plan, convs)
            plans[(PL, adjoint)] = ent
        plan, convs = ent
        if plan.refresh() or not getattr(plan, "published", False):
            for i, c in enumerate(convs):
                w = c.weight
                c.__dict__.setdefault("_hiast_packed", {})[PL] = (w._version, w.data_ptr(), plan.wp[i])
                if plan.wpt[i] is not None:
                    c.__dict__.setdefault("_hiast_packed_adj", {})[PL] = (w._version, w.data_ptr(), plan.wpt[i])
            plan.published = True

    def fast_eval_planes(self, x):
        """-> operand format (K.FMT_*) or 0: inference without autograd on the device runs on the 16-bit channels-last
        kernels: split planes (2, fp32-class) for an fp32 forward — the pseudo-label pass —, plain bf16 (1) / fp16 (3)
        rows under bf16 / fp16 autocast — the teacher forward of the mixed-precision step; 0 = use the module path"""
        if self.training or not x.is_cuda or torch.is_grad_enabled() or x.dtype != torch.float32:
            return 0
        if os.environ.get("HIAST_NO_FAST_EVAL", "0") == "1":
            return 0
        if not torch.is_autocast_enabled():
            return 2
        amp = torch.get_autocast_dtype("cuda")
        return 1 if amp == torch.bfloat16 else (3 if amp == torch.float16 else 0)

    def forward_eval_planes(self, x, PL):
        """PL: operand format (K.FMT_*) -> trunk feature as a 16-bit channels-last tensor [B,h,w,planes*2048]"""
        from hiast_amd import kernels as K
        self.prepack(PL)
        if K.stem_eval_supported(self.conv1, self.bn1, self.maxpool) and x.dtype == torch.float32 and x.shape[1] == 3:
            # conv1 -> bn1 -> ReLU -> maxpool -> operand format of the trunk kernels in ONE kernel (K9j): the full-resolution
            # stem output is never stored
            o = K.stem_eval(x, self.conv1.weight.detach(), self.bn1, PL)
            for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in stage:
                    o = blk.forward_eval_planes(o, PL)
            return o
        x = x.contiguous(memory_format=torch.channels_last)
        o = self.conv1(x)                 # library 7x7 stem (bf16 output under autocast)
        o = o.contiguous(memory_format=torch.channels_last)
        mp = self.maxpool
        if ((mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False)
                and o.shape[1] % 32 == 0 and o.dtype in (torch.float32, torch.bfloat16, torch.float16)
                and os.environ.get("HIAST_NO_STEM_TAIL", "0") != "1"):
            # bn1 -> ReLU -> maxpool -> operand format of the trunk kernels in one pass over the stem's output (K9f)
            o = K.stem_tail(o, self.bn1, PL)
            for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in stage:
                    o = blk.forward_eval_planes(o, PL)
            return o
        B, _, H, W = o.shape
        o = _from2d(K.bn_act_nhwc_infer(_nhwc2d(o), self.bn1, True), B, H, W)
        o = self.maxpool(o)
        B, C, H, W = o.shape
        o2d = _nhwc2d(o.contiguous(memory_format=torch.channels_last))
        if PL == 2:
            o = K.split_planes(o2d.float()).view(B, H, W, 2 * C)
        else:
            o = o2d.to(K.fmt_dtype(PL)).view(B, H, W, C)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                o = blk.forward_eval_planes(o, PL)
        return o

    @staticmethod
    def planes_to_feature(p, PL):
        """16-bit channels-last trunk output -> logical [B,C,h,w] tensor (channels-last memory)"""
        from hiast_amd import kernels as K
        B, h, w, CC = p.shape
        if PL == 2:
            return _from2d(K.merge_planes(p.view(B * h * w, CC)), B, h, w)
        return p.permute(0, 3, 1, 2)

    def forward(self, x, is_return_low=False):
        stem_partial = None
        if not is_return_low:
            PL = self.fast_eval_planes(x)
            if PL:
                return self.planes_to_feature(self.forward_eval_planes(x, PL), PL)
        if (x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") in HF.H16
                and os.environ.get("HIAST_TRAIN_NCHW", "0") != "1"):
            # mixed-precision step: the whole trunk runs channels-last (library stem -> 16-bit NHWC activations)
            if self.training and torch.is_grad_enabled():
                self.prepack(3 if torch.get_autocast_dtype("cuda") == torch.float16 else 1, adjoint=True)
            from hiast_amd import kernels as K
            if (self.bn1.training and K.stem_train_supported(self.conv1) and not x.requires_grad
                    and K.stem_train_shape_ok(x)):
                # K9k: own stem convolution straight from the fp32 NCHW batch, + the sums bn1 needs (HIAST_LIB_STEM=1: library)
                x, stem_partial = HF.stem_conv_train(x, self.conv1)
            else:
                x = x.contiguous(memory_format=torch.channels_last)
                x = self.conv1(x).contiguous(memory_format=torch.channels_last)
        else:
            x = self.conv1(x)
        batched = False
        if self.training and x.is_cuda:
            # one foreach add for the 104 num_batches_tracked counters instead of one tiny kernel per BN layer
            nbt = self.__dict__.get("_hiast_nbt")
            if nbt is None or nbt[0].device != x.device:
                nbt = [m.num_batches_tracked for m in self.modules()
                       if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.track_running_stats
                       and m.num_batches_tracked is not None]
                self.__dict__["_hiast_nbt"] = nbt
            if nbt:
                torch._foreach_add_(nbt, 1)
                batched = True
        prev = HF._nbt_batched[0]
        HF._nbt_batched[0] = batched or prev
        try:
            x = HF.maxpool(bn_act(self.bn1, x, partial=stem_partial), self.maxpool)
            low = self.layer1(x)
            x = self.layer4(self.layer3(self.layer2(low)))
        finally:
            HF._nbt_batched[0] = prev
        return (x, low) if is_return_low else x


def build_resnet101(pretrained=False, output_stride=32):
    """pretrained ImageNet weights are a URL download in the reference (resnet.py:196-197): not
    available offline; load a checkpoint through utils.load_model instead."""
    if pretrained:
        raise RuntimeError("ImageNet weights cannot be downloaded here; pass pretrained=False and load a state dict")
    if output_stride == 8:
        # layer3: stride removed, block0 dilation 1, others 2; layer4: block0 dilation 2, others 4
        return ResNet(strides=(1, 2, 1, 1), dilations=((1, 1), (1, 1), (1, 2), (2, 4)))
    return ResNet()
