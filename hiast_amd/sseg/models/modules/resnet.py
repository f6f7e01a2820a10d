"""ResNet-101 trunk without avgpool/fc, parameter names identical to the reference
(sseg/models/modules/resnet.py:58-98,101-190 — torchvision naming: conv1/bn1/layer{1..4}.{i}.
conv{1,2,3}/bn{1,2,3}/downsample.{0,1}) so released checkpoints load key-for-key.
Round 1: convolutions/BN run on PyTorch-ROCm (MIOpen); hand-written MFMA kernels for the
layer3/4 bottlenecks are the next row of the scope table."""
import torch.nn as nn
import torch.nn.functional as F

from hiast_amd import functional as HF


def bn_act(bn, x, res=None, relu=True):
    """relu?(bn(x) [+ res]) — one fused HIP op on the device (hiast_bn_*), plain torch on CPU tensors"""
    if x.is_cuda:
        return HF.bn_act(x, bn, res, relu)
    y = bn(x)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


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
        idt = x if self.downsample is None else bn_act(self.downsample[1], self.downsample[0](x), relu=False)
        o = bn_act(self.bn1, self.conv1(x))
        o = bn_act(self.bn2, self.conv2(o))
        return bn_act(self.bn3, self.conv3(o), res=idt)        # += identity, ReLU


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

    def forward(self, x, is_return_low=False):
        x = self.maxpool(bn_act(self.bn1, self.conv1(x)))
        low = self.layer1(x)
        x = self.layer4(self.layer3(self.layer2(low)))
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
