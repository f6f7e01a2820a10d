"""DeepLab-V2 / ResNet-101 (reference: sseg/models/modules/seg_models/deeplab_v2.py:8-69).

Same sub-module and state-dict names (backbone.*, aspp.conv2d_list.{0..3}.{weight,bias},
representation.0.{weight,bias}), same forward signature `(prediction, feature)` and LR groups.
The ASPP head runs as ONE hand-written gfx950 kernel over the four dilations on HIP tensors
(hiast_aspp_fwd / _bwd_data / _bwd_weight); on CPU tensors (config 1: CPU-only validate) the
Conv2d holders run as plain torch convs.  `representation` keeps its parameters for checkpoint
compatibility but is not computed: the reference computes it and drops the result (:63)."""
import torch
import torch.nn as nn

from hiast_amd import functional as HF
from hiast_amd.sseg.models.modules.resnet import build_resnet101
from hiast_amd.utils.registry.registries import SEG_MODEL


class ASPP_V2(nn.Module):

    def __init__(self, dilation_series, padding_series, num_classes):
        super().__init__()
        assert list(dilation_series) == list(padding_series)
        self.dilations = tuple(int(d) for d in dilation_series)
        self.conv2d_list = nn.ModuleList(
            nn.Conv2d(2048, num_classes, 3, stride=1, padding=d, dilation=d, bias=True) for d in self.dilations)
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)      # deeplab_v2.py:17-18

    def _wb(self):
        return [m.weight for m in self.conv2d_list], [m.bias for m in self.conv2d_list]

    def planes_ok(self):
        return self.conv2d_list[0].weight.shape[0] <= 32 and self.conv2d_list[0].weight.shape[1] % 128 == 0

    def forward_planes(self, p, PL):
        """inference on the trunk's 16-bit channels-last output [B,h,w,PL*2048] (no autograd)"""
        from hiast_amd import kernels as K
        ws, bs = self._wb()
        wt, _, bias = K.aspp2_pack_weights([w.detach().float().contiguous() for w in ws],
                                           [b.detach().float().contiguous() for b in bs], need_dgrad=False)
        if PL == 2:
            return K.aspp2_fwd(p, wt, bias, self.dilations, planes=2)
        return K.aspp2_fwd(p.permute(0, 3, 1, 2), wt, bias, self.dilations)

    def forward(self, x):
        if x.is_cuda:
            ws, bs = self._wb()
            gemm_ok = self.planes_ok()
            if gemm_ok and x.dtype in HF.H16:
                # mixed-precision step (teacher / student under autocast): channels-last GEMM + shift-add, 16-bit MFMA
                return HF.aspp_nhwc(x.contiguous(memory_format=torch.channels_last), ws, bs, self.dilations)
            if (gemm_ok and x.dtype == torch.float32 and not torch.is_grad_enabled()
                    and x.permute(0, 2, 3, 1).is_contiguous()):
                # fp32 inference on a channels-last feature: split-bf16 GEMM
                return HF.aspp_nhwc(x, ws, bs, self.dilations)
            return HF.aspp(x, ws, bs, self.dilations)   # exact-fp32 direct form (fp32 training, NCHW inference)
        out = self.conv2d_list[0](x)
        for m in self.conv2d_list[1:]:
            out = out + m(x)
        return out


@SEG_MODEL.register("DeepLab_V2")
class DeepLab_V2(nn.Module):

    def __init__(self, num_classes=19, output_dim=256):
        super().__init__()
        self.backbone = build_resnet101(pretrained=False, output_stride=8)
        self.aspp = ASPP_V2([6, 12, 18, 24], [6, 12, 18, 24], num_classes)
        self.representation = nn.Sequential(nn.Conv2d(2048, output_dim, 1))
        # dead branch: its output is discarded by the reference, so it never receives a gradient there
        # either (grad is None, Adam skips it).  Marking it non-trainable keeps DDP's reducer from waiting
        # for gradients that cannot arrive; the parameters stay in the state dict.
        for p in self.representation.parameters():
            p.requires_grad = False

    def forward(self, x, need_feat=True):
        """-> (prediction [B,C,H/8,W/8], feature [B,2048,H/8,W/8]); need_feat=False (internal fast path of this
        package's generator / trainers, which drop the feature) skips materialising the feature tensor"""
        PL = self.backbone.fast_eval_planes(x) if x.shape[1] == 3 else 0
        if PL and self.aspp.planes_ok():
            p = self.backbone.forward_eval_planes(x, PL)
            pred = self.aspp.forward_planes(p, PL)
            return pred, (self.backbone.planes_to_feature(p, PL) if need_feat else None)
        feat = self.backbone(x)            # [B, 2048, H/8, W/8]
        return self.aspp(feat), feat       # [B, C, H/8, W/8]

    def get_optimizer_params(self, lr):
        return [{"params": self.backbone.parameters(), "lr": lr},
                {"params": self.aspp.parameters(), "lr": lr * 10},
                {"params": self.representation.parameters(), "lr": lr * 10}]
