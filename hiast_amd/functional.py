"""torch.autograd bindings of the HIP kernels (one C-ABI call per forward / backward).

CUDA(HIP) tensors only.  Autocast: the losses and the pseudo-label kernels take fp32 inputs (as in the reference
even under apex O1); the ASPP head and the trunk convolutions / BatchNorms run in bf16 on channels-last activations
under bf16 autocast (_Aspp2Fn, _ConvNhwcFn, _BnActNhwcFn) and in fp32 otherwise (_AsppFn, _BnActFn).
"""
import os

import torch

from . import kernels as K
from . import switches as SW

ASPP_DILATIONS = (6, 12, 18, 24)
H16 = (torch.bfloat16, torch.float16)       # the two 16-bit types of the mixed-precision path (K.FMT_BF16 / K.FMT_FP16):
                                            # bf16 = round 1/2's kernels, fp16 = the reference's apex-O1 type


class _AsppFn(torch.autograd.Function):
    """y = Σ_i conv3x3(x; W_i, b_i, dilation=d_i, padding=d_i) — ASPP_V2.forward (deeplab_v2.py:20-24)"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w0, w1, w2, w3, b0, b1, b2, b3, dil):
        x = x.contiguous()
        ws = [w.contiguous() for w in (w0, w1, w2, w3)]
        bs = [b.contiguous() for b in (b0, b1, b2, b3)]
        wpack = K.aspp_pack_weights(ws, bs)
        Cout = ws[0].shape[0]
        y = K.aspp_fwd(x, wpack, Cout, dil)
        ctx.save_for_backward(x, wpack)
        ctx.dil = dil
        ctx.cin = x.shape[1]
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, wpack = ctx.saved_tensors
        gy = gy.contiguous().float()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = K.aspp_bwd_data(gy, wpack, ctx.cin, ctx.dil)
        gws, gb = [None] * 4, None
        if any(ctx.needs_input_grad[1:9]):
            gws, gb = K.aspp_bwd_weight(x, gy, ctx.dil)
        return (gx, gws[0], gws[1], gws[2], gws[3], gb, gb, gb, gb, None)


def aspp(x, weights, biases, dil=ASPP_DILATIONS):
    return _AsppFn.apply(x, *weights, *biases, tuple(dil))


class _Aspp2Fn(torch.autograd.Function):
    """ASPP_V2.forward on a CHANNELS-LAST feature map (fp32: split-bf16 arithmetic, inference only; bf16: the
    mixed-precision training step) as one GEMM + 33-tap shift-add (hiast_aspp2_*); y is fp32 NCHW."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, w0, w1, w2, w3, b0, b1, b2, b3, dil, need_bwd):
        ws = [w.detach().float().contiguous() for w in (w0, w1, w2, w3)]
        bs = [b.detach().float().contiguous() for b in (b0, b1, b2, b3)]
        if need_bwd and x.dtype not in H16:
            raise TypeError("the channels-last ASPP backward is the 16-bit one; fp32 training uses hiast_amd.functional.aspp")
        wt, wd, bias = K.aspp2_pack_weights(ws, bs, need_dgrad=need_bwd and ctx.needs_input_grad[0])
        y = K.aspp2_fwd(x, wt, bias, dil)
        ctx.save_for_backward(x, wd)
        ctx.dil = dil
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, wd = ctx.saved_tensors
        gy = gy.contiguous().float()
        want_dx = ctx.needs_input_grad[0]
        want_dw = any(ctx.needs_input_grad[1:9])
        gx, gws, gb = K.aspp2_bwd(x, gy, wd, ctx.dil, want_dx, want_dw)
        if gws is None:
            gws = [None] * 4
        return (gx, gws[0], gws[1], gws[2], gws[3], gb, gb, gb, gb, None, None)


def aspp_nhwc(x, weights, biases, dil=ASPP_DILATIONS):
    """x: logical [B,Cin,h,w] with channels-last memory (fp32 without grad, or bf16)"""
    need_bwd = torch.is_grad_enabled() and any(t.requires_grad for t in (x, *weights, *biases))
    return _Aspp2Fn.apply(x, *weights, *biases, tuple(dil), need_bwd)


class _UpsampleFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) (self_training_segmentor.py:27)"""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, H, W):
        ctx.hw = x.shape[2:]
        return K.upsample_bilinear_ac_fwd(x.contiguous(), H, W)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        return K.upsample_bilinear_ac_bwd(g.contiguous().float(), ctx.hw[0], ctx.hw[1]), None, None


def upsample_bilinear_ac(x, size):
    return _UpsampleFn.apply(x, int(size[0]), int(size[1]))


class _StLossFn(torch.autograd.Function):
    """The four self-training losses from LOW-RES logits in one kernel pass (K5-K8).
    Returns the raw sums (f64 [8]) as a non-differentiable side output and the four weighted
    losses; backward is one kernel producing d(low-res logits)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits_lr, teacher_lr, plbl, H, W, region, w_t, w_k, w_e, w_c):
        logits_lr = logits_lr.contiguous()
        if teacher_lr is not None:
            teacher_lr = teacher_lr.contiguous()
        plbl = plbl.contiguous()
        B, C, h, w = logits_lr.shape
        ws = K.st_loss_workspace(B, C, h, w, H, W, logits_lr.device)
        sums = K.st_loss_fwd(logits_lr, teacher_lr, plbl, H, W, region, ws)
        ctx.save_for_backward(logits_lr, teacher_lr, plbl, sums)
        ctx.ws = ws
        ctx.args = (H, W, region, w_t, w_k, w_e, w_c)
        s = sums.float()      # 0/0 -> NaN exactly like the reference's tensor divisions
        ce = w_t * s[0] / s[4]
        kld = w_k * s[1] / (C * s[4])
        ent = w_e * s[2] / (C * s[5])
        cst = w_c * s[3] / s[6]
        ctx.mark_non_differentiable(sums)
        return ce, kld, ent, cst, sums

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_ce, g_kld, g_ent, g_cst, _g_sums):
        logits_lr, teacher_lr, plbl, sums = ctx.saved_tensors
        H, W, region, w_t, w_k, w_e, w_c = ctx.args
        coef = torch.stack([g_ce * w_t, g_kld * w_k, g_ent * w_e, g_cst * w_c]).float().contiguous()
        d = K.st_loss_bwd(logits_lr, teacher_lr, plbl, H, W, region, sums, coef, ctx.ws)
        return (d,) + (None,) * 9


def st_loss(logits_lr, teacher_lr, plbl, size, region="ignored", w_t=1.0, w_k=0.1, w_e=1.0, w_c=0.5):
    """-> (ce, kld, ent, cst) 0-dim tensors, already multiplied by their weights.
    `teacher_lr` are the teacher's LOW-RES logits (or None); plbl uint8/int64 [B,H,W]."""
    ce, kld, ent, cst, _ = _StLossFn.apply(logits_lr, teacher_lr, plbl, int(size[0]), int(size[1]), region,
                                            float(w_t), float(w_k), float(w_e), float(w_c))
    return ce, kld, ent, cst


class _DInputFn(torch.autograd.Function):
    """upsample + softmax (+ prob_2_entropy) of LOW-RES logits in one kernel (K15); backward = one kernel
    recomputing the probabilities + the upsample adjoint."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits_lr, H, W, entropy):
        logits_lr = logits_lr.contiguous()
        ctx.save_for_backward(logits_lr)
        ctx.entropy = entropy
        return K.dinput_fwd(logits_lr, H, W, entropy)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (logits_lr,) = ctx.saved_tensors
        return K.dinput_bwd(logits_lr, g.float().contiguous(), ctx.entropy), None, None, None


def discriminator_input(logits_lr, size, entropy=False):
    """D_preprocess_fun(F.interpolate(logits, size)) of adversarial_warmup_segmentor.py:26-29,36,41"""
    return _DInputFn.apply(logits_lr, int(size[0]), int(size[1]), bool(entropy))


def _sync_world(bn):
    """number of ranks a BatchNorm layer's statistics are summed over: 1 = no exchange (plain BN, or no process group);
    N > 1 = SyncBN over N ranks; 0 = SyncBN in the one-rank rehearsal (utils/comm.py: the exchange is issued, the count is
    this rank's)"""
    import torch.distributed as dist
    if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
        from hiast_amd.utils import comm
        w = dist.get_world_size()
        return 0 if (w == 1 and comm.rehearsal()) else w
    return 1


_BN3_FUSION = os.environ.get("HIAST_NO_BN3_FUSION", "0") != "1"     # bn3's backward sums from the next block's conv1 data gradient
_OWN_S2_DGRAD = os.environ.get("HIAST_LIB_DGRAD_S2", "0") != "1"      # (=1: the library's data gradient for the strided 3x3)

# read ONCE at import: the ranks of a job must issue the collectives of the statistics group at matching points, a switch
# that is looked up per call could differ between them (or change between a forward and its backward)
_NO_ASYNC_STAT = os.environ.get("HIAST_NO_ASYNC_STAT", "0") == "1"


def _stat_all_reduce(t, async_op=False):
    """SyncBN exchange: sum of the per-rank statistics, on the communicator reserved for it (utils/comm.py: these
    [C,2] reduces sit between two kernels of the main stream and must not queue behind DDP's 32 MB gradient buckets or
    the pseudo-label histogram on the default communicator).  async_op: -> the work handle (the caller waits)"""
    from hiast_amd.utils import comm
    return comm.all_reduce(t, "stat", async_op=async_op)


def _stat_wait(work):
    """wait for an async SyncBN exchange (host-timed on the statistics communicator: utils/comm.HOST_S)"""
    from hiast_amd.utils import comm
    comm.wait(work, "stat")


class _BnActFn(torch.autograd.Function):
    """y = relu?(BN(x) (+ res)) in two streaming kernels (stats, apply) forward and two backward;
    SyncBN = one all-reduce of the [C,2] double sums between them (same exchange as the reference's SyncBN)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, training, momentum, eps, relu, world):
        x = x.contiguous()
        if res is not None:
            res = res.contiguous()
        ctx.training = training
        if not training:
            y, _, _ = K.bn_act_apply(x, res, gamma, beta, running_mean, running_var, None, 0.0, momentum, eps, relu)
            return y
        part = K.bn_stats(x)
        count = float(x.shape[0] * x.shape[2] * x.shape[3])
        if world != 1:     # SyncBN: sum the per-plane partials across ranks in place (one collective, no extra kernels)
            _stat_all_reduce(part)
            count *= max(world, 1)
        y, sm, si = K.bn_act_apply(x, res, gamma, beta, running_mean, running_var, part, count, momentum, eps, relu)
        ctx.save_for_backward(x, y, gamma, sm, si)
        ctx.relu, ctx.has_res, ctx.world, ctx.count = relu, res is not None, world, count
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise NotImplementedError("backward through inference-mode fused BN is not needed by the hot path")
        x, y, gamma, sm, si = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        part = K.bn_act_bwd_stats(dy, y, x, sm, si, ctx.relu)
        if ctx.world != 1:
            _stat_all_reduce(part)
        want_p = gamma is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        dx, dres, dg, db = K.bn_act_bwd_apply(dy, y, x, gamma, sm, si, part, ctx.count, ctx.relu,
                                              ctx.has_res and ctx.needs_input_grad[1], want_p)
        return (dx, dres, dg if ctx.needs_input_grad[2] else None, db if ctx.needs_input_grad[3] else None,
                None, None, None, None, None, None, None)


def _is_cl(t):
    """4-d tensor whose memory is channels-last contiguous (and not merely a degenerate NCHW view)"""
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


class _BnActNhwcFn(torch.autograd.Function):
    """_BnActFn on channels-last bf16 activations (hiast_bn_nhwc_*): the layout of the hand-written convolution
    kernels, so the student forward / backward needs no NCHW<->NHWC transposes."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, momentum, eps, relu, world, partial, box,
                stat_box=None):
        # batch statistics: from the producing convolution's epilogue when it delivered them, else one pass over x
        count = float(x.shape[0] * x.shape[2] * x.shape[3])
        # ReLU gate of the backward: recomputed from x when there is no residual input; with a residual the forward
        # writes a bit mask (1/16 of y's bytes) that both backward passes (and the identity hand-off) read instead of y
        ctx.gate = 0 if not relu else (3 if res is not None else 2)
        if ctx.gate == 3 and SW.on("HIAST_NO_BN_MASK"):
            ctx.gate = 1                                # A/B switch: gate read from y itself
        want_mask = ctx.gate == 3
        if world == 1 and partial is not None:      # no all-reduce point: statistics + apply straight from the partials
            out = K.bn_nhwc_apply_partial(x, res, gamma, beta, running_mean, running_var, partial, count, momentum, eps,
                                          relu, want_mask)
        else:
            sums = K.bn_nhwc_stats_from_partial(partial) if partial is not None else K.bn_nhwc_stats(x)
            if world != 1:     # SyncBN: one all-reduce of [C,2] double sums
                _stat_all_reduce(sums)
                count *= max(world, 1)
            out = K.bn_nhwc_apply(x, res, gamma, beta, running_mean, running_var, sums, count, momentum, eps, relu,
                                  want_mask)
        y, sm, si = out[:3]
        ctx.save_for_backward(x, out[3] if want_mask else (y if ctx.gate == 1 else None), gamma, beta, sm, si)
        ctx.has_res, ctx.world, ctx.count = res is not None, world, count
        # identity-branch hand-off (see Bottleneck.forward): when the block's first convolution has announced that its
        # data-gradient epilogue will add the ReLU-masked gradient of the identity branch itself, this backward
        # neither writes that masked copy (dres) nor returns it for autograd to add
        ctx.box = box if (box is not None and box.get("armed") and ctx.gate in (1, 3)) else None
        # statistics hand-off (see _ConvNhwcFn): the data-gradient launch of the ONE convolution that consumes y can emit
        # this layer's backward sums (Σg, Σ g*xhat) from its epilogue; it needs x and the batch statistics for that
        ctx.stat_box = None
        if stat_box is not None:
            # leftovers of a previous iteration whose backward through THIS layer was pruned (its input needed no gradient)
            # while the consuming convolution's backward had already started the exchange: finish it, drop the sums
            left = stat_box.pop("bwd_sums", None)
            if left is not None:
                _stat_wait(left[1])
            stat_box.pop("bwd_partial", None)
        if stat_box is not None and ctx.gate == 2:
            stat_box["bn"] = (x.detach(), sm, si, gamma, beta)
            stat_box["world"] = world
            ctx.stat_box = stat_box
        elif stat_box is not None and ctx.gate == 3:
            # bn3 of a bottleneck (residual + ReLU, gate bits written): conv1 of the NEXT identity block may deliver this layer's
            # backward sums from the epilogue that writes the gradient of y (hiast_xconv_dgrad_gated_bn_stats, round 4)
            stat_box["bn3"] = (x.detach(), out[3], sm, si)
            stat_box["world"] = world
            ctx.stat_box = stat_box
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, sm, si = ctx.saved_tensors
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dy = dy.contiguous(memory_format=torch.channels_last)
        fused = ctx.stat_box.pop("bwd_partial", None) if ctx.stat_box is not None else None
        early = ctx.stat_box.pop("bwd_sums", None) if ctx.stat_box is not None else None
        if early is not None:          # SyncBN: the consuming convolution's backward has reduced its epilogue sums and started
            sums, work = early         # the all-reduce BEFORE its weight gradient: the exchange ran beside that kernel
            _stat_wait(work)
        else:
            if fused is not None:      # per-block sums from the epilogue of the consuming convolution's data gradient
                sums = K.bn_nhwc_stats_from_partial(fused)
            else:
                sums = K.bn_nhwc_bwd_stats(dy, y, x, gamma, beta, sm, si, ctx.gate)
            if ctx.world != 1:
                _stat_all_reduce(sums)
        want_p = gamma is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        handoff = ctx.box is not None and ctx.needs_input_grad[1]
        dx, dres, dg, db = K.bn_nhwc_bwd_apply(dy, y, x, gamma, beta, sm, si, sums, ctx.count, ctx.gate,
                                               ctx.has_res and ctx.needs_input_grad[1] and not handoff, want_p)
        if handoff:
            ctx.box["gated"] = (dy, y)            # (gradient, bit mask): consumed by the block's conv1 backward
        return (dx, dres, dg if ctx.needs_input_grad[2] else None, db if ctx.needs_input_grad[3] else None,
                None, None, None, None, None, None, None, None, None)


class _ConvNhwcFn(torch.autograd.Function):
    """bias-free 1x1 / 3x3 ("same" padding = dilation) convolution of the ResNet trunk on channels-last bf16
    activations under mixed precision (the reference trains under apex O1: half-precision convolutions, fp32
    master weights): forward and data gradient on the LDS-DMA implicit-GEMM kernel (hiast_igemm_bn_act with the
    weight / its adjoint packed to bf16), weight gradient by the library (MIOpen) on the same channels-last
    tensors."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, stride, dil, want_stats, box, packed, in_bn=None, wgroup=None, xsum=None, in_bn3=None):
        xv = x.permute(0, 2, 3, 1)
        # in_bn3: x is the output of the PREVIOUS bottleneck (relu(bn3(.) + identity), registered there) and this is conv1 of an
        # identity block whose backward adds the identity gradient itself: its data gradient also delivers that bn3's sums
        ctx.in_bn3 = None
        if (in_bn3 is not None and "bn3" in in_bn3 and box is not None and stride == 1 and weight.shape[2] == 1
                and ctx.needs_input_grad[0] and _BN3_FUSION
                and K.xconv_dgrad_gated_bn_stats_ok(x.shape[0] * x.shape[2] * x.shape[3], weight.shape[0], weight.shape[1])):
            ctx.in_bn3 = in_bn3
        ctx.wgroup = wgroup                   # (group dict, slot): the weight gradient is deferred to _WGroupFn.backward
        # xsum = (dict, role): two convolutions read the SAME x (conv1 and the stride-1 downsample of a stage-entry block).
        # The one whose backward runs first ('give': conv1, created later in the forward) hands its data gradient over
        # instead of returning it, the other ('take') adds it in the epilogue of its own data-gradient launch: x receives
        # ONE gradient and autograd's add kernel over the block input (33 - 134 MB, three per step) is not launched.
        ctx.xsum = xsum if (xsum is not None and stride == 1 and ctx.needs_input_grad[0]) else None
        if ctx.xsum is not None:
            # the 'take' convolution (the downsample: its forward runs first) announces itself; 'give' hands its gradient over only
            # to a taker that exists — if the downsample leg took another path (e.g. its BatchNorm in eval mode while bn1 trains)
            # nothing would ever pop the gradient and conv1's input gradient would be dropped silently
            if ctx.xsum[1] == "take":
                ctx.xsum[0]["taker"] = True
            elif not ctx.xsum[0].get("taker"):
                ctx.xsum = None
        # in_bn: x = relu(bn(x0)) of a BatchNorm that registered itself there (and has no other consumer): the data
        # gradient of this convolution then also delivers that BatchNorm's backward sums (hiast_igemm_dgrad_bn_stats)
        ctx.in_bn = in_bn if (in_bn is not None and "bn" in in_bn and stride == 1 and ctx.needs_input_grad[0]
                              and not SW.on("HIAST_NO_BN_BWD_FUSION")) else None
        ctx.set_materialize_grads(False)     # no zero tensor for the (non-differentiable) statistics output
        ctx.wpt = None
        ctx.box = None
        if box is not None and stride == 1 and ctx.needs_input_grad[0]:
            box["armed"] = True                    # this conv's backward will take over the identity-branch gradient
            ctx.box = box
        # (the adjoint weight serves the stride-1 data gradients and the transposed form of the 3x3 / stride-2 one)
        need_adj = ctx.needs_input_grad[0] and (stride == 1 or (stride == 2 and weight.shape[2] == 3 and dil == 1 and _OWN_S2_DGRAD))
        if packed is not None and packed[0] is not None and (not need_adj or packed[1] is not None):
            wp, ctx.wpt = packed                   # kernel-format copies kept current by ResNet.prepack (one launch)
        elif need_adj:                             # forward + adjoint (data-gradient) weight in one pack launch
            wp, ctx.wpt = K.pack_conv_weight(weight, K.fmt_of(x), both=True)
        else:
            wp = K.pack_conv_weight(weight, K.fmt_of(x))
        ctx.save_for_backward(x, weight)
        ctx.geo = (stride, dil)
        if want_stats:       # + per-block Σy, Σy² of the stored outputs for the BatchNorm that follows
            y, partial = K.igemm_bn_act(xv, wp, 1, None, None, False, stride, dil, want_stats=True)
            ctx.mark_non_differentiable(partial)
            return y.permute(0, 3, 1, 2), partial
        return K.igemm_bn_act(xv, wp, 1, None, None, False, stride, dil).permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, _dpartial=None):
        if dy is None:
            return None, None, None, None, None, None, None, None, None, None, None
        x, weight = ctx.saved_tensors
        stride, dil = ctx.geo
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dy = dy.contiguous(memory_format=torch.channels_last)
        k = weight.shape[2]
        pad = dil if k == 3 else 0
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        own_s2 = need_x and stride == 2 and k == 3 and dil == 1 and _OWN_S2_DGRAD
        lib_x = need_x and stride != 1 and not own_s2
        if own_s2:              # hiast_igemm_dgrad_s2: the transposed form of the tile kernel (was the library's, round 3)
            wpt = ctx.wpt if ctx.wpt is not None else K.pack_conv_weight(weight, K.fmt_of(x), transpose=True)
            dx = K.igemm_dgrad_s2(dy.permute(0, 2, 3, 1), wpt, x.shape[2], x.shape[3]).permute(0, 3, 1, 2)
        if need_x and stride == 1:
            wpt = ctx.wpt if ctx.wpt is not None else K.pack_conv_weight(weight, K.fmt_of(x), transpose=True)
            gated = ctx.box.pop("gated", None) if ctx.box is not None else None
            if ctx.in_bn is not None and gated is None:
                bx, sm, si, gamma, beta = ctx.in_bn["bn"]
                dxv, bpartial = K.igemm_dgrad_bn_stats(dy.permute(0, 2, 3, 1), wpt, dil, bx.permute(0, 2, 3, 1), gamma, beta,
                                                       sm, si)
                if ctx.in_bn.get("world", 1) != 1 and not _NO_ASYNC_STAT:
                    # SyncBN: reduce the per-block sums now and start their all-reduce; the weight gradient below (50 - 300 us,
                    # on the main stream under DDP) runs while the [C,2] exchange (a latency, ~30 us on 8 devices) is in
                    # flight — the BatchNorm's backward then finds the reduced sums instead of waiting for the exchange
                    bsums = K.bn_nhwc_stats_from_partial(bpartial)
                    ctx.in_bn["bwd_sums"] = (bsums, _stat_all_reduce(bsums, async_op=True))
                else:
                    ctx.in_bn["bwd_partial"] = bpartial
                dx = dxv.permute(0, 3, 1, 2)
            elif gated is not None and ctx.in_bn3 is not None and gated[1].dtype == torch.uint8:
                # ... and the backward sums of the previous block's bn3, whose output's gradient this launch writes (K9e')
                bx3, bmask, sm3, si3 = ctx.in_bn3["bn3"]
                dxv, bpartial = K.xconv_dgrad_gated_bn_stats(dy.permute(0, 2, 3, 1), wpt, gated[0].permute(0, 2, 3, 1), gated[1],
                                                             bx3.permute(0, 2, 3, 1), bmask, sm3, si3)
                if ctx.in_bn3.get("world", 1) != 1 and not _NO_ASYNC_STAT:
                    bsums = K.bn_nhwc_stats_from_partial(bpartial)
                    ctx.in_bn3["bwd_sums"] = (bsums, _stat_all_reduce(bsums, async_op=True))
                else:
                    ctx.in_bn3["bwd_partial"] = bpartial
                dx = dxv.permute(0, 3, 1, 2)
            elif gated is not None:    # + dy_block * (y_block > 0): the identity branch's gradient, in the epilogue
                dx = K.igemm_bn_act(dy.permute(0, 2, 3, 1), wpt, 1, None, gated[0].permute(0, 2, 3, 1), False, 1, dil,
                                    res_gate=(gated[1] if gated[1].dtype == torch.uint8
                                              else gated[1].permute(0, 2, 3, 1))).permute(0, 3, 1, 2)
            else:
                other = None
                if ctx.xsum is not None and ctx.xsum[1] == "take":
                    other = ctx.xsum[0].pop("dx", None)       # the gradient conv1's backward left for us (same shape as dx)
                    ctx.xsum[0]["taken"] = True
                dx = K.igemm_bn_act(dy.permute(0, 2, 3, 1), wpt, 1, None, None if other is None else other.permute(0, 2, 3, 1),
                                    False, 1, dil).permute(0, 3, 1, 2)
        if ctx.xsum is not None and ctx.xsum[1] == "give" and dx is not None and not ctx.xsum[0].get("taken"):
            ctx.xsum[0]["dx"] = dx             # the downsample convolution's backward (it runs after this one) adds it
            dx = None
        own_w = need_w and not SW.on("HIAST_LIB_WGRAD") and K.conv_wgrad_preferred(
            weight.shape[1], weight.shape[0], k, stride) and (k == 1 or dy.shape[3] >= 4)
        small_w = need_w and not own_w and not SW.on("HIAST_LIB_WGRAD") and K.conv_wgrad_small_supported(
            weight.shape[1], weight.shape[0], k, stride)
        # The weight gradient is off the critical path of the backward pass (nothing but the optimiser consumes it)
        # and MFMA-bound, while the BatchNorm backward passes that follow on the main stream are HBM-bound: in a
        # single-process run it goes to a side stream and co-runs with them (wgrad_stream_join() before the optimiser
        # step).  Under DDP the reducer reads gradients as soon as autograd delivers them, so it stays on the main stream.
        if need_w and own_w and ctx.wgroup is not None:
            # grouped weight gradient (hiast_conv_wgrad_group_nhwc): hand the operands to the bottleneck's _WGroupFn node,
            # whose backward runs when the last of the block's convolutions has delivered its pair, and return a
            # zero-stride placeholder of the weight's shape (no memory, no kernel) for autograd to pass on to it
            grp, slot = ctx.wgroup
            grp["jobs"][slot] = (dy.permute(0, 2, 3, 1), x.permute(0, 2, 3, 1), k, stride, dil)
            return dx, _wgrad_token(weight), None, None, None, None, None, None, None, None, None
        side = wgrad_side_stream(x.device) if (need_w and not lib_x) else None
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)
            dy.record_stream(side)
            x.record_stream(side)
        with torch.cuda.stream(side if side is not None else main):
            if own_w:            # transposed-read GEMM over the pixel index (hiast_conv_wgrad_nhwc)
                dw = K.conv_wgrad_nhwc(dy.permute(0, 2, 3, 1), x.permute(0, 2, 3, 1), k, stride, dil)
                need_w = False
            elif small_w:        # the layers below 256 channels (hiast_conv_wgrad_small_nhwc)
                dw = K.conv_wgrad_small_nhwc(dy.permute(0, 2, 3, 1), x.permute(0, 2, 3, 1), k, stride, dil)
                need_w = False
            if need_w or lib_x:
                wl = torch.empty(weight.shape, dtype=x.dtype, device=weight.device)
                if lib_x:
                    wl = weight.to(x.dtype)
                gx, gw, _ = torch.ops.aten.convolution_backward(dy, x, wl, None, (stride, stride), (pad, pad), (dil, dil),
                                                                False, (0, 0), 1, (lib_x, need_w, False))
                if lib_x:
                    dx = gx
                if need_w:
                    # fp32, NCHW-contiguous like the parameter (DDP's gradient-layout contract): one cast+layout kernel
                    dw = gw.to(dtype=weight.dtype, memory_format=torch.contiguous_format)
                    if dw.stride() != weight.stride() and dw.is_contiguous() and weight.is_contiguous():
                        dw = dw.view(-1).view(weight.shape)     # 1x1 kernels: canonical strides for the size-1 dims
                                                                # (DDP compares strides with its bucket view literally)
        if side is not None and dw is not None:
            dw.record_stream(main)
        return dx, dw, None, None, None, None, None, None, None, None, None


_wgrad_tokens = {}


def _wgrad_token(weight):
    """a zero-stride tensor of the weight's shape / dtype / device: what a grouped convolution's backward returns in place of
    its weight gradient (autograd checks the metadata and hands it to _WGroupFn.backward, which ignores the values)"""
    key = (weight.device, weight.dtype)
    z = _wgrad_tokens.get(key)
    if z is None:
        z = _wgrad_tokens[key] = torch.zeros((), dtype=weight.dtype, device=weight.device)
    return z.expand(weight.shape)


class _WGroupFn(torch.autograd.Function):
    """The weights of ONE bottleneck pass through this node on their way to its convolutions (identity forward).  Its
    backward runs once all of them have run theirs — the operands (dY, X) of the three products are then all there — and
    computes the weight gradients in one grouped launch (K.conv_wgrad_group: the chip is filled once, with one fp32
    partial tile per CU, for all three instead of once per convolution; reference: autograd of the three nn.Conv2d of
    Bottleneck.forward, sseg/models/modules/resnet.py:78-98).  The gradients reach the parameters (and DDP's reducer
    hooks) complete, from this node."""

    @staticmethod
    def forward(ctx, grp, *ws):
        ctx.grp = grp
        ctx.set_materialize_grads(False)
        return tuple(w.view_as(w) for w in ws)

    @staticmethod
    def backward(ctx, *tokens):
        grp = ctx.grp
        jobs = grp["jobs"]
        live = [i for i, j in enumerate(jobs) if j is not None and tokens[i] is not None]
        # a convolution that did not take the grouped path after all (a shape its backward sends elsewhere) has returned
        # its real weight gradient: passed through
        out = [tokens[i] if jobs[i] is None else None for i in range(len(jobs))]
        if live:
            dev = jobs[live[0]][1].device
            side = wgrad_side_stream(dev)
            main = torch.cuda.current_stream()
            if side is not None:
                side.wait_stream(main)
                for i in live:
                    jobs[i][0].record_stream(side)
                    jobs[i][1].record_stream(side)
            with torch.cuda.stream(side if side is not None else main):
                dws = K.conv_wgrad_group([jobs[i] for i in live]) if len(live) > 1 else [K.conv_wgrad_nhwc(*jobs[live[0]])]
            for i, dw in zip(live, dws):
                if side is not None:
                    dw.record_stream(main)
                out[i] = dw
        for i in range(len(jobs)):
            jobs[i] = None                                  # drop the operand references
        return (None, *out)


def wgroup_weights(convs, x):
    """-> (group dict, [weight views]) when the weight gradients of `convs` (the convolutions of one bottleneck) can be
    computed by one grouped launch on the training path of x, else (None, None)"""
    if (len(convs) < 2 or len(convs) > 4 or SW.on("HIAST_NO_WGROUP")
            or SW.on("HIAST_LIB_WGRAD") or not torch.is_grad_enabled()):
        return None, None
    for c in convs:
        k = c.kernel_size[0]
        if (not c.weight.requires_grad or not conv_nhwc_ok(x, c) or c.stride[0] != 1
                or not K.conv_wgrad_preferred(c.in_channels, c.out_channels, k, 1)):
            return None, None
    # A grouped launch is ONE round of (tiles x pixel ranges) <= 256 blocks: worth it when that fills the chip.  layer3:
    # 4 + 9 + 4 tiles x 15 ranges = 255 blocks (190 us against 239 us one by one); layer4: 16 + 36 + 16 tiles x 3 ranges
    # = 204 blocks leave a fifth of the CUs idle (698 against 622 us) — there only the 1x1 pair is grouped (32 tiles x 8
    # ranges: 265 against 300 us) and the 3x3 keeps its own launch (36 tiles x 7 ranges).
    tiles = [(c.in_channels // 256) * (c.out_channels // 256) * c.kernel_size[0] ** 2 for c in convs]

    cus = K.grid_cus()   # (hiast_device_cus - the CU reserve: the number the library's planner splits the pixel ranges for)

    def fill(idx):      # fraction of those CUs one round of (tiles x pixel ranges) blocks occupies
        t = sum(tiles[i] for i in idx)
        return t * (cus // t) / float(cus) if 0 < t <= cus else 0.0
    member = list(range(len(convs)))
    if fill(member) < 0.9:
        member = [i for i in member if convs[i].kernel_size[0] == 1]
        if len(member) < 2 or fill(member) < 0.9:
            return None, None
    grp = {"jobs": [None] * len(member)}
    views = _WGroupFn.apply(grp, *[convs[i].weight for i in member])
    wv = [None] * len(convs)
    slot = [None] * len(convs)
    for s_, i in enumerate(member):
        wv[i], slot[i] = views[s_], s_
    grp["slot"] = slot
    return grp, wv


class _StemConvFn(torch.autograd.Function):
    """the 7x7 stride-2 stem convolution of the mixed-precision training forward on the hand-written kernels (K9k:
    hiast_stem_train_fwd / hiast_stem_wgrad): reads the fp32 NCHW batch as it is (no channels-last copy, no cast), writes
    16-bit channels-last rows and the per-block sums for the batch-statistics BatchNorm behind it; backward = the weight
    gradient only (the image needs none).  Reference: `self.conv1(x)` in ResNet.forward, resnet.py:177-180, under apex O1."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, fmt):
        x = x.contiguous()
        y, partial = K.stem_train_fwd(x, weight.detach(), fmt)
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(partial)
        return y.permute(0, 3, 1, 2), partial

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, _dpartial=None):
        if dy is None or not ctx.needs_input_grad[1]:
            return None, None, None
        x, = ctx.saved_tensors
        dyv = dy.permute(0, 2, 3, 1)
        if not dyv.is_contiguous():
            dyv = dyv.contiguous()
        side = wgrad_side_stream(x.device)
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)
            dyv.record_stream(side)
            x.record_stream(side)
        with torch.cuda.stream(side if side is not None else main):
            dw = K.stem_wgrad(x, dyv)
        if side is not None:
            dw.record_stream(main)
        return None, dw, None


def stem_conv_train(x, conv):
    """-> (y logical [B,64,Hc,Wc] 16-bit with channels-last memory, partial) of the trunk's stem convolution under 16-bit
    autocast (K9k)"""
    dt = torch.get_autocast_dtype("cuda")
    return _StemConvFn.apply(x.float() if x.dtype != torch.float32 else x, conv.weight,
                             K.FMT_FP16 if dt == torch.float16 else K.FMT_BF16)


_wgrad_streams = {}
_wgrad_overlap = [False]
_eval_streams = {}


def new_stream(device):
    """a side stream for `device`: a plain torch stream — or, while a CU reserve is in effect (comm.apply_cu_reserve: N > 1
    with HIAST_RESERVE_CUS=n), a stream whose kernels cannot be placed on the reserved CUs (K.reserved_stream), so that the
    pseudo-label / teacher / weight-gradient sequences leave a collective's kernel a CU"""
    n = K.reserve_cus() if torch.device(device).type == "cuda" else 0
    return K.reserved_stream(n, device) if n > 0 else torch.cuda.Stream(device=device)
_eval_trunks = {}


def prepack_eval_trunks(model, x):
    """bring the kernel-format weight copies of the model's trunk(s) up to date for an inference forward of x on the
    current stream (one launch per trunk, only when a weight has changed) -> the (trunk, plan) pairs in use"""
    trunks = _eval_trunks.get(id(model))
    if trunks is None or trunks[0]() is not model:
        import weakref
        trunks = _eval_trunks[id(model)] = (weakref.ref(model), [m for m in model.modules()
                                                                 if hasattr(m, "prepack") and hasattr(m, "fast_eval_planes")])
    used = []
    for m in trunks[1]:
        PL = m.fast_eval_planes(x)
        if PL:
            m.prepack(PL)
            ent = m.__dict__.get("_hiast_plans", {}).get((PL, False))
            if ent is not None:
                used.append(ent[0])
    return used


def eval_forward_split(model, x, parts=None):
    """low-resolution logits of an inference forward, computed as `parts` sub-batches on streams of their own (same
    values: an eval forward treats every image alone).  On the fp32-class pseudo-label forward of 8 images two
    half batches finish 0.8-1.0 ms earlier than one launch sequence (bench.py: 56.6 -> 55.8 ms/step): a launch over 4
    images fills half of the CUs, and the two sequences drift apart, so that the HBM-saturated epilogue phases of the
    1x1 + residual launches of one run beside the matrix-pipe phases of the other.
    parts: None = 2 for batches of 8 or more images that divide evenly (HIAST_EVAL_SPLIT overrides, 1 = off)."""
    B = x.shape[0]
    if parts is None:
        parts = int(os.environ.get("HIAST_EVAL_SPLIT", "2" if B >= 8 else "1"))
    if parts <= 1 or B % parts != 0 or not x.is_cuda:
        return model(x, lowres=True)
    main = torch.cuda.current_stream()
    # kernel-format weight copies that are cached per module (ResNet.prepack) are brought up to date HERE, on the main
    # stream, before the sub-batches fork: the first forward to notice a stale copy re-packs it on ITS stream, and the
    # other streams would read the buffers while they are being written
    prepack_eval_trunks(model, x)
    key = (x.device, parts)
    side = _eval_streams.get(key)
    if side is None:
        side = _eval_streams[key] = [new_stream(x.device) for _ in range(parts - 1)]
    sub = B // parts
    outs = [None] * parts
    # the sub-batch launches run side by side: tell the tile kernel's launcher (it would otherwise give a half-chip launch the
    # 128 x 128 / two-blocks-per-CU form, which wins on a launch that runs ALONE: igemm_kernel.h, IG_HALF_AUTO)
    # The hint is state of THIS thread inside the library (hiast_igemm_set_cosched), passed with every launch this thread makes
    # and with the statistics-row counts alike — not an environment variable toggled under other threads' getenv.
    with K.cosched():
        for i, st in enumerate(side):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs[i + 1] = model(x[(i + 1) * sub:(i + 2) * sub], lowres=True)
        outs[0] = model(x[:sub], lowres=True)
    for i, st in enumerate(side):
        main.wait_stream(st)
        outs[i + 1]["logits_lowres"].record_stream(main)
    out = dict(outs[0])
    out["logits_lowres"] = torch.cat([o["logits_lowres"] for o in outs], 0)
    return out


class GraphedEval:
    """`model(x, lowres=True)['logits_lowres']` of an inference forward (no autograd, eval mode), optionally replayed from a
    captured HIP graph (HIAST_GRAPH_EVAL=1; default: eager calls).  The graph holds exactly the launches of the eager
    forward (same kernels, same order, sub-batches on the same streams: bit-identical logits, tests/test_gpu_round2.py);
    the kernel-format weight copies are refreshed OUTSIDE of it, before every replay, by the same `prepack` call the
    eager forward makes (the teacher's weights change with every EMA step, the buffers they are packed into do not).
    The first WARMUP calls per input signature run eagerly (library algorithm search, lazy allocations), the next one
    captures; the returned tensor is then the graph's own output buffer, overwritten by the next call with the same
    signature.  Off by default in the trainers and bench.py, where it does not pay: the ~330 launches of a forward cost
    the host 3 ms (pseudo-label forward: 3.0 -> 0.56 ms of enqueue time) but the host runs 17-48 ms ahead of the device anyway
    — bench.py 61.4 (graphs) vs 60.2 ms/step (eager) on one box, the DataLoader-fed trainer unchanged.  The generator, whose
    host used to start every batch without a lead over the device, gained 10 % from it (147.6 -> 163.0 images/s) until its
    batches were pipelined (forward of batch t+1 enqueued before the histogram of batch t is awaited): 487 eager vs 473
    replayed.
    Round 6 — an automatic mode for small batches exists (HIAST_GRAPH_EVAL unset and HIAST_GRAPH_EVAL_MAX_BATCH=n: forwards over
    at most n images are replayed) and is OFF (n = 0), by measurement (profiles/r06_ab_graph_eval_small_batch.txt): the forwards
    of a launch-bound step ARE cheaper to replay (pseudo-label forward 1.98 -> 0.67 ms of host time at one image), but on this
    runtime (ROCm 7.2) a replayed graph of ~330 kernel nodes slows the eager launches that follow it — bench.py --batch 1
    28.1 -> 34.3 ms/step (the backward's enqueue time 14.6 -> 21.7 ms), --batch 2 21.8 -> 24.3, the generator at batch size 2
    339 -> 315 images/s.  Inside another capture (GraphedTrainStep) the forward always runs eagerly — into that graph."""
    WARMUP = 2
    AUTO_MAX_BATCH = int(os.environ.get("HIAST_GRAPH_EVAL_MAX_BATCH", "0"))

    def __init__(self, model, amp_dtype=None, parts=None, graph=False, auto=True):
        """parts: as eval_forward_split takes them (None = its default: two sub-batches for 8 or more images, 1 = one
        launch sequence); graph: this caller's default; HIAST_GRAPH_EVAL=1 / 0 overrides it for every caller; auto=False: no
        automatic replay of small batches either (a caller whose whole iteration is a captured graph already)"""
        self.model, self.amp_dtype, self.parts = model, amp_dtype, parts
        self.entries = {}
        env = os.environ.get("HIAST_GRAPH_EVAL", "1" if graph else "")
        self.enabled = (None if auto else False) if env == "" else env == "1"        # None: automatic (small batches)

    def wants_graph(self, x):
        return self.enabled if self.enabled is not None else x.shape[0] <= self.AUTO_MAX_BATCH

    def _autocast(self):
        return torch.autocast("cuda", dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None,
                              cache_enabled=False)

    def _eager(self, x, parts):
        with torch.no_grad(), self._autocast():
            return eval_forward_split(self.model, x, parts)["logits_lowres"].float()

    def _fingerprint(self):
        ps = self.model.__dict__.get("_hiast_fp_params")
        if ps is None:
            allp = list(self.model.parameters()) + list(self.model.buffers())
            ps = self.model.__dict__["_hiast_fp_params"] = [allp[0], allp[len(allp) // 2], allp[-1]]
        return tuple(p.data_ptr() for p in ps)

    def __call__(self, x, parts="ctor", eager=False):
        """parts: sub-batches for this call as eval_forward_split takes them (None = its default: 2 for 8 or more images),
        "ctor" = the constructor's; eager: no graph this time"""
        parts = self.parts if isinstance(parts, str) else parts
        if (eager or not x.is_cuda or self.model.training or not self.wants_graph(x)
                or torch.cuda.is_current_stream_capturing()):
            return self._eager(x, parts)
        key = (tuple(x.shape), x.dtype, x.device, self.amp_dtype, parts)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = {"calls": 0, "graph": None}
        if e["graph"] is not None:
            with torch.no_grad(), self._autocast():
                plans = prepack_eval_trunks(self.model, x)
            if self._fingerprint() == e["fp"] and len(plans) == len(e["plans"]) and all(a is b for a, b in zip(plans, e["plans"])):
                e["x"].copy_(x)
                e["graph"].replay()
                return e["y"]
            e["graph"], e["calls"] = None, 0            # the model's storage has moved: start over
        e["calls"] += 1
        if e["calls"] <= self.WARMUP or e.get("failed"):
            return self._eager(x, parts)
        try:
            with torch.no_grad(), self._autocast():
                plans = prepack_eval_trunks(self.model, x)      # nothing stale is left to be packed inside the graph
            xs = torch.empty_like(x)
            xs.copy_(x)
            g = torch.cuda.CUDAGraph()
            cur = torch.cuda.current_stream()
            cap = torch.cuda.Stream(device=x.device)
            cap.wait_stream(cur)
            with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                y = self._eager(xs, parts)
            cur.wait_stream(cap)
        except Exception as err:            # the eager forward is the same computation: say so and go on
            import warnings
            warnings.warn("hiast_amd: graph capture of the inference forward failed (%s: %s); running it eagerly"
                          % (type(err).__name__, err))
            e["failed"] = True
            return self._eager(x, parts)
        e.update(graph=g, x=xs, y=y, plans=plans, fp=self._fingerprint())
        g.replay()
        return y


def enable_wgrad_overlap(on=True):
    """opt in to weight gradients on a side stream.  The caller then owes a wgrad_stream_join() before anything on
    the main stream reads a .grad (the trainers of this package do: BaseTrainer.update_model, bench.py) and must
    start every backward with .grad = None (zero_grad(set_to_none=True)): autograd then only stores the tensor."""
    _wgrad_overlap[0] = bool(on)


def wgrad_side_stream(device):
    """the side stream weight gradients run on, or None (not enabled / DDP / HIAST_NO_WGRAD_STREAM=1)"""
    if not _wgrad_overlap[0] or os.environ.get("HIAST_NO_WGRAD_STREAM", "0") == "1":
        return None
    from hiast_amd.utils import comm
    if comm.multi():
        return None
    st = _wgrad_streams.get(device)
    if st is None:
        st = new_stream(device)
        _wgrad_streams[device] = st
    return st


def wgrad_stream_join():
    """make the current stream wait for every weight gradient issued on a side stream (call before optimizer.step()
    or before reading .grad)"""
    for st in _wgrad_streams.values():
        torch.cuda.current_stream(st.device).wait_stream(st)


def conv_nhwc_ok(x, conv):
    """the trunk convolutions this path covers: bf16 channels-last input, bias-free 1x1 (stride 1) or 3x3 with
    padding == dilation, channel counts the kernel tiles (multiples of 64)"""
    if not (x.is_cuda and x.dtype in H16 and _is_cl(x)) or conv.bias is not None or conv.groups != 1:
        return False
    k = conv.kernel_size
    if k not in ((1, 1), (3, 3)) or conv.stride[0] != conv.stride[1] or conv.in_channels % 64 or conv.out_channels % 64:
        return False
    if k == (1, 1):      # a strided 1x1 (the stage-entry downsample) = the stride-1 kernel on the subsampled input
        return conv.stride[0] in (1, 2) and conv.padding == (0, 0)
    return conv.padding == conv.dilation and conv.dilation[0] == conv.dilation[1] and conv.stride[0] in (1, 2)


class _MaxPool3x3s2ClFn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) of the stem on channels-last bf16 activations (K18): one byte of window position per element
    instead of ATen's int64 index, backward as a gather (95 + 200 us -> fwd / bwd of the layer at 8 x 1024 x 512)"""

    @staticmethod
    def forward(ctx, x):
        y, idx = K.maxpool3x3s2_cl_fwd(x)
        ctx.save_for_backward(idx)
        ctx.hw, ctx.dtype = tuple(x.shape[2:]), x.dtype
        ctx.mark_non_differentiable(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        if dy.dtype != ctx.dtype:
            dy = dy.to(ctx.dtype)
        return K.maxpool3x3s2_cl_bwd(dy.contiguous(memory_format=torch.channels_last), idx, *ctx.hw)


def maxpool(x, pool):
    """pool(x) for an nn.MaxPool2d; the stem's 3x3 / stride 2 / padding 1 pooling of a channels-last bf16 device
    activation runs on the K18 kernels (HIAST_LIB_MAXPOOL=1: the library's)"""
    if (x.is_cuda and x.dtype in H16 and _is_cl(x) and x.shape[1] % 8 == 0
            and (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False)
            and not pool.return_indices and os.environ.get("HIAST_LIB_MAXPOOL", "0") != "1"):
        return _MaxPool3x3s2ClFn.apply(x)
    return pool(x)


class _SubsampleClFn(torch.autograd.Function):
    """x[:, :, ::s, ::s] of a channels-last tensor as a channels-last tensor.  The backward scatters into a CHANNELS-LAST
    zero tensor: autograd's own slice backward builds an NCHW-contiguous one, and the sum with the other (channels-last)
    gradient of the same activation then runs torch's strided add (389 us for the 134 MB layer1 output, once per step)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s, ctx.shape = s, x.shape
        return x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, dy):
        g = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last).zero_()
        g[:, :, ::ctx.s, ::ctx.s] = dy
        return g, None


def conv_nhwc(x, conv, want_stats=False, box=None, in_bn=None, wgroup=None, xsum=None, in_bn3=None):
    """-> y, or (y, partial) with want_stats (see igemm_bn_act); box: identity-branch hand-off of a bottleneck;
    in_bn: statistics hand-off of the BatchNorm whose output x is (bn_act(..., stat_box=in_bn))"""
    w = conv.weight
    wv = w if wgroup is None else wgroup[2]      # wgroup = (group dict, slot, this weight as it comes out of _WGroupFn)
    fmt = K.fmt_of(x)
    fwd = conv.__dict__.get("_hiast_packed", {}).get(fmt)
    adj = conv.__dict__.get("_hiast_packed_adj", {}).get(fmt)
    ok = lambda e: e is not None and e[0] == w._version and e[1] == w.data_ptr()
    packed = (fwd[2] if ok(fwd) else None, adj[2] if ok(adj) else None)
    stride = conv.stride[0]
    if conv.kernel_size == (1, 1) and stride != 1:
        # strided 1x1: every s-th pixel through the stride-1 kernel (forward, data and weight gradient all stay on the
        # hand-written path, and dW arrives with the parameter's own strides — the library's channels-last dW made
        # DDP copy the bucket view); autograd scatters the data gradient back into the skipped pixels
        x = _SubsampleClFn.apply(x, stride)
        stride = 1
    return _ConvNhwcFn.apply(x, wv, stride, conv.dilation[0], bool(want_stats), box, packed, in_bn,
                             None if wgroup is None else (wgroup[0], wgroup[1]), xsum, in_bn3)


_nbt_batched = [False]      # set by ResNet.forward while it has already advanced every num_batches_tracked at once


def bn_act(x, bn, res=None, relu=True, partial=None, box=None, stat_box=None):
    """Fused replacement of `relu(bn(x) [+ res])` for a torch BatchNorm2d / SyncBatchNorm module `bn`
    (which keeps owning the parameters and running statistics)."""
    training = bn.training or (bn.running_mean is None)
    if training and bn.track_running_stats and bn.num_batches_tracked is not None and not _nbt_batched[0]:
        bn.num_batches_tracked.add_(1)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    if (training and x.dtype in H16 and _is_cl(x) and not x.is_contiguous()
            and K.bn_nhwc_supported(x.shape[1]) and (res is None or (res.dtype == x.dtype and _is_cl(res)))):
        return _BnActNhwcFn.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, relu,
                                  _sync_world(bn), partial, box, stat_box)
    return _BnActFn.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum,
                          bn.eps, relu, _sync_world(bn) if training else 1)
