"""Round-4 kernels and wiring: the grouped weight-gradient launch (hiast_conv_wgrad_group_nhwc, K9d) and the autograd node
that defers the three weight gradients of a bottleneck to one launch (functional._WGroupFn)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ round 4: grouped weight gradients
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 24, 40, 512, 256, 2), (1, 33, 47, 1024, 256, 2), (3, 16, 24, 1024, 512, 4)])
def test_conv_wgrad_group_vs_single_launches_and_fp32(dt, cfg):
    """hiast_conv_wgrad_group_nhwc (the three weight gradients of a bottleneck in one launch + one reduction) against the
    fp32 weight gradient of the same 16-bit operands and against the one-by-one launches; ragged pixel ranges; repeatable"""
    from hiast_amd import kernels as K
    B, H, W, cin, mid, dl = cfg
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1234)
    mk = lambda c: torch.randn(B, H, W, c, generator=g).to(dt).to(dev)
    x1, d1, x2, d2, x3, d3 = mk(cin), mk(mid), mk(mid), mk(mid), mk(mid), mk(cin)
    jobs = [(d3, x3, 1, 1, 1), (d2, x2, 3, 1, dl), (d1, x1, 1, 1, 1)]
    got = K.conv_wgrad_group(jobs)
    again = K.conv_wgrad_group(jobs)
    for (dy, x, k, _, dil), dw, dw2 in zip(jobs, got, again):
        Cout, Cin = dy.shape[3], x.shape[3]
        assert dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, Cin, k, k)
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (Cout, Cin, k, k), dy.permute(0, 3, 1, 2).float(),
                                          stride=1, padding=dil if k == 3 else 0, dilation=dil if k == 3 else 1)
        scale = ref.abs().max().item()
        assert (dw - ref).abs().max().item() <= 1e-4 * scale
        one = K.conv_wgrad_nhwc(dy, x, k, 1, dil)
        assert (dw - one).abs().max().item() <= 2e-5 * scale          # the same products, another summation order
        assert torch.equal(dw, dw2)
    # two jobs (the 1x1 pair of a layer4 block) and the argument checks
    two = K.conv_wgrad_group([jobs[0], jobs[2]])
    assert torch.allclose(two[0], got[0], rtol=0, atol=2e-5 * got[0].abs().max().item())
    with pytest.raises(Exception):
        K.conv_wgrad_group([(mk(64), mk(64), 1, 1, 1), jobs[0]])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_bottleneck_grouped_weight_gradients_match_the_per_convolution_path(dt, monkeypatch):
    """a layer3-shaped identity bottleneck on the 16-bit training path: the grouped weight-gradient node (_WGroupFn) against
    the per-convolution launches (HIAST_NO_WGROUP=1): same outputs and data gradient bit for bit, weight gradients up to the
    summation order; with and without the side stream; gradient accumulation over two backward passes"""
    from hiast_amd import functional as HF
    from hiast_amd.sseg.models.modules.resnet import Bottleneck
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    blk = Bottleneck(1024, 256, 1, 2).to(dev).train()
    x0 = torch.randn(2, 1024, 24, 40, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(2, 1024, 24, 40, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    res = {}
    # (two passes: the second ACCUMULATES into .grad — on the main stream only: with the side stream on, a step must not
    # accumulate, as on the per-convolution path)
    for mode, passes in (("single", 2), ("group", 2), ("single1", 1), ("group_side", 1)):
        monkeypatch.setenv("HIAST_NO_WGROUP", "1" if mode.startswith("single") else "0")
        HF._wgrad_overlap[0] = mode == "group_side"
        blk.zero_grad()
        for _ in range(passes):
            src = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=dt):
                y = blk(src * 1.0)
            y.backward(gy)
        HF.wgrad_stream_join()
        torch.cuda.synchronize()
        res[mode] = (y.detach().clone(), src.grad.clone(), [c.weight.grad.clone() for c in (blk.conv1, blk.conv2, blk.conv3)])
    HF._wgrad_overlap[0] = False
    for mode, ref in (("group", "single"), ("group_side", "single1")):
        assert torch.equal(res[mode][0], res[ref][0]) and torch.equal(res[mode][1], res[ref][1])
        for a, b in zip(res[mode][2], res[ref][2]):
            assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-5 * b.abs().max().item(), mode


def test_layer4_block_groups_only_its_1x1_pair(monkeypatch):
    """a layer4-shaped block: 68 tiles would fill 204 of 256 CUs, so only conv1 + conv3 share a launch; results as above"""
    from hiast_amd import functional as HF
    from hiast_amd.sseg.models.modules.resnet import Bottleneck
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    blk = Bottleneck(2048, 512, 1, 4).to(dev).train()
    x0 = torch.randn(1, 2048, 16, 24, device=dev).half().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(1, 2048, 16, 24, device=dev).half().contiguous(memory_format=torch.channels_last)
    grp, wv = HF.wgroup_weights((blk.conv1, blk.conv2, blk.conv3), x0)
    assert grp is not None and wv[1] is None and grp["slot"] == [0, None, 1]
    res = {}
    for mode in ("single", "group"):
        monkeypatch.setenv("HIAST_NO_WGROUP", "1" if mode == "single" else "0")
        blk.zero_grad()
        src = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = blk(src * 1.0)
        y.backward(gy)
        torch.cuda.synchronize()
        res[mode] = (src.grad.clone(), [c.weight.grad.clone() for c in (blk.conv1, blk.conv2, blk.conv3)])
    assert torch.equal(res["group"][0], res["single"][0])
    for a, b in zip(res["group"][1], res["single"][1]):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
