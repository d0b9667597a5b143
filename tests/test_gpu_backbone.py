"""Fused frozen-BatchNorm + residual + ReLU (csrc/bn_act.hip) against the three torch ops it replaces
(mmdet/models/backbones/resnet.py:240-262 under norm_eval=True)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bn(C, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    bn = torch.nn.BatchNorm2d(C).cuda()
    bn.running_mean.copy_(torch.randn(C, generator=g) * 0.3)
    bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn.weight.data.copy_(torch.rand(C, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(C, generator=g) * 0.2)
    return bn.eval()


@pytest.mark.parametrize('shape', [(2, 6, 7, 9), (2, 64, 40, 64), (1, 3, 1, 1), (3, 16, 33, 4), (2, 512, 100, 168)])
@pytest.mark.parametrize('use_res', [False, True])
@pytest.mark.parametrize('relu', [False, True])
def test_frozen_bn_act_matches_torch(shape, use_res, relu):
    from kgdet_amd.backbone import frozen_bn_act, _FrozenBNAct
    g = torch.Generator(device='cpu').manual_seed(1)
    bn = _bn(shape[1], 2)
    x0 = torch.randn(shape, generator=g).cuda()
    r0 = torch.randn(shape, generator=g).cuda()
    gy = torch.randn(shape, generator=g).cuda()

    def run(fused):
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if use_res else None
        bn.zero_grad()
        if fused:
            y = frozen_bn_act(x, bn, r, relu)
            assert isinstance(y.grad_fn, _FrozenBNAct._backward_cls)   # the HIP op ran, not the torch ops
        else:
            y = bn(x)
            if use_res:
                y = y + r
            if relu:
                y = F.relu(y)
        y.backward(gy)
        return (y.detach(), x.grad, bn.weight.grad.clone(), bn.bias.grad.clone(), r.grad if use_res else None)

    got, want = run(True), run(False)
    names = ['y', 'grad_x', 'grad_gamma', 'grad_beta', 'grad_residual']
    for n, a, b in zip(names, got, want):
        if b is None:
            assert a is None
            continue
        scale = b.abs().max().item() + 1e-12
        # elementwise outputs differ by fma contraction only; the channel sums by summation order
        tol = 2e-6 if n in ('y', 'grad_x', 'grad_residual') else 2e-5
        # a ReLU mask may flip where the pre-activation is within rounding of zero: allow isolated elements
        bad = ((a - b).abs() > tol * scale).float().mean().item()
        assert bad <= (1e-5 if n != 'y' else 0.0) or n in ('grad_gamma', 'grad_beta') and bad == 0, (n, bad)


def test_frozen_bn_act_without_input_grad_and_without_affine_grads():
    from kgdet_amd.backbone import frozen_bn_act
    bn = _bn(8, 3)
    for p in bn.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 8, 5, 12, device='cuda')
    r = torch.randn(2, 8, 5, 12, device='cuda', requires_grad=True)
    y = frozen_bn_act(x, bn, r, True)
    y.sum().backward()
    want = (F.relu(bn(x) + r.detach()) > 0).float()
    torch.testing.assert_close(r.grad, want)
    assert bn.weight.grad is None and bn.bias.grad is None


def test_training_backbone_uses_fused_op_and_matches_module_path():
    """ResNet-50 in train() (norm_eval): outputs and every parameter gradient equal the module-by-module path."""
    from kgdet_amd import backbone as bb
    torch.manual_seed(0)
    net = bb.ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch').cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    net.train()
    x = torch.randn(2, 3, 96, 128, device='cuda')

    def run():
        net.zero_grad()
        outs = net(x)
        sum(o.square().mean() for o in outs).backward()
        return [o.detach() for o in outs], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    outs, grads = run()
    orig = bb.frozen_bn_act

    def unfused(x, bn, residual=None, relu=False):
        out = bn(x)
        if residual is not None:
            out = out + residual
        return F.relu(out, inplace=True) if relu else out

    bb.frozen_bn_act = unfused
    bb.MERGE_CONV_BN = False          # the merged conv + BN node does not go through frozen_bn_act
    bb.FUSE_STEM = False
    try:
        outs_ref, grads_ref = run()
    finally:
        bb.frozen_bn_act = orig
        bb.MERGE_CONV_BN = True
        bb.FUSE_STEM = True
    for a, b in zip(outs, outs_ref):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
    assert grads.keys() == grads_ref.keys() and len(grads) > 100
    # fp32 summation-order noise through ~50 layers (MIOpen's strided kernels use atomics): a few 1e-4 of the scale.  The two
    # paths round BatchNorm differently in the last bit, so now and then a ReLU input that close to zero flips: on the 6 x 8 and
    # 3 x 4 maps of this input one flipped element moves the gradients of its block by up to ~1 % (seen with either 3x3
    # kernel, 1e-6 ... 1e-2 depending on the seed) -- a handful of tensors may sit there, none beyond
    rel = {n: (grads[n] - grads_ref[n]).abs().max().item() / (grads_ref[n].abs().max().item() + 1e-12) for n in grads}
    loose = [n for n in rel if rel[n] > 1e-3]
    assert len(loose) <= 6 and all(rel[n] <= 3e-2 for n in loose), sorted(rel.items(), key=lambda kv: -kv[1])[:8]


@pytest.mark.parametrize('B,K,H,W,relu', [(2, 64, 50, 84, True), (1, 128, 25, 42, True), (3, 256, 13, 21, False), (1, 48, 7, 9, True)])
def test_fused_residual_conv_nhwc_matches_conv_plus_epilogue(B, K, H, W, relu):
    """csrc/conv_nhwc.hip: conv3 + folded bn3 + identity add + ReLU of a bottleneck (resnet.py:240-262) in one bf16
    channels-last kernel, against F.conv2d + the separate epilogue pass it replaces (same rounding sequence: the product is
    rounded to bf16 before bias and residual are added) -- ragged pixel counts, 4- and 8-wave variants, K not a multiple of 64"""
    import ctypes
    from kgdet_amd import backbone, _lib
    N = 4 * K if K % 32 == 0 else 128
    g = torch.Generator('cuda').manual_seed(B * 1000 + K)
    x = torch.randn(B, K, H, W, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, N, H, W, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, K, 1, 1, device='cuda', generator=g) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(N, device='cuda', generator=g)
    out = torch.full_like(r, float('nan'))
    _lib.check(_lib.lib().kgdet_conv1x1_nhwc_residual(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r), _lib.ptr(out),
                                                      ctypes.c_int64(B * H * W), ctypes.c_int32(K), ctypes.c_int32(N),
                                                      ctypes.c_int32(1 if relu else 0), _lib.current_stream()), 'nhwc')
    ref = torch.nn.functional.conv2d(x.float(), w.float()).to(torch.bfloat16).float() + b.view(1, -1, 1, 1) + r.float()
    ref = (torch.relu(ref) if relu else ref).to(torch.bfloat16)
    assert out.is_contiguous(memory_format=torch.channels_last) and not torch.isnan(out.float()).any()
    # one bf16 ulp where the fp32 sums of the two routes round differently
    assert float((out.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())
    assert float((out.float() - ref.float()).abs().mean()) <= 1e-3 * float(ref.float().abs().mean())
    # the same with conv2's epilogue deferred into the activation loads: x raw, relu(x + in_bias) applied on the fly
    ib = torch.randn(K, device='cuda', generator=g)
    out2 = torch.full_like(r, float('nan'))
    _lib.check(_lib.lib().kgdet_conv1x1_nhwc_residual_in(_lib.ptr(x), _lib.ptr(ib), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r),
                                                         _lib.ptr(out2), ctypes.c_int64(B * H * W), ctypes.c_int32(K),
                                                         ctypes.c_int32(N), ctypes.c_int32(1 if relu else 0),
                                                         _lib.current_stream()), 'nhwc_in')
    x2 = torch.relu(x.float() + ib.view(1, -1, 1, 1)).to(torch.bfloat16)
    ref2 = torch.nn.functional.conv2d(x2.float(), w.float()).to(torch.bfloat16).float() + b.view(1, -1, 1, 1) + r.float()
    ref2 = (torch.relu(ref2) if relu else ref2).to(torch.bfloat16)
    assert float((out2.float() - ref2.float()).abs().max()) <= 2 ** -7 * float(ref2.float().abs().max())


def test_folded_batchnorm_step_equals_the_unfolded_step():
    """From a (convolution, BatchNorm) pair's second step inside conv1x1.step_scope on, the frozen-statistics BatchNorm is
    folded into the convolution (backbone._ConvBNActFold: images of w * s, bias t, no BatchNorm pass, grad_gamma from
    <w, grad_w_raw>).  Outputs and every parameter gradient of that step equal the unfolded path's -- with the last
    BatchNorm of every bottleneck zero-initialised as mmdet does (gamma = 0: nothing may divide by it)."""
    from kgdet_amd import backbone as bb, conv1x1
    x = torch.randn(2, 3, 128, 160, device='cuda', generator=torch.Generator('cuda').manual_seed(3))

    def run(fold):
        conv1x1._entries.clear(); conv1x1._fold_entries.clear()
        torch.manual_seed(0)
        net = bb.ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch').cuda()
        for n, m in net.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.1)
                if n.endswith('bn3') and 'layer3' in n:
                    m.weight.data.zero_()          # zero_init_residual
        net.train()
        opt = torch.optim.SGD(net.parameters(), lr=1e-3)
        conv1x1.FOLD_BN = fold
        try:
            for step in range(3):
                opt.zero_grad()
                with conv1x1.step_scope():
                    outs = net(x)
                sum(o.square().mean() for o in outs).backward()
                if step == 2:
                    return ([o.detach().clone() for o in outs],
                            {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
                            len(conv1x1._fold_entries))
                opt.step()
        finally:
            conv1x1.FOLD_BN = True
            conv1x1._entries.clear(); conv1x1._fold_entries.clear()

    outs, grads, n_folded = run(True)
    outs_ref, grads_ref, n_ref = run(False)
    assert n_folded >= 30 and n_ref == 0, (n_folded, n_ref)
    for a, b in zip(outs, outs_ref):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    assert grads.keys() == grads_ref.keys() and len(grads) > 100
    rel = {n: (grads[n] - grads_ref[n]).abs().max().item() / (grads_ref[n].abs().max().item() + 1e-12) for n in grads}
    # (a ReLU input within rounding of zero may flip between the two roundings of BatchNorm: see the test above)
    loose = [n for n in rel if rel[n] > 1e-3]
    assert len(loose) <= 6 and all(rel[n] <= 3e-2 for n in loose), sorted(rel.items(), key=lambda kv: -kv[1])[:8]
    zero_gamma = [n for n in grads if n.endswith('bn3.weight') and 'layer3' in n]
    assert zero_gamma and all(rel[n] <= 1e-3 and grads_ref[n].abs().max().item() > 0 for n in zero_gamma)


@pytest.mark.parametrize('size', [(96, 128), (97, 131), (800, 1344), (30, 33)])
@pytest.mark.parametrize('grad_mode', [True, False])
def test_fused_stem_matches_module_path(size, grad_mode):
    """maxpool(relu(norm1(conv1(x)))) of the frozen stem as conv1 + ONE pass (bn_relu_maxpool) equals the three modules
    (resnet.py:528), odd sizes included; a stem that trains (frozen_stages = -1) keeps the module path."""
    from kgdet_amd import backbone as bb
    torch.manual_seed(1)
    net = bb.ResNet(depth=50, num_stages=1, strides=(1,), dilations=(1,), out_indices=(0,), frozen_stages=0,
                    style='pytorch').cuda()
    net.norm1.running_mean.normal_(0, 0.3)
    net.norm1.running_var.uniform_(0.5, 1.5)
    net.norm1.weight.data.normal_(0, 1.0)          # negative scales too
    net.norm1.bias.data.normal_(0, 0.3)
    net.train()
    x = torch.randn(2, 3, *size, device='cuda')
    with torch.set_grad_enabled(grad_mode):
        got = net._stem(x)
        want = F.max_pool2d(F.relu(net.norm1(net.conv1(x))), 3, 2, 1)
    assert got.shape == want.shape and not got.requires_grad
    # (conv1 itself runs on the split-bf16 stem kernel: 5e-6 of the scale against MIOpen's fp32 result)
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    net2 = bb.ResNet(depth=50, num_stages=1, strides=(1,), dilations=(1,), out_indices=(0,), frozen_stages=-1,
                     style='pytorch').cuda()
    net2.train()
    y = net2._stem(x)
    assert y.requires_grad       # a trainable stem stays on autograd-recording modules


@pytest.mark.parametrize('B,H,W', [(2, 96, 128), (1, 97, 131), (2, 800, 1344), (3, 7, 9), (1, 15, 16)])
def test_stem_conv_kernel_matches_fp64(B, H, W):
    """conv1 of the stem (7x7, stride 2, padding 3, 3 -> 64) on stem_conv7x7_s2 against the fp64 convolution: split-bf16
    accuracy (2e-5 of the output scale), odd sizes, images smaller than a tile"""
    from kgdet_amd import backbone as bb
    torch.manual_seed(H)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    x = torch.randn(B, 3, H, W, device='cuda') * 2
    with torch.no_grad():
        y = bb._stem_conv(conv, x)
        ref = F.conv2d(x.double(), conv.weight.double(), stride=2, padding=3)
    assert y.shape == ref.shape and torch.isfinite(y).all()
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 2e-5     # (147 products per output: the maximum over 4e5 outputs sits at ~1e-5)
    with torch.no_grad():
        conv.weight.mul_(2.0)           # the cached pack follows the weight (an in-place update bumps its version)
        y2 = bb._stem_conv(conv, x)
    assert ((y2.double() - 2 * ref).abs().max() / ref.abs().max()).item() < 4e-5


@pytest.mark.parametrize('size', [(96, 128), (97, 131), (800, 1344)])
def test_fused_stem_inference_bf16_channels_last(size):
    """inference under bf16 autocast: conv1 (BatchNorm folded, channels-last) + ONE pass for bias, ReLU and the max pooling
    (bias_relu_maxpool_nhwc) gives the same tensor, bit for bit, as the in-place epilogue followed by nn.MaxPool2d"""
    from kgdet_amd import backbone as bb
    torch.manual_seed(2)
    net = bb.ResNet(depth=50, num_stages=1, strides=(1,), dilations=(1,), out_indices=(0,), frozen_stages=0,
                    style='pytorch').cuda()
    net.norm1.running_mean.normal_(0, 0.3)
    net.norm1.running_var.uniform_(0.5, 1.5)
    net.norm1.weight.data.normal_(0, 1.0)
    net.norm1.bias.data.normal_(0, 0.3)
    net.eval()
    x = torch.randn(2, 3, *size, device='cuda')
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        got = net._stem(x)
        bb.FUSE_STEM = False
        try:
            want = net._stem(x)
        finally:
            bb.FUSE_STEM = True
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


@pytest.mark.parametrize('modulated', [False, True])
@pytest.mark.parametrize('stride', [1, 2])
def test_bottleneck_dcn_option_matches_torch_reference(modulated, stride):
    """SURVEY 8f row 4 (resnet.py:162-186, 231-238): conv2 replaced by conv2_offset + DeformConv /
    ModulatedDeformConv.  Forward and all gradients against the grid_sample formulation (tests/torch_ref.py)."""
    from kgdet_amd import backbone as bb
    from tests import torch_ref
    torch.manual_seed(3)
    down = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, stride=stride, bias=False), torch.nn.BatchNorm2d(128))
    blk = bb.Bottleneck(64, 32, stride=stride, downsample=down,
                        dcn=dict(modulated=modulated, deformable_groups=1, fallback_on_stride=False)).cuda()
    assert type(blk.conv2).__name__ == ('ModulatedDeformConv' if modulated else 'DeformConv')
    assert blk.conv2_offset.out_channels == (27 if modulated else 18)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    blk.eval()            # norm_eval
    blk.conv2_offset.weight.data.normal_(0, 0.05)
    x0 = torch.randn(2, 64, 20, 28, device='cuda')

    def reference(x):
        out = blk.relu(blk.norm1(blk.conv1(x)))
        om = blk.conv2_offset(out)
        if modulated:
            off, mask = om[:, :18], om[:, -9:].sigmoid()
        else:
            off, mask = om, None
        out = torch_ref.deform_conv(out, off, blk.conv2.weight, stride=stride, padding=1, dilation=1, mask=mask)
        if modulated and blk.conv2.bias is not None:
            out = out + blk.conv2.bias.view(1, -1, 1, 1)
        out = blk.relu(blk.norm2(out))
        out = blk.norm3(blk.conv3(out))
        return blk.relu(out + blk.downsample(x))

    def run(fn):
        blk.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = fn(x)
        y.square().mean().backward()
        return y.detach(), x.grad, {n: p.grad.clone() for n, p in blk.named_parameters() if p.grad is not None}

    y, gx, gp = run(blk)
    yr, gxr, gpr = run(reference)
    assert y.shape == (2, 128, 20 // stride, 28 // stride)
    assert (y - yr).abs().max().item() <= 1e-4 * yr.abs().max().item()
    assert (gx - gxr).abs().max().item() <= 1e-3 * gxr.abs().max().item()
    assert gp.keys() == gpr.keys() and 'conv2_offset.weight' in gp
    for n in gp:
        assert (gp[n] - gpr[n]).abs().max().item() <= 1e-3 * (gpr[n].abs().max().item() + 1e-12), n


def test_resnet_with_dcn_stages_builds_and_steps():
    from kgdet_amd import backbone as bb
    torch.manual_seed(0)
    net = bb.ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch',
                    dcn=dict(modulated=False, deformable_groups=1, fallback_on_stride=False),
                    stage_with_dcn=(False, True, True, True)).cuda()
    net.train()
    outs = net(torch.randn(1, 3, 128, 160, device='cuda'))
    assert [o.shape[1] for o in outs] == [256, 512, 1024, 2048]
    sum(o.mean() for o in outs).backward()
    assert net.layer3[0].conv2.weight.grad.abs().sum() > 0 and net.layer3[0].conv2_offset.weight.grad is not None
    assert not hasattr(net.layer1[0], 'conv2_offset')


@pytest.mark.parametrize('k', [1, 3])
@pytest.mark.parametrize('B,C,O,H,W', [(2, 64, 256, 20, 34), (1, 256, 64, 9, 14), (2, 512, 128, 25, 42), (2, 128, 512, 13, 10),
                                       (3, 1024, 256, 6, 8), (2, 2048, 512, 25, 42), (2, 16, 48, 5, 6), (2, 32, 16, 129, 2),
                                       (2, 128, 128, 30, 44), (2, 256, 128, 9, 84), (1, 128, 256, 7, 4),
                                       (2, 512, 128, 100, 168), (2, 64, 2048, 25, 42), (2, 128, 128, 100, 168)])
def test_conv1x1_split_matches_fp64_convolution(B, C, O, H, W, k, monkeypatch):
    """csrc/conv1x1.hip (bf16 hi/lo-split MFMA GEMMs): forward, grad_input and grad_weight against the fp64
    convolution, to fp32-level accuracy (1e-5 of the result's scale; MIOpen's fp32 kernels sit at ~1e-6); ragged
    pixel tiles, M < 128, the K-split and the 8- and 16-byte load variants of grad_weight; the last three shapes take the
    160-pixel tiles (264 / 288 tiles of 128 pixels: conv_nn<1, 5>, conv3x3_patch4<4, 5>)."""
    from kgdet_amd import conv1x1 as c1
    monkeypatch.setattr(c1, 'SPLIT_GRAD_WEIGHT_3X3', True)     # exercise the 3x3 grad_weight kernel too
    g = torch.Generator(device='cpu').manual_seed(C + O + H)
    x = torch.randn(B, C, H, W, generator=g).cuda().requires_grad_()
    if k == 3 and C * O > 512 * 512:
        pytest.skip('3x3 at 2048 x 512 channels: not a backbone shape')
    w = (torch.randn(O, C, k, k, generator=g) * 0.1).cuda().requires_grad_()
    gy = torch.randn(B, O, H, W, generator=g).cuda()
    assert c1.applicable(x, w, padding=(k // 2, k // 2))
    y = c1.conv1x1(x, w)
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    yd = F.conv2d(xd, wd, padding=k // 2)
    yd.backward(gy.double())
    for name, a, b in (('y', y, yd), ('grad_x', x.grad, xd.grad), ('grad_w', w.grad, wd.grad)):
        err = (a.double() - b).abs().max().item() / b.abs().max().item()
        assert err < 1e-5, (name, err)
    # deterministic: same bits on a second run
    x.grad = None; w.grad = None
    y2 = c1.conv1x1(x, w)
    y2.backward(gy)
    assert torch.equal(y, y2)
    x2g, w2g = x.grad.clone(), w.grad.clone()
    x.grad = None; w.grad = None
    c1.conv1x1(x, w).backward(gy)
    assert torch.equal(x.grad, x2g)
    if k == 1 or C % 128 == 0:     # otherwise grad_weight is MIOpen's (kgdet_amd/conv1x1.py), which makes no such promise
        assert torch.equal(w.grad, w2g)


@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('B,C,O,H,W,k', [(2, 256, 256, 25, 42, 3), (2, 256, 256, 100, 168, 3), (1, 64, 32, 9, 14, 1)])
def test_conv_bias_act_matches_fp64(B, C, O, H, W, k, relu):
    """conv1x1.conv_bias_act: the biased plain convolutions of the head's first stage (KP3:69-71, 119-120) -- bias and ReLU in
    the store of the split-bf16 kernel; output and the three gradients against the fp64 modules."""
    from kgdet_amd import conv1x1 as c1
    torch.manual_seed(C + H)
    conv = torch.nn.Conv2d(C, O, k, 1, k // 2).cuda()
    conv.bias.data.normal_(0, 0.5)
    x = torch.randn(B, C, H, W, device='cuda', requires_grad=True)
    gy = torch.randn(B, O, H, W, device='cuda')
    y = c1.conv_bias_act(conv, x, relu=relu)
    assert type(y.grad_fn).__name__ == '_ConvBiasActBackward'
    y.backward(gy)
    conv_d = torch.nn.Conv2d(C, O, k, 1, k // 2).cuda().double()
    conv_d.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    xd = x.detach().double().requires_grad_()
    yd = conv_d(xd)
    # the ReLU mask of the fp32 result (an element within rounding of zero may flip in fp64)
    yd = yd * (y.detach() > 0) if relu else yd
    yd.backward(gy.double())
    for name, a_, b_ in (('y', y, yd), ('grad_x', x.grad, xd.grad), ('grad_w', conv.weight.grad, conv_d.weight.grad),
                         ('grad_b', conv.bias.grad, conv_d.bias.grad)):
        err = (a_.double() - b_).abs().max().item() / b_.abs().max().item()
        assert err < 1e-5, (name, err)
    # not applicable (CPU / no_grad): the module itself
    with torch.no_grad():
        y0 = c1.conv_bias_act(conv, x, relu=relu)
    assert (y0 - y).abs().max().item() <= 1e-4 * y.abs().max().item()


@pytest.mark.parametrize('B,C,O,H,W', [(2, 768, 13, 25, 42), (2, 768, 588, 25, 42), (2, 588, 166, 25, 42), (2, 256, 13, 100, 168),
                                       (1, 588, 166, 13, 21), (2, 40, 24, 8, 8), (2, 7, 6, 6, 6), (2, 256, 588, 50, 84), (3, 17, 130, 9, 10)])
def test_ragged_1x1_convolutions_match_fp64(B, C, O, H, W):
    """Round 6: the head's 13- / 588- / 166-channel output convolutions on the split MFMA kernels.  Channel counts that are not
    multiples of 16 on either side: the operand images pad the reduction with zeros, conv_nn re-reads the last channel for the
    padding, the weight-gradient kernels take partial tiles.  Output and all three gradients against the fp64 module, bit-repeatably
    (the vendor GEMMs these replaced made no such promise), inside a step scope too (second step: the one-launch pack)."""
    from kgdet_amd import conv1x1 as c1
    torch.manual_seed(C + O + H)
    conv = torch.nn.Conv2d(C, O, 1).cuda()
    conv.bias.data.normal_(0, 0.5)
    x = torch.randn(B, C, H, W, device='cuda', requires_grad=True)
    gy = torch.randn(B, O, H, W, device='cuda')
    conv_d = torch.nn.Conv2d(C, O, 1).cuda().double()
    conv_d.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    xd = x.detach().double().requires_grad_()
    yd = conv_d(xd)
    yd.backward(gy.double())
    runs = []
    c1._entries.clear()
    try:
        for scoped in (False, True, True):
            x.grad = None; conv.weight.grad = None; conv.bias.grad = None
            if scoped:
                with c1.step_scope():
                    y = c1.conv_bias_act(conv, x)
            else:
                y = c1.conv_bias_act(conv, x)
            assert type(y.grad_fn).__name__ == '_ConvBiasActBackward'
            y.backward(gy)
            runs.append((y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
            for name, a_, b_ in (('y', y, yd), ('grad_x', x.grad, xd.grad), ('grad_w', conv.weight.grad, conv_d.weight.grad),
                                 ('grad_b', conv.bias.grad, conv_d.bias.grad)):
                err = (a_.double() - b_).abs().max().item() / b_.abs().max().item()
                assert err < 1e-5, (name, err, scoped)
    finally:
        c1._entries.clear()
    for later in runs[1:]:
        for a_, b_ in zip(runs[0][:3], later[:3]):
            assert torch.equal(a_, b_)


def test_dense_conv_kernels_random_shapes_against_fp64():
    """60 random problems through the dense convolution kernels (1x1, 3x3 stride 1 on the patch kernels, 3x3 stride 2 with its
    parity-class grad_input): tiny and ragged maps, one-chunk reductions, row counts around the 32 / 64 / 128-row tile edges,
    K splits with uneven parts -- output, grad_input and grad_weight against the fp64 convolution"""
    from kgdet_amd import conv1x1 as c1
    rng = np.random.RandomState(7)
    done = 0
    for trial in range(200):
        if done == 60:
            break
        k = int(rng.choice([1, 3, 3]))
        stride = 2 if (k == 3 and rng.rand() < 0.3) else 1
        B = int(rng.randint(1, 4))
        C = 16 * int(rng.randint(1, 18))
        O = 16 * int(rng.randint(1, 18))
        H, W = int(rng.randint(1, 41)), int(rng.randint(1, 41))
        x = torch.randn(B, C, H, W, device='cuda', requires_grad=True)
        w = (torch.randn(O, C, k, k, device='cuda') * 0.1).requires_grad_()
        if stride == 1:
            if not c1.applicable(x, w, padding=(k // 2, k // 2)):
                continue
            y = c1.conv_split(x, w)
        else:
            if not c1.applicable_stride2(x, w, (2, 2), (1, 1), (1, 1), 1):
                continue
            y = c1.conv3x3_stride2(x, w)
        gy = torch.randn_like(y)
        y.backward(gy)
        xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
        yd = F.conv2d(xd, wd, stride=stride, padding=k // 2)
        yd.backward(gy.double())
        for name, a_, b_ in (('y', y, yd), ('grad_x', x.grad, xd.grad), ('grad_w', w.grad, wd.grad)):
            scale = b_.abs().max().item()
            err = (a_.double() - b_).abs().max().item() / max(scale, 1e-30)
            tol = 1e-4 if (name == 'grad_w' and stride == 2) else 2e-5     # (stride-2 grad_weight: MIOpen fp32)
            assert err < tol and torch.isfinite(a_).all(), (name, err, (B, C, O, H, W, k, stride))
        done += 1
    assert done == 60


def test_conv1x1_not_applicable_cases_fall_back():
    from kgdet_amd import conv1x1 as c1
    x = torch.randn(2, 64, 8, 8, device='cuda')
    w = torch.randn(32, 64, 1, 1, device='cuda')
    assert c1.applicable(x, w) and not c1.applicable(x, w, stride=(2, 2)) and not c1.applicable(x.half(), w.half())
    w3 = torch.randn(32, 64, 3, 3, device='cuda')
    assert c1.applicable(x, w3, padding=(1, 1)) and not c1.applicable(x, w3) and not c1.applicable(x, w3, padding=(1, 1), dilation=(2, 2))
    assert not c1.applicable(x, w3, stride=(2, 2), padding=(1, 1)) and not c1.applicable(x, torch.randn(32, 64, 5, 5, device='cuda'), padding=(2, 2))
    assert c1.applicable(torch.randn(2, 40, 8, 8, device='cuda'), torch.randn(32, 40, 1, 1, device='cuda')) == c1.RAGGED_1X1
    assert not c1.applicable(torch.randn(2, 40, 8, 8, device='cuda'), torch.randn(32, 40, 3, 3, device='cuda'), padding=(1, 1))
    assert not c1.applicable(torch.randn(2, 7, 8, 8, device='cuda'), torch.randn(5, 7, 1, 1, device='cuda'))     # (the weight-gradient sum handles pairs)
    assert c1.applicable(torch.randn(2, 64, 3, 3, device='cuda'), w) == c1.DIRECT_ODD_MAPS     # odd H*W: in place since round 4
    assert not c1.applicable(torch.randn(2, 64, 1, 3, device='cuda'), w)                       # fewer than 4 pixels
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert not c1.applicable(x, w)


@pytest.mark.parametrize('H,W', [(13, 21), (7, 11), (5, 5), (3, 3), (2, 2), (25, 42), (9, 6)])
@pytest.mark.parametrize('k', [1, 3])
def test_split_convolutions_take_odd_and_unaligned_maps_in_place(H, W, k):
    """csrc/conv1x1.hip on maps whose pixel count is odd / whose rows are not a multiple of 4 floats (the two coarsest levels of a
    five-level head, 13 x 21 and 7 x 11; the 25 x 42 head maps): forward, grad_input and grad_weight (conv_ntp: a tap is a shift
    of a 4-byte aligned 16-byte load, ragged last pieces moved back into place) against torch's float64 convolution, and the
    biased convolution + ReLU of config 5's FPN.  Rounds 2-3 appended a zero column around every such convolution."""
    from kgdet_amd import conv1x1 as c1
    g = torch.Generator(device='cpu').manual_seed(H * 100 + W * 3 + k)
    x = torch.randn(2, 64, H, W, generator=g).cuda().requires_grad_()
    w = (torch.randn(128, 64, k, k, generator=g) * 0.1).cuda().requires_grad_()
    gy = torch.randn(2, 128, H, W, generator=g).cuda()
    assert c1.applicable(x, w, (1, 1), (k // 2, k // 2))
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    gx_r, gw_r = torch.autograd.grad(ref, (x, w), gy.double())
    y = c1.conv_split(x, w)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    for a, b in ((y, ref), (gx, gx_r), (gw, gw_r)):
        assert float((a.double() - b).abs().max()) < 2e-5 * float(b.abs().max())
    if k == 3:
        conv = torch.nn.Conv2d(64, 128, 3, padding=1).cuda()
        ref = F.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1))
        r = torch.autograd.grad(ref, (x, conv.weight, conv.bias), gy.double())
        o = torch.autograd.grad(c1.conv_bias_act(conv, x, relu=True), (x, conv.weight, conv.bias), gy)
        for a, b in zip(o, r):
            assert float((a.double() - b).abs().max()) < 2e-5 * float(b.abs().max())


@pytest.mark.parametrize('B,C,O,H,W', [(2, 128, 128, 20, 36), (1, 64, 256, 17, 13), (2, 256, 256, 50, 84), (1, 64, 64, 9, 12),
                                       (2, 128, 128, 200, 336), (2, 512, 512, 50, 84), (3, 32, 48, 7, 20), (1, 160, 16, 33, 11),
                                       (2, 256, 256, 25, 42), (2, 256, 256, 13, 21), (1, 64, 32, 5, 5)])     # (odd output maps: 13 x 21, 7 x 11, 3 x 3)
def test_conv3x3_stride2_forward_matches_fp64(B, C, O, H, W):
    """conv_nn<9> with stride 2 (padding 1): forward against the fp64 convolution; grad_input on conv3x3_s2_grad_input (the four
    parity classes of the output pixels; odd input sizes, ragged tiles, C not a multiple of 128) against fp64 to 1e-5;
    grad_weight = MIOpen's, same as F.conv2d"""
    from kgdet_amd import conv1x1 as c1
    g = torch.Generator(device='cpu').manual_seed(H * W)
    x = torch.randn(B, C, H, W, generator=g).cuda().requires_grad_()
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.1).cuda().requires_grad_()
    if not c1.applicable_stride2(x, w, (2, 2), (1, 1), (1, 1), 1):
        pytest.skip('odd number of output pixels')
    y = c1.conv3x3_stride2(x, w)
    assert torch.equal(y, c1.conv3x3_stride2(x, w)), 'forward must be deterministic'
    ref = F.conv2d(x.detach().double(), w.detach().double(), stride=2, padding=1)
    assert y.shape == ref.shape
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    F.conv2d(xr, wr, stride=2, padding=1).backward(gy)
    assert (x.grad - xr.grad).abs().max().item() <= 1e-4 * xr.grad.abs().max().item()
    assert (w.grad - wr.grad).abs().max().item() <= 1e-4 * wr.grad.abs().max().item()
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    F.conv2d(xd, wd, stride=2, padding=1).backward(gy.double())
    assert ((x.grad.double() - xd.grad).abs().max() / xd.grad.abs().max()).item() < 1e-5
    # (round 6: grad_weight on the gather + 1x1 weight-gradient GEMM, csrc/conv1x1.hip kgdet_conv3x3_s2_grad_weight)
    assert ((w.grad.double() - wd.grad).abs().max() / wd.grad.abs().max()).item() < 1e-5
    assert torch.isfinite(x.grad).all() and torch.isfinite(w.grad).all()
    x.grad = None; w.grad = None
    c1.conv3x3_stride2(x, w).backward(gy)          # deterministic
    g1, gw1 = x.grad.clone(), w.grad.clone()
    x.grad = None; w.grad = None
    c1.conv3x3_stride2(x, w).backward(gy)
    assert torch.equal(g1, x.grad) and torch.equal(gw1, w.grad)


@pytest.mark.parametrize('B,K,N,H,W,res,relu', [(2, 64, 64, 50, 84, False, True), (1, 256, 64, 25, 42, False, True), (2, 256, 128, 13, 21, False, True),
                                               (1, 512, 2048, 25, 42, True, True), (2, 512, 128, 9, 7, False, False), (1, 512, 256, 13, 21, True, False)])
def test_fused_conv_nhwc_without_residual_narrow_tiles_and_long_reductions(B, K, N, H, W, res, relu):
    """csrc/conv_nhwc.hip beyond the bottleneck's conv3: no residual (conv1 + bn1 + ReLU), N = 64 (layer 1: half a channel tile),
    K = 512 (layer 4's conv3: the four-wave variant with the 128 KB weight tile), with and without the deferred input epilogue --
    against F.conv2d + the separate pass, same rounding sequence."""
    import ctypes
    from kgdet_amd import _lib
    g = torch.Generator('cuda').manual_seed(B * 1000 + K + N)
    x = torch.randn(B, K, H, W, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, N, H, W, device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    w = (torch.randn(N, K, 1, 1, device='cuda', generator=g) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(N, device='cuda', generator=g)
    ib = torch.randn(K, device='cuda', generator=g)
    for in_bias in (None, ib):
        out = torch.full((B, N, H, W), float('nan'), device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        _lib.check(_lib.lib().kgdet_conv1x1_nhwc_residual_in(_lib.ptr(x), _lib.ptr(in_bias), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r),
                                                             _lib.ptr(out), ctypes.c_int64(B * H * W), ctypes.c_int32(K),
                                                             ctypes.c_int32(N), ctypes.c_int32(1 if relu else 0),
                                                             _lib.current_stream()), 'nhwc')
        xin = x if in_bias is None else torch.relu(x.float() + ib.view(1, -1, 1, 1)).to(torch.bfloat16)
        ref = F.conv2d(xin.float(), w.float()).to(torch.bfloat16).float() + b.view(1, -1, 1, 1) + (r.float() if res else 0)
        ref = (torch.relu(ref) if relu else ref).to(torch.bfloat16)
        assert not torch.isnan(out.float()).any()
        assert float((out.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())
        assert float((out.float() - ref.float()).abs().mean()) <= 1e-3 * float(ref.float().abs().mean())


def test_bf16_inference_backbone_with_raw_branches_matches_separate_passes():
    """bf16 inference: the stride-2 downsample branch and the stride-2 conv2 of a layer's first bottleneck stay RAW (their folded
    biases join conv3's epilogue / ride on conv3's activation loads) -- same features as with their own bias passes, to bf16
    rounding (one rounding fewer on the raw route)."""
    from kgdet_amd import backbone as bb
    torch.manual_seed(0)
    net = bb.ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch').cuda()
    net.eval()
    for n, m in net.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 3, 256, 320, device='cuda', generator=torch.Generator('cuda').manual_seed(7))

    def run(raw):
        bb.RAW_BRANCHES = raw
        bb.clear_fold_cache(); bb._gemm_choice.clear()
        try:
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                for _ in range(3):        # (the measured route choices settle in the first calls)
                    outs = net(x)
            return [o.float() for o in outs]
        finally:
            bb.RAW_BRANCHES = True
    a, b = run(True), run(False)
    with torch.no_grad():
        ref = [o.float() for o in net(x)]     # fp32
    for u, v, r in zip(a, b, ref):
        scale = r.abs().max().item()
        assert (u - v).abs().max().item() <= 0.05 * scale
        # both routes are equally far from the fp32 features
        eu, ev = (u - r).abs().mean().item(), (v - r).abs().mean().item()
        assert eu <= 1.15 * ev + 1e-6 * scale, (eu, ev)


@pytest.mark.parametrize('k,stride', [(1, 1), (3, 1), (3, 2), (1, 2)])
@pytest.mark.parametrize('B,C,O,H,W', [(2, 64, 128, 20, 34), (1, 512, 64, 6, 8), (2, 128, 256, 13, 10)])
def test_conv_apply_epilogue_flags(B, C, O, H, W, k, stride):
    """kgdet_conv_apply_epilogue: y = [relu](conv(x) + bias[m] [+ residual]) for every flag combination, on the direct
    store path and on the K-split path (small maps), 1x1 / 3x3, stride 1 / 2 -- against torch in fp64."""
    from kgdet_amd import conv1x1 as c1
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    if (Ho * Wo) % 2:
        pytest.skip('odd number of output pixels')
    g = torch.Generator(device='cpu').manual_seed(B * C + H)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.randn(O, C, k, k, generator=g) * 0.1).cuda()
    bias = torch.randn(O, generator=g).cuda()
    res = torch.randn(B, O, Ho, Wo, generator=g).cuda()
    img = c1._pack(w, False)
    ref0 = F.conv2d(x.double(), w.double(), stride=stride, padding=k // 2)
    for use_b in (False, True):
        for use_r in (False, True):
            for relu in (False, True):
                want = ref0 + (bias.double().view(1, -1, 1, 1) if use_b else 0) + (res.double() if use_r else 0)
                want = want.clamp(min=0) if relu else want
                got = c1._apply(img, x, O, k * k, stride, bias if use_b else None, res if use_r else None, relu)
                assert got.shape == want.shape
                assert ((got.double() - want).abs().max() / ref0.abs().max()).item() < 1e-5, (use_b, use_r, relu)


@pytest.mark.parametrize('k', [1, 3])
@pytest.mark.parametrize('B,C,O,H,W', [(2, 64, 128, 20, 34), (1, 512, 64, 6, 8), (2, 128, 256, 13, 10), (2, 128, 128, 100, 168),
                                       (2, 256, 64, 50, 84)])
def test_conv_apply_gate_is_the_mask_of_the_ungated_result(B, C, O, H, W, k):
    """kgdet_conv_apply_gated_fmt: y = [gate > 0] * (conv + residual), bit for bit the ungated result with the mask applied
    afterwards -- direct-store kernels (conv_nn, conv3x3_patch4: large maps) and the K-split sum (small maps)."""
    from kgdet_amd import conv1x1 as c1
    g = torch.Generator(device='cpu').manual_seed(B * C + H + k)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.randn(O, C, k, k, generator=g) * 0.1).cuda()
    res = torch.randn(B, O, H, W, generator=g).cuda()
    gate = torch.randn(B, O, H, W, generator=g).clamp(min=0).cuda()        # a ReLU output: half zeros
    for img in (c1._pack(w, False), ):
        for use_r in (False, True):
            plain = c1._apply(img, x, O, k * k, 1, None, res if use_r else None, False)
            gated = c1._apply(img, x, O, k * k, 1, None, res if use_r else None, False, gate=gate)
            assert torch.equal(gated, torch.where(gate > 0, plain, torch.zeros_like(plain))), (use_r,)
    with pytest.raises(ValueError):
        c1._apply(img, x, O, k * k, gate=gate[:, :1])


@pytest.mark.gpu
def test_gate_fusion_step_equals_the_separate_mask_passes():
    """backbone._GateLink: with the ReLU backward of conv1 / conv2 / the block output applied in the store of the NEXT
    convolution's grad_input kernel, a ResNet-50 training step returns bit-identical outputs and parameter gradients, and
    the masking pass (relu_sum_bwd<true>) is taken only where a ReLU output has more than one consumer."""
    from kgdet_amd import backbone as bb, conv1x1
    x = torch.randn(2, 3, 128, 160, device='cuda', generator=torch.Generator('cuda').manual_seed(5))

    def run(fuse):
        conv1x1._entries.clear(); conv1x1._fold_entries.clear()
        torch.manual_seed(0)
        net = bb.ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch').cuda()
        for n, m in net.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
        net.train()
        opt = torch.optim.SGD(net.parameters(), lr=1e-3)
        bb.GATE_FUSION = fuse
        masked = [0, 0]
        real = bb._bn_fold_lib().kgdet_bn_fold_backward

        class Spy(object):            # counts the masking / sum-only calls of the last step
            def __call__(self, gz, z, relu, *rest):
                masked[1 if relu else 0] += 1
                return real(gz, z, relu, *rest)
        try:
            for step in range(3):
                opt.zero_grad()
                with conv1x1.step_scope():
                    outs = net(x)
                if step == 2:
                    bb._bn_fold_lib().kgdet_bn_fold_backward = Spy()
                sum(o.square().mean() for o in outs).backward()
                if step == 2:
                    return ([o.detach().clone() for o in outs],
                            {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}, masked)
                opt.step()
        finally:
            bb._bn_fold_lib().kgdet_bn_fold_backward = real
            bb.GATE_FUSION = True
            conv1x1._entries.clear(); conv1x1._fold_entries.clear()

    outs, grads, calls = run(True)
    outs_ref, grads_ref, calls_ref = run(False)
    for a, b in zip(outs, outs_ref):
        assert torch.equal(a, b)
    assert grads.keys() == grads_ref.keys() and len(grads) > 100
    for n in grads:
        if grads[n].dim() == 4:       # convolution weights: the same kernels on the same values
            assert torch.equal(grads[n], grads_ref[n]), n
        else:                         # BatchNorm gamma / beta: the per-channel sums are formed in another (fixed) order
            scale = grads_ref[n].abs().max().item() + 1e-30
            assert (grads[n] - grads_ref[n]).abs().max().item() <= 2e-5 * scale, n
    # layers 2-4 train (13 bottlenecks): 39 ReLU nodes, 36 of them folded nodes (the three stride-2 conv2 are not) + 3 downsample
    # nodes without ReLU.  Fused: what still masks are the three layer outputs (several consumers) and conv1 of each layer's first
    # block (its consumer is the stride-2 convolution); nobody takes a sum-only pass: the weight-gradient kernels sum the rows
    assert calls_ref == [3, 36] and calls == [0, 6], (calls, calls_ref)


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,O,H,W,k', [(2, 128, 512, 20, 36, 1), (2, 512, 128, 25, 42, 1), (1, 64, 256, 13, 21, 1), (2, 256, 64, 100, 168, 1),
                                         (2, 128, 128, 20, 36, 3), (2, 256, 256, 25, 42, 3), (1, 128, 128, 13, 21, 3),
                                         (2, 128, 128, 100, 168, 3), (3, 128, 200, 9, 12, 1)])
def test_weight_gradient_kernels_sum_the_rows_of_grad_y(B, C, O, H, W, k):
    """kgdet_conv*_grad_weight_fold with bn_partial == NULL: grad_beta / grad_gamma from row sums formed inside the
    weight-gradient kernel (conv_nt8 / conv_ntp, aligned and ragged maps) equal those from kgdet_bn_fold_backward's partials,
    the weight gradient itself is bit-identical, and both match float64."""
    from kgdet_amd import backbone as bb, conv1x1 as c1, _lib
    g = torch.Generator(device='cpu').manual_seed(B * C + H + k)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    gy = torch.randn(B, O, H, W, generator=g).cuda()
    w = (torch.randn(O, C, k, k, generator=g) * 0.1).cuda()
    s = (torch.rand(O, generator=g) + 0.5).cuda()
    mean, var = torch.randn(O, generator=g).cuda(), (torch.rand(O, generator=g) + 0.5).cuda()
    assert c1.grad_weight_fold_route(x, w) == k
    L = bb._bn_fold_lib()
    P = L.kgdet_bn_act_partials(B, O, H * W)
    partial = torch.empty((O, max(P, 1)), device='cuda')
    _lib.check(L.kgdet_bn_fold_backward(gy.data_ptr(), None, 0, None, partial.data_ptr(), B, O, H * W, _lib.raw_stream(0)), 'sum')
    gw_a, sums_a = c1.grad_weight_fold(x, w, gy, s, mean, var, 1e-5, partial, max(P, 1))
    gw_b, sums_b = c1.grad_weight_fold(x, w, gy, s, mean, var, 1e-5, None, 0)
    assert torch.equal(gw_a, gw_b)
    beta = gy.double().sum((0, 2, 3))
    G = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=k // 2)
    gamma = ((G * w.double()).sum((1, 2, 3)) - mean.double() * beta) / (var.double() + 1e-5).sqrt()
    for got in (sums_a, sums_b):
        assert (got[0].double() - beta).abs().max().item() <= 1e-5 * beta.abs().max().item()
        assert (got[1].double() - gamma).abs().max().item() <= 2e-4 * gamma.abs().max().item()
    assert (gw_b.double() - G * s.double().view(-1, 1, 1, 1)).abs().max().item() <= 2e-4 * G.abs().max().item() * 1.5


@pytest.mark.gpu
def test_multi_pack_images_equal_single_packs_bit_for_bit():
    """the step's one-launch pack of all weights (conv1x1_pack_multi; 3x3 forward images: one thread per 72 contiguous floats)
    writes the same operand images, forward and transposed, as kgdet_conv_pack_both per weight -- 1x1 and 3x3, row counts below,
    at and above one 128-row tile"""
    from kgdet_amd import conv1x1
    torch.manual_seed(1)
    shapes = [(64, 32, 1), (64, 32, 3), (256, 128, 3), (128, 512, 1), (144, 48, 3), (16, 16, 1), (512, 256, 1), (272, 160, 3),
              (13, 768, 1), (588, 768, 1), (166, 588, 1), (5, 7, 1)]      # (round 6: reductions that end inside a 16-channel chunk)
    ws = [torch.nn.Parameter(torch.randn(O, C, k, k, device='cuda')) for O, C, k in shapes]
    x = [torch.randn(1, C, 6, 8, device='cuda', requires_grad=True) for O, C, k in shapes]
    for _ in range(2):          # first scope: every weight joins the set (packed on its own); second: ONE launch for all
        with conv1x1.step_scope():
            for xi, w in zip(x, ws):
                conv1x1.conv_split(xi, w)
    for w in ws:
        e = conv1x1._entries[id(w)]
        img, img_t = conv1x1._pack_both(w.detach())
        assert torch.equal(e.img.view(torch.uint8), img.view(torch.uint8)), tuple(w.shape)
        assert torch.equal(e.img_t.view(torch.uint8), img_t.view(torch.uint8)), tuple(w.shape)
    conv1x1._entries.clear()


def test_step_scope_packs_follow_weight_updates():
    """conv1x1.step_scope: one multi-weight pack launch per training forward into persistent images.  The images must
    follow in-place weight updates (fused optimizers do not bump `_version`), a weight met for the first time inside
    a scope must work, and outside a scope nothing may be served from the persistent images."""
    import torch.nn.functional as F
    from kgdet_amd import conv1x1
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(64, 32, k, k, device='cuda') * 0.1) for k in (1, 3, 1)]
    x = torch.randn(2, 32, 12, 16, device='cuda', requires_grad=True)
    opt = torch.optim.Adam(ws, lr=0.05, fused=True)

    def run(active):
        outs = []
        for w in ws[:active]:
            outs.append(conv1x1.conv_split(x, w))
        return outs

    for step in range(4):
        n_active = 2 if step < 2 else 3          # the third weight joins the set at step 2
        with conv1x1.step_scope():
            outs = run(n_active)
        refs = [F.conv2d(x, w, padding=w.shape[2] // 2) for w in ws[:n_active]]
        for o, r in zip(outs, refs):
            assert float((o - r).abs().max()) < 2e-5 * float(r.abs().max()), step
        loss = sum(o.pow(2).sum() for o in outs)
        gref = torch.autograd.grad(sum(r.pow(2).sum() for r in refs), [x] + ws[:n_active])
        g = torch.autograd.grad(loss, [x] + ws[:n_active])          # backward AFTER the scope closed
        for a, b in zip(g, gref):
            assert float((a - b).abs().max()) < 5e-5 * float(b.abs().max()), step
        for w, gw in zip(ws[:n_active], g[1:]):
            w.grad = gw
        opt.step()
        opt.zero_grad()
    assert len(conv1x1._entries) >= 3
    with torch.no_grad():
        ws[0].mul_(2.0)
    out = conv1x1.conv_split(x, ws[0])                                # no scope: must re-pack
    ref = F.conv2d(x, ws[0])
    assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('xs,ws,tol', [(1.0, 0.03, 1e-6), (100.0, 0.03, 1e-6), (1.0, 1.0, 1e-6), (1e-3, 0.03, 1e-4), (1e-5, 0.03, 1e-2)])
def test_fp16_forward_parts_envelope(xs, ws, tol):
    """The forward operands of the dense convolutions are split into two fp16 parts (csrc/conv1x1.hip split_pair_t, weights
    pre-scaled by 2^8): fp32-class results (<= 1e-6 of the output scale against float64; the bf16 parts gave 5e-6) for
    activations of magnitude ~1e-2 .. 6e4 and weights up to 255 -- what BatchNorm / GroupNorm-normalised networks produce.
    Below that the lo part of an activation becomes an fp16 subnormal and the error floor is an ABSOLUTE ~3e-8 per activation:
    1e-4 of the output scale at |x| ~ 1e-3, 1e-2 at 1e-5 (documented in include/kgdet_hip.h; KGDET_CONV_FWD_F16=0 selects
    bf16 parts, which have no such floor and 16 instead of 22 bits).  The same holds for 3x3 and 1x1 kernels."""
    from kgdet_amd import conv1x1 as c1
    assert c1.FORWARD_F16
    g = torch.Generator().manual_seed(0)
    for k in (1, 3):
        x = (torch.randn(2, 64, 40, 44, generator=g) * xs).cuda()
        w = (torch.randn(128, 64, k, k, generator=g) * ws).cuda()
        img = c1._pack(w, False)
        assert img.kgdet_f16
        y = c1._apply(img, x, 128, k * k)
        ref = F.conv2d(x.double(), w.double(), padding=k // 2)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        assert err < tol, (k, err)


@pytest.mark.gpu
def test_fp16_forward_parts_saturate_instead_of_nan():
    """Round-3 ADVICE (medium): an operand beyond fp16's range used to become hi = inf, lo = v - inf = -inf and the MFMA sum
    inf + (-inf) = NaN poisoned the whole output channel.  The fp16-part kernels now run with the MODE register's FP16_OVFL
    bit set (csrc/conv1x1.hip f16_saturate_on): a part beyond +/-65504 is clamped, so (a) activations up to 131008 = 2 x
    65504 are still represented by hi + lo -- hi = 65504 and a LARGE lo part, i.e. with the 11 bits of one fp16 number
    (measured 2e-4 of the output scale; fp32-class accuracy ends at 65504 as documented) --, (b) a folded BatchNorm channel with a
    tiny running variance (var = 1e-12, gamma = 1: w * s ~ 3e3 beyond the 255 the 2^8-scaled image holds) gives a FINITE,
    clamped channel and leaves every other channel exact."""
    from kgdet_amd import conv1x1 as c1
    assert c1.FORWARD_F16
    g = torch.Generator().manual_seed(1)
    for k in (1, 3):
        x = torch.randn(2, 64, 40, 44, generator=g)
        x[:, 3] *= 1e5                                  # |x| up to ~4e5 in one input channel: beyond fp16, partly beyond 2 x max
        x[:, 5] = x[:, 5].sign() * 1.2e5                # inside 2 x 65504: exactly representable by the two parts
        x = x.cuda()
        w = (torch.randn(128, 64, k, k, generator=g) * 0.05).cuda()
        img = c1._pack(w, False)
        assert img.kgdet_f16
        y = c1._apply(img, x, 128, k * k)
        assert torch.isfinite(y).all(), 'an out-of-range activation produced inf / NaN'
        xc = x.clone()
        xc[:, 3] = xc[:, 3].clamp(-131008.0, 131008.0)  # what hi + lo hold after saturation
        ref = F.conv2d(xc.double(), w.double(), padding=k // 2)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-3, (k, err)
        # a weight row beyond the image's range (a folded BatchNorm scale of 1 / sqrt(1e-12 + 1e-5) ~ 316 on |w| ~ 10)
        w2 = w.clone()
        w2[7] *= 4e4
        img2 = c1._pack(w2, False)
        x2 = torch.randn(2, 64, 40, 44, generator=g).cuda()
        y2 = c1._apply(img2, x2, 128, k * k)
        assert torch.isfinite(y2).all(), 'an out-of-range weight row produced inf / NaN'
        ref2 = F.conv2d(x2.double(), w.double(), padding=k // 2)
        keep = [o for o in range(128) if o != 7]
        err2 = float((y2[:, keep].double() - ref2[:, keep]).abs().max() / ref2[:, keep].abs().max())
        assert err2 < 2e-6, (k, err2)
