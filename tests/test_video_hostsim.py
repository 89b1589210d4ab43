"""CPU: the video (3-D) path of the planner -- frame-major clips, temporal taps / strides / dilation, stride-parity
classes over (t, h, w), the class-packed stem gradient, 3-D pooling, the two-pathway SlowFast graph -- through the
C ABI on the host simulation backend, against plain PyTorch (`nn.Conv3d`, autograd) in float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from i2v_amd import graphs, weights
from oracle import video_models as vm
from tests.hostsim_util import hostsim_engine
from tests.test_planner_hostsim import write_hook_grads


def capture(model, mods, x):
    feats, hs = [], []
    for m in mods:
        hs.append(m.register_forward_hook(lambda mod, i, o: feats.append(o)))
    model(x)
    for h in hs:
        h.remove()
    return feats


@pytest.mark.parametrize("model_type,thw", [("i3d_resnet50", (8, 32, 32)), ("slowfast_resnet50", (8, 32, 32)), ("tpn_resnet50", (4, 32, 32)),
                                            ("i3d_resnet50", (16, 24, 40)),
                                            ("i3d_resnet50", (16, 96, 96))])      # (576 positions: the attention gradients' K-split path)
def test_video_backbone_forward_backward(model_type, thw):
    eng = hostsim_engine()
    g = graphs.build_video_tiny(model_type, thw)
    sd = weights.synthetic_state_dict(g, 1)
    hooks = graphs.video_hooks(g, model_type)
    b, T = 2, thw[0]
    net = eng.build_net(g, sd, hooks, b * T)
    model = vm.load_weights(vm.make(model_type, True), sd).double()
    torch.manual_seed(5)
    x = torch.randn(b, 3, *thw, dtype=torch.float64, requires_grad=True)
    feats = capture(model, vm.hook_modules(model, model_type), x)
    assert len(feats) == len(hooks)
    net.forward(vm.to_frames(x.detach()).float())
    ffeat = [vm.to_frames(f.detach()) for f in feats]
    for i, f in enumerate(ffeat):
        assert net.hook_frames(i, b * T) == f.shape[0]
        got = net.save_hook(i, f.shape[0]).double()
        assert got.shape == f.shape
        # (absolute part relative to the tensor's scale: behind a non-local block with synthetic weights the features reach +-50)
        assert torch.allclose(got, f, rtol=1e-4, atol=1e-5 * max(1.0, float(f.abs().max()))), (i, (got - f).abs().max())
    hg = [torch.randn_like(f) for f in feats]
    cost = sum((f * h).sum() for f, h in zip(feats, hg))       # autograd applies the hooks' own ReLU gates
    ref = vm.to_frames(torch.autograd.grad(cost, x)[0])
    write_hook_grads(net, ffeat, [vm.to_frames(h) for h in hg], None)
    gx = torch.empty(b * T, 3, thw[1], thw[2])
    net.backward(gx)
    err = (gx.double() - ref).abs().max() / ref.abs().max()
    assert err < (3e-4 if "i3d" in model_type else 1e-4), err        # (i3d: softmax attention of the non-local blocks in float32)
    gx2 = gx.clone()
    net.backward(gx2, accumulate=True)
    assert torch.allclose(gx2, 2 * gx, rtol=1e-5, atol=1e-6 * float(gx.abs().max()))   # two stems: (gx+g1)+g2 vs 2(g1+g2)
    # a smaller batch than planned (whole clips only)
    net.forward(vm.to_frames(x.detach()[:1]).float())
    assert torch.allclose(net.save_hook(0, ffeat[0].shape[0] // b).double(), ffeat[0][: ffeat[0].shape[0] // b], rtol=1e-4,
                          atol=1e-5 * max(1.0, float(ffeat[0].abs().max())))
    from i2v_amd.lib import I2VError
    with pytest.raises(I2VError):
        net.forward(torch.zeros(T + 1, 3, thw[1], thw[2]))           # not a whole number of clips
    net.close()


def test_slowfast_8x8_configuration():
    """The configuration the reference names (`utils.py:11-12`: slowfast_8x8): slow pathway every 8th frame, FAST pathway every 2nd
    (a 5x7x7 stem with temporal stride AND dilation 2 on the quad-row path), 7x1x1 lateral kernel with stride alpha = 4 -- narrow
    widths, against `nn.Conv3d` + autograd in float64."""
    eng = hostsim_engine()
    thw, b = (16, 32, 32), 2
    g = graphs.slowfast_res2(16, thw, "slowfast_8x8_narrow", beta_inv=4, blocks=2, **graphs.SLOWFAST_8X8)
    sd = weights.synthetic_state_dict(g, 2)
    hooks = graphs.video_hooks(g, "slowfast_resnet50")
    net = eng.build_net(g, sd, hooks, b * thw[0])
    model = vm.load_weights(vm.SlowFastRes2(16, slow_stride=8, fast_stride=2, beta_inv=4, fusion_kernel=7, blocks=2), sd).double()
    torch.manual_seed(8)
    x = torch.randn(b, 3, *thw, dtype=torch.float64, requires_grad=True)
    feats = capture(model, vm.hook_modules(model, "slowfast_resnet50"), x)
    assert [tuple(f.shape[1:3]) for f in feats] == [(16, 8), (64, 2)]             # fast: 16 / 2 frames, slow: 16 / 8
    net.forward(vm.to_frames(x.detach()).float())
    ffeat = [vm.to_frames(f.detach()) for f in feats]
    for i, f in enumerate(ffeat):
        got = net.save_hook(i, f.shape[0]).double()
        assert got.shape == f.shape and torch.allclose(got, f, rtol=1e-4, atol=1e-5), (i, (got - f).abs().max())
    hg = [torch.randn_like(f) for f in feats]
    ref = vm.to_frames(torch.autograd.grad(sum((f * h).sum() for f, h in zip(feats, hg)), x)[0])
    write_hook_grads(net, ffeat, [vm.to_frames(h) for h in hg], None)
    gx = torch.empty(b * thw[0], 3, thw[1], thw[2])
    net.backward(gx)
    assert (gx.double() - ref).abs().max() / ref.abs().max() < 1e-4


GEOM = [  # cin, cout, (kt,k), (st,s), (pt,p), dil_t, T, H
    (5, 7, (3, 1), (1, 1), (1, 0), 1, 6, 9),
    (4, 6, (3, 3), (2, 2), (1, 1), 1, 7, 10),
    (16, 8, (5, 1), (4, 1), (2, 0), 1, 8, 6),
    (3, 5, (1, 3), (2, 1), (0, 1), 1, 8, 7),
    (16, 16, (3, 3), (1, 2), (1, 1), 1, 5, 11),
    (6, 4, (5, 3), (2, 1), (4, 1), 2, 12, 6),
    (2, 3, (2, 2), (2, 2), (0, 0), 1, 6, 8),
]


@pytest.mark.parametrize("case", GEOM)
@pytest.mark.parametrize("as_stem", [False, True])
def test_conv3d_geometry(case, as_stem):
    """One convolution behind (or as) the stem: forward, input gradient through every parity class, or the
    class-packed stem gradient."""
    cin, cout, k, s, p, dil, T, H = case
    if as_stem:
        cin = 3
    eng = hostsim_engine()
    g = graphs.Graph("geom", (H, H + 3), video=True)
    x = g.new_tensor(3, H, H + 3, False, "input", T=T)
    g.input = x
    if as_stem:
        y = g.conv3d(x, cout, k, s, p, "c.weight", bn="c_bn", relu=True, dil_t=dil)
    else:
        a = g.conv3d(x, cin, (1, 1), (1, 1), (0, 0), "a.weight", bn="a_bn", relu=True)
        y = g.conv3d(a, cout, k, s, p, "c.weight", bn="c_bn", relu=True, dil_t=dil)
    g.hooks[1] = y
    sd = weights.synthetic_state_dict(g, 2)
    b = 2
    net = eng.build_net(g, sd, [y], b * T)
    torch.manual_seed(sum(k) + T)
    xv = torch.randn(b, 3, T, H, H + 3, dtype=torch.float64, requires_grad=True)

    def bn(t, pre):
        return F.batch_norm(t, sd[pre + ".running_mean"].double(), sd[pre + ".running_var"].double(),
                            sd[pre + ".weight"].double(), sd[pre + ".bias"].double(), False, 0.0, 1e-5)
    h = xv if as_stem else F.relu(bn(F.conv3d(xv, sd["a.weight"].double()), "a_bn"))
    yv = F.relu(bn(F.conv3d(h, sd["c.weight"].double(), None, (s[0], s[1], s[1]), (p[0], p[1], p[1]), (dil, 1, 1)), "c_bn"))
    net.forward(vm.to_frames(xv.detach()).float())
    fy = vm.to_frames(yv.detach())
    assert torch.allclose(net.save_hook(0, fy.shape[0]).double(), fy, rtol=1e-4, atol=1e-5)
    hg = torch.randn_like(yv)
    ref = vm.to_frames(torch.autograd.grad((yv * hg).sum(), xv)[0])
    write_hook_grads(net, [fy], [vm.to_frames(hg)], None)
    gx = torch.empty(b * T, 3, H, H + 3)
    net.backward(gx)
    assert (gx.double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    net.close()


def test_ilaf_kernels_against_autograd():
    """i2v_ilaf_reduce/grad + the masked sign step against the reference's loss expression
    (image_attacks.py:595-617) differentiated by autograd."""
    eng = hostsim_engine()
    g = graphs.build_video_tiny("i3d_resnet50", (8, 16, 16))
    sd = weights.synthetic_state_dict(g, 0)
    hooks = graphs.video_hooks(g, "i3d")
    net = eng.build_net(g, sd, hooks, 16)
    torch.manual_seed(0)
    x = torch.rand(16, 3, 16, 16)
    net.forward(x)
    hi = net.hooks[0]
    n = net.hook_frames(0, 16)
    a = net.save_hook(0, n)
    ori = (a + 0.3 * torch.randn_like(a)).contiguous()
    adv0 = (a + 0.2 * torch.randn_like(a)).contiguous()
    scratch = torch.zeros(net.scratch_bytes(n) // 4 + 8)
    net.ilaf_reduce(0, ori, adv0, scratch, n, act=adv0)
    n0 = float(scratch[:4].view(torch.float64)[0]) ** 0.5
    assert abs(n0 - float((adv0 - ori).double().norm())) < 1e-6 * n0
    loss = torch.zeros(1)
    net.ilaf_reduce(0, ori, adv0, scratch, n)
    net.ilaf_grad(0, ori, adv0, n0, loss, scratch, n)
    ad = a.double().requires_grad_(True)
    d, d0 = ad - ori.double(), (adv0 - ori).double()
    ref_loss = -(0.5 * d.norm() / d0.norm() + (d0 / d0.norm() * d / d.norm()).sum())
    gref = torch.autograd.grad(ref_loss, ad)[0]
    if hi.post_relu:                                                  # a hooked ReLU output gates its own gradient (the I3D hook, a
        gref = gref * (a > 0)                                         # non-local block's output, is not one)
    assert abs(float(loss) - float(ref_loss.detach())) < 1e-5 * abs(float(ref_loss.detach()))
    import ctypes
    got = torch.empty_like(a)
    for f in range(n):
        ctypes.memmove(got[f].data_ptr(), hi.grad + 4 * f * hi.grad_stride, 4 * hi.D)
    assert (got.double() - gref).abs().max() < 1e-5 * gref.abs().max()
    # masked sign step
    u = torch.rand(2, 3, 5, 5)
    delta = (torch.rand(2, 3, 5, 5) - 0.5) * 0.2
    gx = torch.randn(2, 3, 5, 5)
    gx[0, 0, 0, 0] = 0.0
    eps, step = 16 / 255, 0.005
    s = u + delta.clamp(-eps, eps)
    mask = (delta >= -eps) & (delta <= eps) & (s >= 0) & (s <= 1)
    want = delta - step * torch.sign(gx) * mask
    eng.sign_step_delta_gx(delta, gx, u, eps, step)
    assert torch.equal(delta, want) and mask.float().mean() < 1
    net.close()


@pytest.mark.parametrize("thw", [(8, 24, 40), (7, 20, 20)])
def test_stem_gradient_per_temporal_class_equals_the_packed_launch(thw, monkeypatch):
    """Round 5: the input gradient of a stem with a DENSE temporal stride (the I3D's 5x7x7 / (2,2,2)) runs as one launch per temporal
    class when the 16-row halo-tile kernel is a candidate (pack_img) instead of one launch over the union of the classes' frame
    taps: zero weights add exact zeros to the k-ordered chains and the real taps keep their order, so the two packings must agree
    bit for bit -- here on the host simulation, with a stem of 16 channels (the split needs whole 16-channel groups), an odd frame
    count included, and accumulating onto an existing gradient."""
    eng = hostsim_engine()
    g = graphs.i3d_resnet((1, 1, 1, 1), 16, thw, "i3d_w16", inflate=((1,), (1,), (1,), (0,)))
    sd = weights.synthetic_state_dict(g, 2)
    hooks = [g.hooks[2]]
    T, clips = thw[0], 2
    x = torch.randn(clips * T, 3, thw[1], thw[2], generator=torch.Generator().manual_seed(1))
    outs = []
    for split in ("1", "0"):
        monkeypatch.setenv("I2V_IMG_SPLIT", split)
        net = eng.build_net(g, sd, hooks, clips * T)
        net.forward(x)
        nf = net.hook_frames(0, clips * T)
        f = net.save_hook(0, nf)
        hg = torch.randn(f.shape, generator=torch.Generator().manual_seed(4))
        write_hook_grads(net, [f], [hg], None)
        gx = torch.full((clips * T, 3, thw[1], thw[2]), float("nan"))
        net.backward(gx)
        gacc = gx.clone()
        net.backward(gacc, accumulate=True)
        outs.append((gx, gacc))
        net.close()
    assert torch.isfinite(outs[0][0]).all() and float(outs[0][0].abs().max()) > 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
