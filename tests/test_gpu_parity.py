"""GPU (-m gpu): the hand-written HIP path of libi2v_hip.so, called through the C ABI, against
the CPU oracle and the golden vectors captured from the reference.

Tolerances (fp32 path; north_star: atol 1e-4 on teacher-forced tensors):
  * elementwise kernels: bit-exact or <= 2 ulp (stated per test);
  * backbone activations: rtol 1e-4 / atol 1e-5 against the oracle run in float64;
  * input gradients for well-conditioned hook gradients: max-abs error <= 1e-4 * max|g|;
  * attack loops: cost trajectory rtol 2e-4 per step, L_inf/box invariants exact, statistics.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from i2v_amd import attacks, graphs, weights  # noqa: E402
from oracle import restate  # noqa: E402
from tests import golden_util as gu  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    e = attacks.get_engine("cuda:0")
    assert e.capi.i2v_backend() == b"hip:gfx950"
    return e


def dev(t):
    return t.to("cuda:0").contiguous()


# ------------------------------------------------------------------ elementwise kernels
def test_frames_compose_bit_exact(eng):
    gen = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (2, 3, 5, 24, 20), generator=gen, dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    x = torch.empty(10, 3, 24, 20, device="cuda:0")
    u = torch.empty_like(x)
    eng.frames_from_video(dev(vid), x, u)
    xr = restate.flatten_frames(vid).contiguous()
    ur = restate.unnormalise(xr)
    assert torch.equal(x.cpu(), xr) and torch.equal(u.cpu(), ur)
    delta = (torch.rand(10, 3, 24, 20, generator=gen) - 0.5) * 0.2        # beyond +-eps too
    delta[0, 0, 0, :4] = torch.tensor([16 / 255, -16 / 255, 0.0, 1.0])     # exact clamp edges
    out = torch.empty_like(x)
    eng.compose(u, dev(delta), out, 2, 5, 16 / 255)
    ref, _ = restate.compose(ur, delta, 16 / 255)
    assert torch.equal(out.cpu(), ref)
    outv = torch.empty(2, 3, 5, 24, 20, device="cuda:0")
    eng.compose(u, dev(delta), outv, 2, 5, 16 / 255, video_layout=True)
    assert torch.equal(outv.cpu(), restate.unflatten_frames(ref, 2, 5).contiguous())


def test_clip_from_u8_bit_exact(eng):
    u8 = torch.randint(0, 256, (2, 4, 9, 10, 3), generator=torch.Generator().manual_seed(4), dtype=torch.uint8)
    got = eng.clip_from_u8(u8.to("cuda:0")).cpu()
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    assert torch.equal(got, (u8.permute(0, 4, 1, 2, 3).float() / 255 - mean) / std)


def test_adam_step_matches_torch_optim(eng):
    """Compose-backward + Adam against restate (itself bit-equal to torch.optim.Adam)."""
    gen = torch.Generator().manual_seed(6)
    N, H, W = 4, 16, 12
    u = torch.randint(0, 256, (N, 3, H, W), generator=gen).float() / 255   # exact 0.0 / 1.0 occur
    delta = torch.full((N, 3, H, W), 0.01 / 255)
    delta.view(-1)[::7] = 16 / 255           # exactly on the clamp edge -> gradient passes (inclusive)
    delta.view(-1)[3::11] = 0.07             # outside -> gradient gated off
    st = restate.AdamState(delta, 0.005)
    d_ref = delta.clone()
    d, m, v = dev(delta), torch.zeros(N, 3, H, W, device="cuda:0"), torch.zeros(N, 3, H, W, device="cuda:0")
    ud = dev(u)
    for t in range(1, 5):
        gx = torch.randn(N, 3, H, W, generator=gen) * 1e-5
        d.copy_(d_ref); m.copy_(st.m); v.copy_(st.v)                 # teacher-forced: one step at a time
        _, mask = restate.compose(u, d_ref, 16 / 255)
        st.step(d_ref, restate.compose_backward(gx, mask))
        eng.adam_step(d, m, v, dev(gx), ud, 16 / 255, 0.005, t)
        # same op order and host-side scalar preparation; the only freedom left is fma contraction inside
        # ATen's vectorised CPU kernels: <= 2 ulp on the states, <= lr * 3e-7 on the step
        assert torch.allclose(m.cpu(), st.m, rtol=2.5e-7, atol=0)
        assert torch.allclose(v.cpu(), st.v, rtol=2.5e-7, atol=0)
        assert torch.allclose(d.cpu(), d_ref, rtol=2.5e-7, atol=0.005 * 3e-7)
    # gated-off elements never moved
    assert torch.equal(d.cpu().view(-1)[3::11], torch.full_like(d_ref.view(-1)[3::11], 0.07))


def test_sign_step_golden_bit_exact(eng):
    """BIM update (base_attacks.py:289-293) replayed on the gradients the reference saw."""
    fx = gu.load("sign_step")
    vid = gu.videos_of(fx)
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    u = dev(vid.clone().mul_(std).add_(mean))
    adv = dev(vid.clone())
    eps, steps = float(fx["eps"]), int(fx["steps"])
    cs = vid.shape[2] * vid.shape[3] * vid.shape[4]
    for g in torch.from_numpy(fx["BIM_grads"]):
        eng.sign_step(adv, u, dev(g), cs, eps / steps, eps)
    assert torch.equal(adv.cpu(), torch.from_numpy(fx["BIM_adv"]))
    d = torch.full((5, 7), 0.01, device="cuda:0")
    g = dev(torch.tensor([[-1.0, 0.0, 2.0, -0.0, 1e-30, -3.0, 5.0]]).repeat(5, 1))
    eng.sign_step_delta(d, g, 0.005)
    assert torch.equal(d.cpu(), restate.sign_step_ilaf(torch.full((5, 7), 0.01), g.cpu(), 0.005))


def test_aens_coeffs(eng):
    prev = torch.tensor([3.9, 4.0, 3.5, 3.99, 2.0, 3.0])
    c = torch.tensor([0.2, 0.1, 0.3, 0.15, 0.05, 0.2])
    cd = dev(c)
    eng.aens_coeffs(dev(prev), cd, 0.7)
    assert torch.allclose(cd.cpu(), restate.aens_coeffs(prev, c, 0.7), rtol=1e-6)


@pytest.mark.parametrize("D,N,stride_pad", [(200704, 3, 0), (5000, 4, 24), (37, 2, 3)])
def test_cossim_fwd_bwd(eng, D, N, stride_pad):
    gen = torch.Generator().manual_seed(D)
    b = torch.relu(torch.randn(N, D, generator=gen))
    a = torch.relu(b + 0.05 * torch.randn(N, D, generator=gen))
    stride = D + stride_pad
    abuf = torch.zeros(N, stride)
    abuf[:, :D] = a
    ad, bd = dev(abuf), dev(b)
    cos = torch.empty(N, device="cuda:0")
    grad = torch.full((N, stride), 7.0, device="cuda:0")
    scratch = torch.empty(eng.capi.i2v_cossim_scratch_bytes(D, N), dtype=torch.uint8, device="cuda:0")
    from i2v_amd import lib
    P = ctypes.c_void_p
    lib.check(eng.capi, eng.capi.i2v_cossim_fwd_bwd_f32(P(ad.data_ptr()), stride, P(bd.data_ptr()), D, D, N, P(0), 0, 0.5,
                                                        1, 0, P(cos.data_ptr()), P(grad.data_ptr()), stride,
                                                        P(scratch.data_ptr()), eng.stream()))
    cr, gr = restate.cosine_fwd_bwd(a.double(), b.double())
    gr = 0.5 * gr * (a > 0)
    assert torch.allclose(cos.cpu().double(), cr, atol=2e-7)
    got = grad.cpu()[:, :D].double()
    assert (got - gr).abs().max() <= 2e-6 * gr.abs().max()
    assert torch.equal(grad.cpu()[:, D:], torch.full((N, stride_pad), 7.0))     # padding untouched


# ------------------------------------------------------------------ backbone forward / backward
CASES = [("resnet", [3], 64), ("resnet", [2, 3], 64), ("resnet", [1], 32), ("vgg", [2], 32), ("vgg", [3], 32),
         ("alexnet", [3], 64), ("alexnet", [2, 4], 64), ("squeezenet", [2], 64), ("squeezenet", [2, 3], 64),
         ("squeezenet", [4], 64), ("resnet", [4], 96), ("densenet121", [1], 64), ("densenet121", [3], 64),
         ("densenet121", [2, 4], 96)]


_hip = None


def _memcpy_d2d(dst, src, nbytes):
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert _hip.hipMemcpy(dst, src, nbytes, 3) == 0          # hipMemcpyDeviceToDevice


def write_hook_grads(net, feats, hg, N):
    """Put d(cost)/d(hook) where the library expects it, gated by the hook's own ReLU (what
    i2v_cossim_fwd_bwd_f32 does on the device)."""
    torch.cuda.synchronize()
    for i, hi in enumerate(net.hooks):
        gate = (feats[i] > 0).to(hg[i].dtype) if hi.post_relu else torch.ones_like(feats[i])
        flat = dev((hg[i] * gate).float().reshape(N, -1))
        for n in range(N):
            _memcpy_d2d(hi.grad + 4 * n * hi.grad_stride, flat[n].data_ptr(), 4 * hi.D)
    torch.cuda.synchronize()


@pytest.mark.parametrize("model,depths,hw", CASES)
def test_net_forward_backward_match_oracle(eng, model, depths, hw, monkeypatch):
    # (every activation is read back: the tiny ResNet's 4- / 8-channel bottlenecks qualify for the fused fast-pathway block, whose
    #  intermediates are never stored -- planned here as separate launches; the fused block has its own tests in test_gpu_video.py)
    monkeypatch.setenv("I2V_FASTBLOCK", "0")
    g = graphs.build_tiny(model, (hw, hw))
    sd = weights.synthetic_state_dict(g, 3)
    hooks = [g.hooks[d] for d in depths]
    N = 5
    net = eng.build_net(g, sd, hooks, N)
    onet = restate.OracleNet(g, sd, hooks, dtype=torch.float64)
    torch.manual_seed(hw + len(depths))
    x = torch.randn(N, 3, hw, hw)
    feats = onet.forward(x.double())
    net.forward(dev(x))
    for nd in net.graph.nodes:
        got = net.read_tensor(nd.dst, N).cpu().double()
        ref = onet.tensor(nd.dst)
        assert (got - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-6, nd
    hg = [torch.randn_like(f) for f in feats]
    write_hook_grads(net, feats, hg, N)
    gx = torch.empty(N, 3, hw, hw, device="cuda:0")
    net.backward(gx)
    ref = onet.backward(hg)
    err = (gx.cpu().double() - ref).abs().max() / ref.abs().max()
    assert err < 1e-4, err
    gx2 = gx.clone()
    net.backward(gx2, accumulate=True)
    assert torch.allclose(gx2, 2 * gx, rtol=1e-6, atol=1e-12)
    # fewer frames than planned
    net.forward(dev(x[:2]))
    assert torch.allclose(net.read_tensor(hooks[-1], 2).cpu().double(), feats[-1][:2], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("model,depths", [("resnet50", [3]), ("vgg", [3]), ("alexnet", [2, 4]), ("squeezenet", [2, 4]),
                                          ("densenet121", [2, 4]),
                                          # round 5: the reference's OWN model grid -- `'resnet'` is ResNet-101, the CLI default
                                          # (image_attacks.py:94-95), and run_image_guided.py:55-60 sweeps 4 models x depths 1-4.
                                          # ResNet layer4 at 224^2 has 7x7 = 49-pixel planes (not a multiple of 4: the scalar epilogue
                                          # at Cd = 2048); VGG depth 1 is a 64-channel 224^2 plane.
                                          ("resnet", [1]), ("resnet", [2]), ("resnet", [3]), ("resnet", [4]), ("vgg", [1]), ("vgg", [4]),
                                          ("vgg", [2]), ("alexnet", [1, 3]), ("squeezenet", [1, 3])])
def test_full_size_layers(eng, model, depths):
    """Real shapes (224^2) on 2 frames: every conv configuration of SURVEY.md 8(a4) (ResNet-50 to
    layer3), VGG-16 to features[20], AlexNet 11x11/4 + 5x5 + 3x3, SqueezeNet ceil-mode pools and Fire
    concat -- every activation and the input gradient against the oracle's ATen ops."""
    g = graphs.build(model, (224, 224))
    sd = weights.synthetic_state_dict(g, 0)
    hooks = [g.hooks[d] for d in depths]
    N = 2
    net = eng.build_net(g, sd, hooks, N)
    onet = restate.OracleNet(g, sd, hooks, dtype=torch.float32)
    x = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, N, 224, 224), generator=torch.Generator().manual_seed(1),
                                               dtype=torch.uint8).numpy()})
    x = restate.flatten_frames(x).contiguous()
    onet.forward(x)
    net.forward(dev(x))
    for nd in net.graph.nodes:
        got = net.read_tensor(nd.dst, N).cpu()
        ref = onet.tensor(nd.dst)
        assert (got - ref).abs().max() <= 2e-4 * ref.abs().max() + 1e-5, nd
    # gradient comparison with the oracle gated by the DEVICE's activations: millions of activations per
    # frame, a handful within fp32 noise of zero would otherwise gate differently
    onet.adopt_activations({nd.dst: net.read_tensor(nd.dst, N).cpu() for nd in net.graph.nodes})
    feats = [onet.tensor(h) for h in hooks]
    hg = [torch.randn_like(f) for f in feats]
    write_hook_grads(net, feats, hg, N)
    gx = torch.empty(N, 3, 224, 224, device="cuda:0")
    net.backward(gx)
    ref = onet.backward(hg)
    assert (gx.cpu() - ref).abs().max() <= 2e-4 * ref.abs().max()


def test_full_size_mid_trajectory_teacher_forced_step_resnet50(eng):
    """VERDICT r2 (weak 1): the headline configuration -- ResNet-50 layer3, 224^2, one 32-frame clip -- run for 3 free steps
    on the HIP engine; from (delta_3, m_3, v_3) of two frames ONE engine iteration against ONE float64 oracle iteration
    (`image_attacks.py:325-358`): cost rtol 2e-4, gradient 1e-4 max|g| on >= 99 % of the pixels, delta_4 atol 1e-4 on every
    pixel with |g| >= 5 % max.  This is the well-conditioned counterpart of the delta_0 checks (cos = 1 - 1e-9 there)."""
    g = graphs.build("resnet50", (224, 224))
    onet = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[3]], dtype=torch.float64)
    u8 = torch.randint(0, 256, (1, 3, 32, 224, 224), generator=torch.Generator().manual_seed(1000), dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    mk = lambda steps: attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=steps, weight_seed=0)   # noqa: E731
    gu.check_mid_trajectory_step(mk, [onet], vid, [5, 29], t=3, lr=0.005, tag="resnet50 layer3 224^2")


def test_full_size_mid_trajectory_teacher_forced_step_resnet101_depth2(eng):
    """The paper's Table-3 setting (run_image_guided.py:63-70: `--direction_image_model resnet --depth 2`, i.e. ResNet-101 layer2 --
    image_attacks.py:94-95,260-271) at 224^2: 3 free steps on the HIP engine over one 32-frame clip, then ONE engine iteration from
    (delta_3, m_3, v_3) of two frames against ONE float64 oracle iteration -- the same bounds as the ResNet-50 layer3 step above."""
    g = graphs.build("resnet", (224, 224))
    assert g.arch == "resnet101"
    onet = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[2]], dtype=torch.float64)
    u8 = torch.randint(0, 256, (1, 3, 32, 224, 224), generator=torch.Generator().manual_seed(1001), dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    mk = lambda steps: attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=2, step_size=0.005, steps=steps, weight_seed=0)   # noqa: E731
    gu.check_mid_trajectory_step(mk, [onet], vid, [3, 30], t=3, lr=0.005, tag="resnet101 layer2 224^2")


def _one_conv_graph(cin, cout, k, stride, pad, hw, relu, residual):
    """input(3) -> 3x3 conv to `cin` channels -> the convolution under test [-> hook]."""
    g = graphs.Graph("unit", (hw, hw))
    x = g.new_tensor(3, hw, hw, False, "input")
    g.input = x
    a = g.conv(x, cin, 3, 1, 1, "a.weight", bias="a.bias", relu=True, name="a")
    res = None
    if residual and stride == 1 and 2 * pad == k - 1 and cin == cout:
        res = a
    b = g.conv(a, cout, k, stride, pad, "b.weight", bn="b.bn", relu=relu, residual=res, name="b")
    g.hooks[1] = b
    return g


def test_conv_property_based(eng):
    """Randomised geometry (hypothesis-style sweep with a fixed seed so the GPU box needs no database):
    odd planes (HW % 4 != 0 -> scalar epilogue), channel counts that are not multiples of 16/32 (K tails,
    padded Cd rows, per-row k-table mode), strides 1-3, kernels 1-5, every tile configuration."""
    import random
    rnd = random.Random(1234)
    for case in range(40):
        k = rnd.choice([1, 1, 3, 3, 5, 2])
        stride = rnd.choice([1, 1, 2, 3])
        pad = rnd.choice([0, k // 2])
        cin = rnd.choice([3, 8, 16, 24, 32, 48, 64, 96, 130])
        cout = rnd.choice([4, 16, 31, 32, 64, 80, 128, 200])
        hw = rnd.choice([7, 12, 13, 16, 28, 30])
        if hw + 2 * pad < k:
            continue
        N = rnd.choice([1, 3, 5])
        relu, residual = rnd.random() < 0.7, rnd.random() < 0.5
        g = _one_conv_graph(cin, cout, k, stride, pad, hw, relu, residual)
        sd = weights.synthetic_state_dict(g, case)
        net = eng.build_net(g, sd, [g.hooks[1]], N)
        onet = restate.OracleNet(g, sd, [g.hooks[1]], dtype=torch.float64)
        x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(case))
        feats = onet.forward(x.double())
        net.forward(dev(x))
        tag = (case, cin, cout, k, stride, pad, hw, N, relu, residual)
        for nd in net.graph.nodes:
            got, ref = net.read_tensor(nd.dst, N).cpu().double(), onet.tensor(nd.dst)
            assert (got - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-6, tag
        hg = [torch.randn_like(f) for f in feats]
        write_hook_grads(net, feats, hg, N)
        gx = torch.empty(N, 3, hw, hw, device="cuda:0")
        net.backward(gx)
        ref = onet.backward(hg)
        assert (gx.cpu().double() - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-7, tag


@pytest.mark.parametrize("k,stride,pad", [(3, 1, 1), (5, 3, 2), (2, 1, 0), (3, 2, 1), (4, 4, 0)])
def test_maxpool_geometries(eng, k, stride, pad):
    """Max-pooling windows beyond the three the backbones use (those have compile-time specialisations of pool_bwd;
    everything else runs the generic instantiation): conv -> pool -> conv, forward and input gradient vs the oracle."""
    g = graphs.Graph("poolgeom", (29, 26))
    x = g.new_tensor(3, 29, 26, False, "input")
    g.input = x
    a = g.conv(x, 12, 3, 1, 1, "a.weight", bn="a_bn", relu=True)
    pl = g.maxpool(a, k, stride, pad)
    y = g.conv(pl, 8, 3, 1, 1, "c.weight", bn="c_bn", relu=True)
    g.hooks[1] = y
    sd = weights.synthetic_state_dict(g, k * 10 + stride)
    N = 3
    net = eng.build_net(g, sd, [y], N)
    onet = restate.OracleNet(g, sd, [y], dtype=torch.float64)
    xin = torch.randn(N, 3, 29, 26, generator=torch.Generator().manual_seed(k))
    feats = onet.forward(xin.double())
    net.forward(dev(xin))
    assert torch.allclose(net.read_tensor(pl, N).cpu().double(), onet.tensor(pl), rtol=1e-5, atol=1e-6)
    hg = [torch.randn_like(f) for f in feats]
    write_hook_grads(net, feats, hg, N)
    gx = torch.empty(N, 3, 29, 26, device="cuda:0")
    net.backward(gx)
    ref = onet.backward(hg)
    assert (gx.cpu().double() - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-7
    net.close()


# ------------------------------------------------------------------ attack loops
MODE = {"i2v": attacks.ImageGuidedFMDirection_Adam, "std": attacks.ImageGuidedStd_Adam}


@pytest.mark.parametrize("name", ["i2v_resnet_d3_f64", "i2v_resnet_d2_f32", "i2v_vgg_d2_f64", "i2v_alexnet_d3_f64",
                                  "i2v_squeezenet_d2_f64", "std_resnet_d2_f64"])
def test_attack_loop_against_golden(eng, name):
    fx = gu.load(name)
    atk = MODE[fx["kind"]](fx["models"], depth=fx["depth"], step_size=fx["lr"], steps=fx["steps"],
                           graph_builder=graphs.build_tiny, weight_seed=fx["wseed"])
    vid = gu.videos_of(fx)
    adv = atk(vid, torch.zeros(fx["b"], dtype=torch.long), ["clip0"]).cpu()
    ref_cost = np.array([float(s) for s in fx["cost_str"]])
    np.testing.assert_allclose(atk.last_costs, ref_cost, rtol=2e-4)
    assert adv.shape == vid.shape
    assert atk.loss_info["clip0"][0]["cost"] == str(np.float32(atk.last_costs[0]))
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    un = adv * std + mean
    clean = torch.from_numpy(fx["clip_u8"]).float() / 255
    assert (un - clean).abs().max() <= 16 / 255 + 1e-6
    assert un.min() >= -1e-6 and un.max() <= 1 + 1e-6
    assert np.abs(adv.numpy() - fx["adv"]).mean() < 5e-3
    dl, rl = atk._delta.cpu().numpy(), fx["delta_last"]
    assert abs(np.abs(dl).mean() / np.abs(rl).mean() - 1) < 0.02


def test_first_step_against_reference_gradient(eng):
    fx = gu.load("i2v_resnet_d3_f64")
    atk = attacks.ImageGuidedFMDirection_Adam(fx["models"], depth=fx["depth"], step_size=fx["lr"], steps=1,
                                              graph_builder=graphs.build_tiny)
    atk(gu.videos_of(fx), torch.zeros(1, dtype=torch.long), ["c"])
    r0, d1 = fx["grad0"], atk._delta.cpu().numpy()
    well = np.abs(r0) > 5e-2 * np.abs(r0).max()
    assert (np.abs(d1 - fx["delta_first"])[well] < 1e-4).all()          # north_star atol on a forced step
    assert (np.abs(d1 - fx["delta_first"])[well] < 2e-5).mean() > 0.99
    assert (np.abs(d1 - fx["delta_first"]) < 1e-4).mean() > 0.9


@pytest.mark.parametrize("name", ["tf_i2v_resnet_d3_f64", "tf_ens_f64", "tf_aens_f64"])
def test_teacher_forced_steps_against_reference_states(eng, name):
    """north_star's atol 1e-4 on EVERY step, not only the first: each iteration of the HIP loop is restarted from the
    reference's (delta_i, exp_avg_i, exp_avg_sq_i) [and AENS coefficients] and must land on the reference's
    delta_{i+1} / cost_i (I2V image_attacks.py:325-358, ENS :456-490, AENS TPAMI_attack.py:258-312; the fixture's
    AENS case hooks SqueezeNet's whole Fire modules, :195-197).  Tolerances: gu.check_teacher_forced."""
    fx = gu.load(name)
    gu.check_teacher_forced(fx, gu.make_attack(fx, attacks), to_dev=dev)


def test_ens_and_aens_against_golden(eng):
    fx = gu.load("ens_4models_f64")
    atk = attacks.ImageGuidedFML2_Adam_MultiModels(fx["models"], depths=fx["depth"], steps=fx["steps"],
                                                   graph_builder=graphs.build_tiny)
    adv = atk(gu.videos_of(fx), torch.zeros(fx["b"], dtype=torch.long), ["clip0", "clip1"]).cpu()
    np.testing.assert_allclose(atk.last_costs, np.array([float(s) for s in fx["cost_str"]]), rtol=2e-4)
    assert np.abs(adv.numpy() - fx["adv"]).mean() < 5e-3
    for name in ("aens_2x2_f64", "aens_coefce_f64"):
        fx = gu.load(name)
        atk = attacks.AENS_I2V_MF(fx["models"], depths=fx["depth"], step_size=fx["lr"], steps=fx["steps"],
                                  graph_builder=graphs.build_tiny, **fx["kw"])
        adv, used_time, cost_saved = atk(gu.videos_of(fx), torch.zeros(fx["b"], dtype=torch.long), ["c"] * fx["b"])
        np.testing.assert_allclose(cost_saved, fx["cost_saved"], rtol=2e-4)
        np.testing.assert_allclose(np.stack(atk.weights), fx["weights"], rtol=1e-4)
        np.testing.assert_allclose(atk.coeffs.cpu().numpy(), fx["coeffs_after"], rtol=1e-4)


def test_full_size_properties_resnet50(eng):
    """BASELINE config 1 size (1 clip x 32 x 224^2, ResNet-50 layer3, 10 steps): size-independent
    properties -- L_inf bound, [0,1] box, falling cost, bit-reproducibility (no atomics anywhere),
    frame independence (a frame's trajectory does not depend on its batch neighbours)."""
    gen = torch.Generator().manual_seed(1000)
    u8 = torch.randint(0, 256, (1, 3, 32, 224, 224), generator=gen, dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10)
    adv = atk(vid, torch.zeros(1, dtype=torch.long), ["v"]).cpu()
    costs = atk.last_costs.copy()
    assert costs[0] > 31.9 and costs[-1] < costs[0] and np.all(np.diff(costs) < 1e-3)
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    un = adv * std + mean
    assert (un - u8.float() / 255).abs().max() <= 16 / 255 + 1e-6
    assert un.min() >= -1e-6 and un.max() <= 1 + 1e-6
    adv2 = atk(vid, torch.zeros(1, dtype=torch.long), ["v"]).cpu()
    assert torch.equal(adv, adv2)
    sub = atk(vid[:, :, 8:12].contiguous(), torch.zeros(1, dtype=torch.long), ["w"]).cpu()
    assert torch.equal(sub, adv[:, :, 8:12])


def test_clip_lanes_full_size_bit_identical(eng):
    """BASELINE configs[1] shape (4 clips x 32 x 224^2, ResNet-50 layer3, 10 steps): the default concurrent clip
    lanes (two clips each) against a single lane -- same bytes out, since frames are independent and no kernel's summation order depends
    on the batch."""
    vid = torch.cat([gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, 32, 224, 224), generator=torch.Generator().manual_seed(1000 + i),
                                                           dtype=torch.uint8).numpy()}) for i in range(4)])
    names = [f"c{i}" for i in range(4)]
    two = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10)
    assert two._lane_count(4, 32) == 2 and two._lane_count(8, 32) == 2
    got = two(vid, torch.zeros(4, dtype=torch.long), names).cpu()
    one = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10)
    one.clip_lanes = 1
    ref = one(vid, torch.zeros(4, dtype=torch.long), names).cpu()
    assert torch.equal(got, ref)
    assert np.array_equal(two.last_costs, one.last_costs)      # canonical per-clip summation: the split does not show in the log either
    assert torch.equal(two(vid, torch.zeros(4, dtype=torch.long), names).cpu(), ref)
    # one clip (the reference CLI's default batch): one lane by default; on request two lanes take its frames 0..15 and 16..31
    assert two._lane_count(1, 32) == 1
    assert torch.equal(two(vid[:1], torch.zeros(1, dtype=torch.long), names[:1]).cpu(), ref[:1])
    two.clip_lanes = 2
    assert two._lane_count(1, 32) == 2
    assert torch.equal(two(vid[:1], torch.zeros(1, dtype=torch.long), names[:1]).cpu(), ref[:1])


def test_launch_overlap_full_size_bit_identical(eng, monkeypatch):
    """Round 6: one 32-frame 224^2 clip, ResNet-50 layer3, 10 steps -- the reference CLI's default batch -- with the
    projection shortcuts (and their input gradients) overlapped on the net's side stream (`mark_overlap`; measured slower than the plain
    list -- 589 vs 610 frames/s -- and therefore OFF unless `I2V_OVERLAP_MAX_FRAMES` asks) against the same attack planned without:
    same bytes out, same costs, and the side stream really ran launches."""
    vid = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, 32, 224, 224), generator=torch.Generator().manual_seed(1000), dtype=torch.uint8).numpy()})
    lab = torch.zeros(1, dtype=torch.long)
    monkeypatch.setenv("I2V_OVERLAP_MAX_FRAMES", "0")
    off = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10)
    before = eng.capi.i2v_backend_stat(b"overlap_launches")
    ref = off(vid, lab, ["c0"]).cpu()
    assert eng.capi.i2v_backend_stat(b"overlap_launches") == before
    monkeypatch.setenv("I2V_OVERLAP_MAX_FRAMES", "32")
    on = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10)
    got = on(vid, lab, ["c0"]).cpu()
    ran = eng.capi.i2v_backend_stat(b"overlap_launches") - before
    assert ran >= 11 * 3 + 10 * 3, ran          # three projection shortcuts: 11 forward passes, 10 backward passes
    assert torch.equal(got, ref) and np.array_equal(on.last_costs, off.last_costs)
    assert torch.equal(on(vid, lab, ["c0"]).cpu(), ref)          # and again (event pool reused)
    # ResNet-101 to layer2 and the four-backbone ensemble's nets take the same path
    for models, depth in ((["resnet"], 2), (["squeezenet"], 2)):
        monkeypatch.setenv("I2V_OVERLAP_MAX_FRAMES", "0")
        a = attacks.ImageGuidedFMDirection_Adam(models, depth=depth, step_size=0.005, steps=3)
        ra = a(vid[:, :, :8].contiguous(), lab, ["c0"]).cpu()
        monkeypatch.setenv("I2V_OVERLAP_MAX_FRAMES", "32")
        b = attacks.ImageGuidedFMDirection_Adam(models, depth=depth, step_size=0.005, steps=3)
        assert torch.equal(b(vid[:, :, :8].contiguous(), lab, ["c0"]).cpu(), ra)


def test_clip_lanes_ensemble_and_odd_split(eng):
    """ENS-I2V (four backbones, gradient accumulation) with 3 clips -> lanes of 1 and 2 clips; and a 24-frame single
    clip -> frame lanes of 12 + 12: same bytes as one lane."""
    gen = torch.Generator().manual_seed(31)
    vid = gu.videos_of({"clip_u8": torch.randint(0, 256, (3, 3, 4, 64, 64), generator=gen, dtype=torch.uint8).numpy()})
    depths = {"resnet": 2, "vgg": 3, "squeezenet": 2, "alexnet": 3}
    kw = dict(model_name_lists=list(depths), depths=depths, steps=3, graph_builder=graphs.build_tiny)
    two = attacks.ImageGuidedFML2_Adam_MultiModels(**kw)
    two.clip_lanes = 2
    one = attacks.ImageGuidedFML2_Adam_MultiModels(**kw)
    one.clip_lanes = 1
    names = ["a", "b", "c"]
    assert two._lane_count(3, 4) == 2
    assert torch.equal(two(vid, torch.zeros(3, dtype=torch.long), names), one(vid, torch.zeros(3, dtype=torch.long), names))
    assert np.array_equal(two.last_costs, one.last_costs)      # canonical per-clip summation: the split does not show in the log either
    clip = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, 24, 64, 64), generator=gen, dtype=torch.uint8).numpy()})
    assert two._lane_count(1, 24) == 2
    assert torch.equal(two(clip, torch.zeros(1, dtype=torch.long), ["v"]), one(clip, torch.zeros(1, dtype=torch.long), ["v"]))


def test_ensemble_full_size_and_frame_slicing(eng):
    """Full-size shapes of the other backbones (BASELINE configs[2]-style ensemble on the reference's own
    model list, image_main.py:73-79): AlexNet 11x11/4, SqueezeNet ceil-mode pools and Fire concat, VGG
    224^2 planes, ResNet-101 layer2 -- 2 steps, invariants + bit-reproducibility.  Then VGG on 192 frames:
    its first activation buffer (64x224^2 per frame) spans 2.4 GiB, so the convolution launches are
    sliced over frames (32-bit buffer offsets); frames must not notice."""
    gen = torch.Generator().manual_seed(2000)
    u8 = torch.randint(0, 256, (1, 3, 32, 224, 224), generator=gen, dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    depths = {"resnet": 2, "vgg": 3, "squeezenet": 2, "alexnet": 3}
    atk = attacks.ImageGuidedFML2_Adam_MultiModels(["resnet", "vgg", "squeezenet", "alexnet"], depths=depths, steps=2)
    adv = atk(vid, torch.zeros(1, dtype=torch.long), ["v"]).cpu()
    costs = atk.last_costs.copy()
    assert abs(costs[0] - 4 * 32) < 0.5 and costs[1] < costs[0]
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    un = adv * std + mean
    assert (un - u8.float() / 255).abs().max() <= 16 / 255 + 1e-6 and un.min() >= -1e-6 and un.max() <= 1 + 1e-6
    assert torch.equal(adv, atk(vid, torch.zeros(1, dtype=torch.long), ["v"]).cpu())
    del atk
    torch.cuda.empty_cache()
    g = graphs.build("vgg", (224, 224))
    sd = weights.synthetic_state_dict(g, 0)
    big = eng.build_net(g, sd, [g.hooks[2]], 192)
    small = eng.build_net(g, sd, [g.hooks[2]], 2)
    x = restate.flatten_frames(vid).contiguous()[:2]
    xb = torch.zeros(192, 3, 224, 224, device="cuda:0")
    xb[0], xb[191] = x[0], x[1]
    big.forward(xb)
    small.forward(dev(x))
    fb, fs = big.read_tensor(g.hooks[2], 192), small.read_tensor(g.hooks[2], 2)
    assert torch.equal(fb[0], fs[0]) and torch.equal(fb[191], fs[1])


def test_aens_full_size_one_step(eng):
    """Adaptive ENS at full size: two hooks per backbone (the shallower hooked tensor also feeds deeper
    layers -> side-buffer hook gradient merged by addmask), device-side coefficients."""
    gen = torch.Generator().manual_seed(2001)
    u8 = torch.randint(0, 256, (1, 3, 8, 224, 224), generator=gen, dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    names = ["resnet", "vgg", "squeezenet", "alexnet"]
    atk = attacks.AENS_I2V_MF(names, depths={n: [2, 3] for n in names}, step_size=0.005, steps=2, momentum=0.5)
    adv, used, costs = atk(vid, torch.zeros(1, dtype=torch.long), ["v"])
    w = np.stack(atk.weights)
    assert w.shape == (2, 8) and np.allclose(w.sum(1), 1, atol=1e-5) and np.allclose(w[0], 1 / 8, atol=1e-6)
    assert abs(costs[0] - 8.0 * (1 / 8)) < 0.05 and costs[1] < costs[0]       # mean_l coeff_l * sum_frames cos ~ 8 * 1/8
    adv2, _, costs2 = attacks.AENS_I2V_MF(names, depths={n: [2, 3] for n in names}, step_size=0.005, steps=2,
                                          momentum=0.5)(vid, torch.zeros(1, dtype=torch.long), ["v"])
    assert torch.equal(adv, adv2) and np.array_equal(costs, costs2)


def test_proxy_fooling_rate_parity(eng, tmp_path, monkeypatch):
    """The metric's second half on what is available offline: adversarial clips from the CPU oracle and
    from the HIP path, scored by the SAME evaluator code (reference.py contract) with its built-in proxy
    video models, must give the same top-1 / fooling rate (north_star: within +-0.5 %)."""
    import reference as ev
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    g = graphs.build_tiny("resnet", (64, 64))
    sd = weights.synthetic_state_dict(g, 0)
    onet = restate.OracleNet(g, sd, [g.hooks[3]])
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=3, step_size=0.005, steps=6, graph_builder=graphs.build_tiny)
    for d in ("oracle", "hip", "clean"):
        (tmp_path / d).mkdir()
    for label in range(10):
        u8 = torch.randint(0, 256, (1, 3, 4, 64, 64), generator=torch.Generator().manual_seed(300 + label), dtype=torch.uint8)
        vid = gu.videos_of({"clip_u8": u8.numpy()})
        np.save(tmp_path / "clean" / f"{label}-ori.npy", vid[0].numpy())
        np.save(tmp_path / "oracle" / f"{label}-adv.npy", restate.run_attack([onet], vid, steps=6, step_size=0.005)["adv"][0].contiguous().numpy())
        np.save(tmp_path / "hip" / f"{label}-adv.npy", atk(vid, torch.zeros(1, dtype=torch.long), [str(label)])[0].cpu().numpy())
    common = ["--models", "i3d_resnet50,slowfast_resnet50,tpn_resnet50", "--clean_dir", str(tmp_path / "clean")]
    a = ev.main(["--adv_path", "oracle"] + common)
    b = ev.main(["--adv_path", "hip"] + common)
    for k in a:
        assert abs(a[k] - b[k]) <= 0.5, (a, b)
    assert (tmp_path / "hip" / "results_all_models_prediction.csv").read_text() == \
        (tmp_path / "oracle" / "results_all_models_prediction.csv").read_text()


def test_config2_ensemble_resnet50_vgg16_densenet121(eng):
    """BASELINE.json configs[2] model set at full size (1 clip x 8 frames, 2 steps): invariants,
    bit-reproducibility, and agreement of the first-step cost with L*N (cos = 1 at delta_0)."""
    gen = torch.Generator().manual_seed(2002)
    u8 = torch.randint(0, 256, (1, 3, 8, 224, 224), generator=gen, dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    names = ["resnet50", "vgg", "densenet121"]
    mk = lambda: attacks.ImageGuidedFML2_Adam_MultiModels(names, depths={n: 3 for n in names}, steps=2)   # noqa: E731
    atk = mk()
    adv = atk(vid, torch.zeros(1, dtype=torch.long), ["v"]).cpu()
    assert abs(atk.last_costs[0] - 3 * 8) < 0.1 and atk.last_costs[1] < atk.last_costs[0]
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    un = adv * std + mean
    assert (un - u8.float() / 255).abs().max() <= 16 / 255 + 1e-6 and un.min() >= -1e-6 and un.max() <= 1 + 1e-6
    assert torch.equal(adv, mk()(vid, torch.zeros(1, dtype=torch.long), ["v"]).cpu())


def test_rccl_exchange_paths_single_rank(eng):
    """The collectives of the clip-sharded runs on the real backend (RCCL through torch.distributed
    'nccl'), with a 1-rank group: AENS' 2L-float all-reduce and DR's 3-double all-reduce must leave the
    single-device results untouched.  (Two-rank semantics are covered by the gloo tests on CPU.)"""
    import os
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        fx = gu.load("aens_2x2_f64")
        kw = dict(depths=fx["depth"], step_size=fx["lr"], steps=fx["steps"], graph_builder=graphs.build_tiny, **fx["kw"])
        vid = gu.videos_of(fx)
        lab = torch.zeros(fx["b"], dtype=torch.long)
        a0 = attacks.AENS_I2V_MF(fx["models"], **kw)
        adv0, _, c0 = a0(vid, lab, ["c"] * fx["b"])
        a1 = attacks.AENS_I2V_MF(fx["models"], distributed=True, **kw)
        adv1, _, c1 = a1(vid, lab, ["c"] * fx["b"])
        assert torch.equal(adv0, adv1) and np.array_equal(c0, c1) and np.array_equal(np.stack(a0.weights), np.stack(a1.weights))
        d0 = attacks.ImageGuidedStd_Adam(["resnet"], depth=2, step_size=0.005, steps=3, graph_builder=graphs.build_tiny)
        d1 = attacks.ImageGuidedStd_Adam(["resnet"], depth=2, step_size=0.005, steps=3, graph_builder=graphs.build_tiny,
                                         distributed=True)
        assert torch.equal(d0(vid, lab, ["c"]), d1(vid, lab, ["c"])) and np.array_equal(d0.last_costs, d1.last_costs)
    finally:
        if created:
            dist.destroy_process_group()


def test_resize_crop_normalise_bit_exact(eng):
    """N4 on the device: decoded uint8 frames -> Resize(256, cv2-style 8-bit bilinear) -> CenterCrop(224) -> /255 ->
    Normalize -> (b,3,t,224,224) in ONE kernel (`i2v_clip_resize_crop_u8_f32`), bit for bit against the oracle's
    restatement of the reference loader's validation transform (datasets.py:86-93), on a Kinetics-like 340x256 clip,
    a portrait one and an already-sized one; and straight into an attack."""
    for H, W, t in ((256, 340, 8), (360, 300, 2), (256, 256, 2), (224, 224, 1)):
        fr = torch.randint(0, 256, (2, t, H, W, 3), generator=torch.Generator().manual_seed(H * W), dtype=torch.uint8)
        got = eng.clip_resize_crop(fr.to("cuda:0")).cpu()
        assert torch.equal(got, restate.resize_center_crop_normalise(fr.numpy()))
    fr = torch.randint(0, 256, (1, 4, 100, 130, 3), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    vid = eng.clip_resize_crop(fr.to("cuda:0"), short_side=72, crop=64)
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=3, step_size=0.005, steps=2, graph_builder=graphs.build_tiny, weight_seed=0)
    adv = atk(vid, torch.zeros(1, dtype=torch.long), ["v"])
    assert adv.shape == (1, 3, 4, 64, 64) and atk.last_costs[1] < atk.last_costs[0]
