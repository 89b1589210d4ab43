"""The white-box CLASSIFIER path of the BIM family with everything behind the C ABI (`attack.py:63-96`,
`base_attacks.py:261-340`): `VideoModel(..., num_classes=K)` = I3D or SlowFast graph to its last stage + global-pool / fc head
(SlowFast: both pathways pooled separately and concatenated, slow first);
`autograd.grad(CrossEntropyLoss()(model(adv), labels), adv)` becomes forward -> `i2v_head_ce_f32` -> input-gradient.
Checked against torch autograd in float64 on the same weights (the plain-PyTorch I3D of oracle/video_models.py + pool + fc),
on the host simulation here and on the HIP kernels in the `gpu`-marked twin."""
import numpy as np
import pytest
import torch

from i2v_amd import graphs, sign_attacks, video, weights
from oracle import video_models as vm


def torch_classifier(model_type, thw, seed, K):
    g = graphs.build_video_tiny(model_type, thw, full=True)
    sd = weights.synthetic_state_dict(g, seed)
    m = video.VideoModel(model_type, thw, weight_seed=seed, tiny=True, num_classes=K)
    W, b = m.head_weights(g)
    back = vm.load_weights(vm.make(model_type, True, full=True), sd).double()

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.back = back
            self.fc = torch.nn.Linear(W.shape[1], K).double()
            with torch.no_grad():
                self.fc.weight.copy_(W.double()); self.fc.bias.copy_(b.double())

        def forward(self, x):
            f = self.back(x.double())
            feats = f if isinstance(f, tuple) else (f,)
            return self.fc(torch.cat([t.mean(dim=(2, 3, 4)) for t in feats], dim=1))
    return m, Net().eval()


def check_gradient(eng, dev, model_type="i3d_resnet50"):
    thw, K, b = (8, 32, 32), 7, 2
    m, ref = torch_classifier(model_type, thw, 3, K)
    gen = torch.Generator().manual_seed(11)
    vid = torch.randn(b, 3, *thw, generator=gen)
    labels = torch.tensor([2, 5])
    atk = sign_attacks.BIM(m, steps=1, engine=eng)
    g = atk._grad(vid.to(dev), labels).cpu().double()
    x = vid.double().requires_grad_(True)
    logits = ref(x)
    loss = torch.nn.CrossEntropyLoss()(logits, labels)
    gref = torch.autograd.grad(loss, x)[0]
    np.testing.assert_allclose(atk.last_logits.cpu().double().numpy(), logits.detach().numpy(), rtol=1e-4, atol=1e-5)
    assert abs(float(atk.last_loss) - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss)))
    assert (g - gref).abs().max() <= 2e-4 * gref.abs().max(), float((g - gref).abs().max() / gref.abs().max())
    # targeted attacks flip the sign of the cost (base_attacks.py:229-231, 282)
    atk._targeted = -1
    g2 = atk._grad(vid.to(dev), labels).cpu().double()
    assert torch.equal(g2, -g)
    return m, ref, vid, labels


def check_attacks(eng, dev, model_type="i3d_resnet50"):
    """FGSM / BIM / MI-FGSM with the native classifier against the same classes driving the float64 torch module (the
    reference's calling convention): same update kernel on both sides, so the clips agree wherever the gradient's sign is
    not decided in the last bits."""
    thw, K = (32, 32, 32), 5         # norm_grads asserts 32 frames, like the reference (utils.py:58-67)
    m, ref = torch_classifier(model_type, thw, 4, K)
    vid = torch.randn(1, 3, *thw, generator=torch.Generator().manual_seed(12)) * 0.5
    labels = torch.tensor([1])

    class F32(torch.nn.Module):          # the torch path wants float32 in / out
        def __init__(self, net):
            super().__init__()
            self.net = net
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return self.net(x).float()
    # The I3D graphs carry non-local blocks (the reference's i3d_nl5): with SYNTHETIC weights their un-normalised softmax attention
    # saturates, and after two eps/3 steps the loss surface is chaotic -- measured on this clip: two float64 evaluations whose
    # inputs differ in 0.008 % of the pixels have gradients of max 49 vs 194, while the native gradient at the SAME point agrees
    # with float64 autograd in 99.999 % of the signs (float32 torch: the same).  So the free-running comparison runs two steps
    # there, and the third step is checked teacher-forced (both paths from the torch path's clip).
    nl = "i3d" in model_type and "plain" not in model_type
    ns = 2 if nl else 3
    for cls, kw in ((sign_attacks.FGSM, {}), (sign_attacks.BIM, dict(steps=ns)), (sign_attacks.MIFGSM, dict(steps=ns))):
        a = cls(m, engine=eng, **kw)(vid.to(dev), labels).cpu()
        r = cls(F32(ref).to(dev), engine=eng, **kw)(vid.clone().to(dev), labels).cpu()     # the module lives where the engine does
        assert a.shape == vid.shape
        agree = float(((a - r).abs() < 1e-5).float().mean())
        assert agree > 0.97, (cls.__name__, agree)
        un = a * torch.tensor(sign_attacks.STD).view(1, 3, 1, 1, 1) + torch.tensor(sign_attacks.MEAN).view(1, 3, 1, 1, 1)
        assert un.min() >= -1e-5 and un.max() <= 1 + 1e-5
    if nl:      # teacher-forced third step: the gradient at the torch path's two-step clip, native vs float64 autograd
        r2 = sign_attacks.BIM(F32(ref).to(dev), engine=eng, steps=2)(vid.clone().to(dev), labels).cpu()
        g = sign_attacks.BIM(m, steps=1, engine=eng)._grad(r2.clone().to(dev), labels).cpu().double()
        x = r2.double().to(dev).requires_grad_(True)            # (`F32(ref).to(dev)` moved the shared module)
        gref = torch.autograd.grad(torch.nn.CrossEntropyLoss()(ref.eval()(x), labels.to(dev)), x)[0].cpu()
        assert float((torch.sign(g) == torch.sign(gref)).float().mean()) > 0.999


def check_sgm(eng, dev, model_type):
    """Skip Gradient Method on the native classifier (`i2v_net_set_relu_gain` on the ReLUs `base_attacks.py:511-513` selects by name)
    against the torch-module SGM of this repo -- itself pinned hook for hook against the reference class
    (tests/test_sign_family.py) -- on the float64 torch twin of the same network."""
    thw, K, b = (8, 32, 32), 6, 2
    m, ref = torch_classifier(model_type, thw, 3, K)
    vid = torch.randn(b, 3, *thw, generator=torch.Generator().manual_seed(13))
    labels = torch.tensor([4, 0])
    plain = sign_attacks.BIM(m, steps=1, engine=eng)._grad(vid.to(dev), labels).cpu().double()
    for gamma in (0.5, 0.2):
        atk = sign_attacks.SGM(m, steps=1, gamma=gamma, engine=eng)
        g = atk._grad(vid.to(dev), labels).cpu().double()
        m2, ref2 = torch_classifier(model_type, thw, 3, K)                   # (hooks stay on the module they were put on)
        twin = sign_attacks.SGM(ref2, gamma=gamma, engine=eng)
        assert [n.split("back.")[-1] for n in twin.hooked] and len(twin.hooked) == len(atk.hooked) or "slowfast" in model_type
        x = vid.double().requires_grad_(True)
        gref = torch.autograd.grad(torch.nn.CrossEntropyLoss()(ref2(x), labels), x)[0]
        assert (g - gref).abs().max() <= 2e-4 * gref.abs().max(), (gamma, float((g - gref).abs().max() / gref.abs().max()))
        assert (g - plain).abs().max() > 1e-2 * plain.abs().max()           # the gain does something
    a = sign_attacks.SGM(m, steps=2, engine=eng)(vid.to(dev), labels).cpu()
    assert a.shape == vid.shape and not torch.equal(a, vid)
    return atk


@pytest.mark.parametrize("model_type", ["i3d_plain_resnet50", "i3d_resnet50", "slowfast_resnet50"])
def test_native_sgm_hostsim(model_type):
    from tests.hostsim_util import hostsim_engine
    atk = check_sgm(hostsim_engine(), "cpu", model_type)
    if "i3d" in model_type:
        assert atk.hooked == ["relu", "res_layers.0.1.relu", "res_layers.1.1.relu"]


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", ["i3d_resnet50", "slowfast_resnet50"])
def test_native_sgm_gpu(model_type):
    from i2v_amd import attacks
    check_sgm(attacks.get_engine("cuda:0"), "cuda:0", model_type)


MODELS = ["i3d_resnet50", "slowfast_resnet50"]


@pytest.mark.parametrize("model_type", MODELS)
def test_native_ce_gradient_hostsim(model_type):
    from tests.hostsim_util import hostsim_engine
    check_gradient(hostsim_engine(), "cpu", model_type)


@pytest.mark.parametrize("model_type", MODELS)
def test_native_sign_attacks_hostsim(model_type):
    from tests.hostsim_util import hostsim_engine
    check_attacks(hostsim_engine(), "cpu", model_type)


def test_classifier_needs_a_full_graph():
    with pytest.raises(KeyError):
        video.VideoModel("tpn_resnet50", num_classes=400)
    with pytest.raises(ValueError):
        from tests.hostsim_util import hostsim_engine
        sign_attacks.BIM(video.VideoModel("i3d_resnet50", (8, 32, 32), weight_seed=0, tiny=True), engine=hostsim_engine())


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", MODELS)
def test_native_ce_gradient_gpu(model_type):
    from i2v_amd import attacks
    check_gradient(attacks.get_engine("cuda:0"), "cuda:0", model_type)


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", MODELS)
def test_native_sign_attacks_gpu(model_type):
    from i2v_amd import attacks
    check_attacks(attacks.get_engine("cuda:0"), "cuda:0", model_type)


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", MODELS)
def test_native_classifier_full_size(model_type):
    """One 32 x 224^2 clip through the whole I3D-ResNet-50 / SlowFast-R50 (all four stages; SlowFast: both pathways and the four
    lateral connections) + 400-way head and back: finite, non-trivial gradient, reproducible, and BIM moves the clip within its
    eps box."""
    from i2v_amd import attacks
    eng = attacks.get_engine("cuda:0")
    m = video.VideoModel(model_type, (32, 224, 224), weight_seed=0, num_classes=400)
    vid = torch.randn(1, 3, 32, 224, 224, generator=torch.Generator().manual_seed(1)).clamp(-2, 2)
    atk = sign_attacks.BIM(m, steps=2, engine=eng)
    g = atk._grad(vid.to("cuda:0"), torch.tensor([7]))
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0 and atk.last_logits.shape == (1, 400)
    assert torch.equal(g, atk._grad(vid.to("cuda:0"), torch.tensor([7])))
    adv = atk(vid.to("cuda:0"), torch.tensor([7])).cpu()
    std = torch.tensor(sign_attacks.STD).view(1, 3, 1, 1, 1)
    un = adv * std + torch.tensor(sign_attacks.MEAN).view(1, 3, 1, 1, 1)
    assert un.min() >= -1e-5 and un.max() <= 1 + 1e-5 and not torch.equal(adv, vid)


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", MODELS)
def test_native_classifier_full_architecture_vs_torch(model_type):
    """The FULL-depth I3D-R50 / SlowFast-R50 graphs (every stage, every lateral connection, real channel widths) + 400-way
    head on 32 x 64^2 clips against torch autograd in float64 on the same weights: logits, loss, and the input gradient.
    On a clip this small single last-bit ReLU / arg-max decisions carry up to percents of the gradient (torch's own float32
    run differs from float64 by 1e-6 ... 6e-3 relative L2 depending on the clip, measured), so three clips are run and the
    MEDIAN relative L2 error is held to 5e-3, the worst to 5e-2: an arithmetic defect would show on every clip."""
    from i2v_amd import attacks
    eng = attacks.get_engine("cuda:0")
    thw, K = (32, 64, 64), 400
    m = video.VideoModel(model_type, thw, weight_seed=5, num_classes=K)
    g = m.graph_for(thw)
    W, b = m.head_weights(g)
    back = vm.load_weights(vm.make(model_type, False, full=True), weights.synthetic_state_dict(g, 5)).double()
    labels = torch.tensor([123])
    atk = sign_attacks.BIM(m, steps=1, engine=eng)
    rels = []
    for seed in (21, 22, 23):
        vid = torch.randn(1, 3, *thw, generator=torch.Generator().manual_seed(seed))
        gx = atk._grad(vid.to("cuda:0"), labels).cpu().double()
        x = vid.double().requires_grad_(True)
        f = back(x)
        feats = f if isinstance(f, tuple) else (f,)
        logits = torch.cat([t.mean(dim=(2, 3, 4)) for t in feats], dim=1) @ W.double().t() + b.double()
        loss = torch.nn.CrossEntropyLoss()(logits, labels)
        gref = torch.autograd.grad(loss, x)[0]
        np.testing.assert_allclose(atk.last_logits.cpu().double().numpy(), logits.detach().numpy(), rtol=2e-3, atol=2e-4)
        assert abs(float(atk.last_loss) - float(loss.detach())) < 1e-4 * max(1.0, abs(float(loss.detach())))
        rels.append(float((gx - gref).norm() / gref.norm()))
    # (with the non-local blocks torch's OWN float32 run is at 5.6e-3 / 6.3e-4 / 1.8e-3 of float64 on these three clips, measured on the
    #  CPU; the engine: 5.6e-3 / 1.8e-3 / 6.9e-2 -- one early gate on the third clip)
    nl = "i3d" in model_type
    assert sorted(rels)[1] < (1e-2 if nl else 5e-3) and max(rels) < (1e-1 if nl else 5e-2), rels


@pytest.mark.parametrize("model_type", MODELS)
def test_native_evaluator_classifier_hostsim(model_type):
    """forward-only logits for the evaluator (`reference.py:108-129`) through the C ABI
    (`video.NativeClassifier`) against the float64 torch module on the same weights: rtol 1e-4."""
    from tests.hostsim_util import hostsim_engine
    thw, K = (8, 32, 32), 7
    m, ref = torch_classifier(model_type, thw, 3, K)
    vid = torch.randn(3, 3, *thw, generator=torch.Generator().manual_seed(5))
    clf = video.NativeClassifier(m, engine=hostsim_engine())
    got = clf.to("cpu").eval()(vid)
    want = ref(vid.double()).detach()
    np.testing.assert_allclose(got.double().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    # the same numbers as the attack path's logits
    atk = sign_attacks.BIM(m, steps=1, engine=hostsim_engine())
    atk._grad(vid, torch.tensor([0, 1, 2]))
    assert torch.equal(atk.last_logits, got)


def test_evaluator_with_the_native_factory(tmp_path, monkeypatch):
    """`reference.py --model_factory native` end to end on the host simulation: the native classifiers score saved clips, and the
    prediction of an unperturbed clip equals the torch module's arg-max."""
    import functools
    import reference as evaluator
    from i2v_amd import attacks
    from tests.hostsim_util import hostsim_engine
    monkeypatch.setattr(attacks, "get_engine", lambda *a, **k: hostsim_engine())
    thw, K = (8, 32, 32), 7
    m, ref = torch_classifier("i3d_resnet50", thw, 3, K)
    gen = torch.Generator().manual_seed(9)
    clips = torch.randn(4, 3, *thw, generator=gen)
    pred = ref(clips.double()).argmax(1)
    for k in range(4):
        np.save(tmp_path / f"{int(pred[k]) if k < 3 else (int(pred[k]) + 1) % K}-adv-{k}.npy", clips[k].numpy())
    monkeypatch.setattr(evaluator, "native", functools.partial(evaluator.native, num_classes=K, in_thw=thw, weight_seed=3, tiny=True))
    acc = evaluator.main(["--adv_path", str(tmp_path), "--model_factory", "native", "--models", "i3d_resnet50", "--batch_size", "3"])
    assert abs(acc["i3d_resnet50"] - 75.0) < 1e-4            # three clips carry their own arg-max as the label, one does not
    with pytest.raises(KeyError):
        evaluator.native("tpn_resnet50")


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", MODELS)
def test_native_evaluator_classifier_gpu(model_type):
    from i2v_amd import attacks
    thw, K = (8, 32, 32), 7
    m, ref = torch_classifier(model_type, thw, 3, K)
    vid = torch.randn(3, 3, *thw, generator=torch.Generator().manual_seed(5))
    got = video.NativeClassifier(m, engine=attacks.get_engine("cuda:0"))(vid.to("cuda:0")).cpu()
    np.testing.assert_allclose(got.double().numpy(), ref(vid.double()).detach().numpy(), rtol=1e-4, atol=1e-5)


def tap_twins(model_type, thw, seed, K):
    """The native classifier and its float32 torch twin with the stages under gluoncv's names on the classifier itself (what TAP hooks)."""
    g = graphs.build_video_tiny(model_type, thw, full=True)
    m = video.VideoModel(model_type, thw, weight_seed=seed, tiny=True, num_classes=K)
    W, b = m.head_weights(g)
    back = vm.load_weights(vm.make(model_type, True, full=True), weights.synthetic_state_dict(g, seed)).float()
    twin = vm.StageClassifier(back, W.shape[1], K)
    with torch.no_grad():
        twin.fc.weight.copy_(W); twin.fc.bias.copy_(b)
    return m, twin.eval()


def check_tap(eng, dev, model_type, conv3d):
    """`base_attacks.TAP` with everything behind the C ABI (stages hooked in the planned net, `i2v_tap_distance_f32`, cross-entropy head,
    box-filter smoothness gradient from the depthwise kernels) against the torch-module TAP of this repo -- itself pinned against the
    imported reference class (tests/test_sign_family.py) -- on the twin of the same network: the three logged cost terms of every
    step and the perturbed clip."""
    thw, K = (8, 32, 32), 5
    m, twin = tap_twins(model_type, thw, 4, K)
    twin = twin.to(dev)
    vid = (torch.randn(1, 3, *thw, generator=torch.Generator().manual_seed(21)) * 0.5).to(dev)
    labels = torch.tensor([3])
    params = dict(kernlen=3, temporal_kernlen=3, eta=1e3, conv3d=conv3d, model_type=model_type)
    a_nat = sign_attacks.TAP(m, params, steps=3, engine=eng)
    a_ref = sign_attacks.TAP(twin, params, steps=3, engine=eng)
    out_n, out_r = a_nat(vid.clone(), labels).cpu(), a_ref(vid.clone(), labels).cpu()
    for step in range(3):
        for term, tol in (("ce loss", 2e-4), ("reg_cost", 2e-3), ("distance", 2e-4)):
            got, want = float(np.asarray(a_nat.loss_info[step][term]).reshape(-1)[0]), float(np.asarray(a_ref.loss_info[step][term]).reshape(-1)[0])
            # (from the third step on the two clips differ where a gradient was zero to rounding -- each such pixel a sign step apart)
            assert abs(got - want) <= (tol if step < 2 else 1e-2) * max(1.0, abs(want)), (step, term, got, want)
    first = [float(np.asarray(a_nat.loss_info[s]["distance"]).reshape(-1)[0]) for s in (0, 1)]
    assert first[0] == 0.0 and first[1] > 0
    assert float(((out_n - out_r).abs() < 1e-5).float().mean()) > 0.97
    with pytest.raises(RuntimeError):
        a_nat(torch.cat([vid, vid]), torch.tensor([3, 3]))                  # one clip per call, as in the reference


@pytest.mark.parametrize("conv3d", [True, False])
@pytest.mark.parametrize("model_type", ["i3d_plain_resnet50", "i3d_resnet50", "slowfast_resnet50"])
def test_native_tap_hostsim(model_type, conv3d):
    from tests.hostsim_util import hostsim_engine
    check_tap(hostsim_engine(), "cpu", model_type, conv3d)


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", ["i3d_plain_resnet50", "i3d_resnet50", "slowfast_resnet50"])
def test_native_tap_gpu(model_type):
    from i2v_amd import attacks
    check_tap(attacks.get_engine("cuda:0"), "cuda:0", model_type, True)


@pytest.mark.gpu
@pytest.mark.parametrize("cls_name,kw", [("BIM", {}), ("MIFGSM", {}), ("TIFGSM", dict(momentum=True)), ("TIFGSM3D", {}), ("DIFGSM", dict(momentum=True)),
                                         ("SGM", dict(momentum=True)), ("TAP", {})])
def test_native_path_launches_no_framework_kernel_between_backward_and_sign_step(cls_name, kw):
    """on the native path the steps between the input gradient and the sign step -- layout change, `norm_grads`,
    the L1 form, TI-FGSM's column mean, momentum, TAP's perturbation / sign / regulariser terms -- are library kernels
    (`i2v_grad_post_f32`, `i2v_tap_*_f32`, `i2v_dwconv1d_f32`), not framework elementwise kernels.  A two-step call is traced with
    torch.profiler; between the last backbone kernel of a step and that step's `sign_bim_kernel` no `at::native` kernel may run."""
    from torch.profiler import ProfilerActivity, profile
    from i2v_amd import attacks, video
    import random
    eng = attacks.get_engine("cuda:0")
    thw = (32, 32, 32)
    m = video.VideoModel("slowfast_resnet50", thw, num_classes=5, weight_seed=4, tiny=True)
    vid = (torch.randn(1, 3, *thw, generator=torch.Generator().manual_seed(12)) * 0.5).to("cuda:0")
    labels = torch.tensor([1])
    if cls_name == "TAP":
        atk = sign_attacks.TAP(m, dict(kernlen=3, temporal_kernlen=3, eta=1e3, conv3d=True, model_type="slowfast_resnet50"), steps=2, engine=eng)
    else:
        atk = getattr(sign_attacks, cls_name)(m, steps=2, engine=eng, **kw)
    assert atk.path == "native"
    random.seed(3); torch.manual_seed(3)
    atk(vid.clone(), labels)                                     # plans, autotunes, allocates: outside the trace
    torch.cuda.synchronize()
    random.seed(3); torch.manual_seed(3)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        atk(vid.clone(), labels)
        torch.cuda.synchronize()
    ker = sorted(((e.time_range.start, e.name) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name
                  and "Memset" not in e.name), key=lambda t: t[0])
    names = [n for _, n in ker]
    steps = [i for i, n in enumerate(names) if "sign_bim_kernel" in n]
    assert len(steps) == 2, names
    backbone = ("conv_igemm", "pool", "attn_", "softmax_rows", "addmask", "head_grad")
    for at in steps:
        last_bwd = max(i for i in range(at) if any(k in names[i] for k in backbone))
        between = names[last_bwd + 1:at]
        assert between, (cls_name, "the post-processing kernel is missing")
        assert not [n for n in between if "at::native" in n or "elementwise" in n], (cls_name, between)
