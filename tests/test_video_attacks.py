"""`video_attacks.TemporalTranslation` (`/root/reference/video_attacks.py:14-229`; what `attack.py --attack_type video` runs):
live against the imported reference class on one and the same torch classifier (CPU; skipped where /root/reference is absent),
and with the NATIVE classifier against the class driving a float64 torch module -- host simulation here, HIP kernels in the
`gpu`-marked twins."""
import os

import numpy as np
import pytest
import torch

from i2v_amd import sign_attacks, video_attacks as va
from oracle import ref_shim
from tests.test_native_classifier import torch_classifier

needs_ref = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference is absent")


class F32(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.p = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return self.net(x).float()


@needs_ref
@pytest.mark.parametrize("mode,kernlen", [("gaussian", 15), ("gaussian", 5), ("linear", 7), ("random", 9)])
def test_temporal_kernels_match_reference(mode, kernlen):
    ref = ref_shim.import_reference("video_attacks").TemporalTranslation
    fn = {"gaussian": ref._initial_kernel_gaussian, "linear": ref._initial_kernel_linear, "random": ref._initial_kernel_uniform}[mode]
    want = fn(None, kernlen).astype(np.float32)
    got = va.TemporalTranslation._temporal_kernel(mode, kernlen).astype(np.float32)
    assert np.array_equal(got, want)


@needs_ref
@pytest.mark.parametrize("move_type,momentum,weight,kernlen", [("adj", False, 1.0, 5), ("adj", True, 0.4, 5), ("large", False, 0.7, 15)])
def test_temporal_translation_matches_reference(move_type, momentum, weight, kernlen):
    """Same float32 classifier module on both sides; ours mixes the gradients and takes the step through the C ABI (host
    simulation).  The reference's 1 x D matmul may add in another order than the kernel's fmaf chain, so a few pixels whose
    mixed gradient is ~0 may step the other way."""
    from tests.hostsim_util import hostsim_engine
    refmod = ref_shim.import_reference("video_attacks")
    thw, K = (32, 32, 32), 5
    _, net = torch_classifier("i3d_resnet50", thw, 4, K)
    model = F32(net).eval()
    params = {"kernlen": kernlen, "momentum": momentum, "weight": weight, "move_type": move_type, "kernel_mode": "gaussian"}
    vid = torch.randn(1, 3, *thw, generator=torch.Generator().manual_seed(3)) * 0.5
    labels = torch.tensor([2])
    with ref_shim.quiet():
        want = refmod.TemporalTranslation(model, dict(params), steps=3)(vid.clone(), labels)
    got = va.TemporalTranslation(model, dict(params), steps=3, engine=hostsim_engine())(vid.clone(), labels)
    assert got.shape == want.shape == vid.shape
    same = float(((got - want).abs() < 1e-6).float().mean())
    assert same > 0.999, same
    std = torch.tensor(sign_attacks.STD).view(1, 3, 1, 1, 1)
    assert float(((got - want).abs() * std).max()) <= 2 * (16 / 255) / 3 + 1e-6


def check_native(eng, dev, model_type):
    thw, K = (32, 32, 32), 5
    m, ref = torch_classifier(model_type, thw, 4, K)
    vid = torch.randn(1, 3, *thw, generator=torch.Generator().manual_seed(13)) * 0.5
    labels = torch.tensor([3])
    params = {"kernlen": 5, "momentum": True, "weight": 0.5, "move_type": "adj", "kernel_mode": "linear"}
    # (the I3D graphs carry the reference's non-local blocks: with synthetic weights their saturated attention makes the SECOND eps/2
    # step chaotic -- tests/test_native_classifier.py:check_attacks has the measurement -- so that model takes one step here)
    ns = 1 if ("i3d" in model_type and "plain" not in model_type) else 2
    a = va.TemporalTranslation(m, dict(params), steps=ns, engine=eng)(vid.to(dev), labels).cpu()
    r = va.TemporalTranslation(F32(ref).to(dev), dict(params), steps=ns, engine=eng)(vid.clone().to(dev), labels).cpu()
    # the mix of D shifted gradients cancels more often than a single gradient: a few % of the pixels have a mixed gradient whose SIGN fp32
    # and float64 backbones decide differently (BIM alone: > 97 %, tests/test_native_classifier.py)
    assert a.shape == vid.shape and float(((a - r).abs() < 1e-5).float().mean()) > 0.94
    un = a * torch.tensor(sign_attacks.STD).view(1, 3, 1, 1, 1) + torch.tensor(sign_attacks.MEAN).view(1, 3, 1, 1, 1)
    assert un.min() >= -1e-5 and un.max() <= 1 + 1e-5 and not torch.equal(a, vid)


@pytest.mark.parametrize("model_type", ["i3d_resnet50", "i3d_plain_resnet50", "slowfast_resnet50"])
def test_temporal_translation_native_classifier_hostsim(model_type):
    from tests.hostsim_util import hostsim_engine
    check_native(hostsim_engine(), "cpu", model_type)


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", ["i3d_resnet50", "slowfast_resnet50"])
def test_temporal_translation_native_classifier_gpu(model_type):
    from i2v_amd import attacks
    check_native(attacks.get_engine("cuda:0"), "cuda:0", model_type)


@pytest.mark.gpu
def test_tt_grad_mix_gpu_matches_hostsim_bitwise():
    from i2v_amd import attacks
    from tests.hostsim_util import hostsim_engine
    g = torch.randn(7, 1, 3, 32, 14, 14, generator=torch.Generator().manual_seed(5))
    k = va.TemporalTranslation._temporal_kernel("gaussian", 7).astype(np.float32)
    mv = list(range(-3, 4))
    a = attacks.get_engine("cuda:0").tt_grad_mix(g.to("cuda:0"), k, mv, 0.3).cpu()
    b = hostsim_engine().tt_grad_mix(g, k, mv, 0.3)
    assert torch.equal(a, b)


def test_attack_cli_video_type(tmp_path, monkeypatch):
    """`attack.py --attack_type video --attack_method TemporalTranslation` end to end (tiny native classifier on the host
    simulation): `{label}-adv.npy` / `{label}-ori.npy` pairs as `attack.py:86-96` writes them."""
    from i2v_amd import attacks, video
    from tests.hostsim_util import hostsim_engine
    monkeypatch.setitem(attacks._ENGINES, attacks.default_device(), hostsim_engine())
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    orig = video.VideoModel.__init__
    monkeypatch.setattr(video.VideoModel, "__init__", lambda self, *a, **k: orig(self, *a, **dict(k, tiny=True, weight_seed=0)))
    import importlib
    import attack
    importlib.reload(attack)
    out = attack.main(["--attack_type", "video", "--attack_method", "TemporalTranslation", "--model", "i3d_resnet50", "--model_factory", "native",
                       "--step", "1", "--kernlen", "5", "--num_clips", "2", "--batch_size", "1", "--frames", "32", "--hw", "32", "--num_classes", "7",
                       "--file_prefix", "t"])
    assert sorted(os.listdir(out)) == ["0-adv.npy", "0-ori.npy", "1-adv.npy", "1-ori.npy"]
    adv, ori = np.load(os.path.join(out, "0-adv.npy")), np.load(os.path.join(out, "0-ori.npy"))
    assert adv.shape == ori.shape == (3, 32, 32, 32) and not np.array_equal(adv, ori)
    with pytest.raises(UnboundLocalError):       # any other video attack name: `spe_params` is never bound (attack.py:78-82)
        attack.main(["--attack_type", "video", "--attack_method", "Nope", "--model", "i3d_resnet50", "--model_factory", "native", "--num_clips", "1",
                     "--frames", "32", "--hw", "32", "--file_prefix", "t"])
