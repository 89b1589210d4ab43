"""CPU: real-checkpoint readiness of `i2v_amd.weights` -- the loader against checkpoints with EXACTLY the key set of the
torchvision 0.10.1 models the reference downloads (`image_attacks.py:88-101`; manifest restated in oracle/tv_manifest.py,
committed as tests/golden/torchvision_0_10_1_state_dict_keys.json), the explicit opt-in for synthetic weights, and the
gluoncv -> graph key converter for the ILAF white-box models."""
import json
import os

import pytest
import torch

from i2v_amd import graphs, weights
from oracle import tv_manifest

NAMES = {"resnet50": "resnet50", "resnet": "resnet101", "vgg": "vgg16", "alexnet": "alexnet", "squeezenet": "squeezenet1_1",
         "densenet121": "densenet121", "densenet161": "densenet161"}


def test_committed_manifest_is_the_generators_output():
    assert json.load(open(tv_manifest.PATH)) == tv_manifest.manifest()


@pytest.mark.parametrize("name", list(NAMES))
def test_graph_parameters_exist_in_torchvision_layout(name):
    """Every parameter the graph IR reads exists in the torchvision state_dict of that architecture with the same shape
    (the IR is truncated at the deepest hook, the checkpoint is not)."""
    man = json.load(open(tv_manifest.PATH))[NAMES[name]]
    g = graphs.build(name, (224, 224))
    assert g.arch == NAMES[name]
    shapes = g.param_shapes()
    assert shapes and all(k in man and list(shp) == man[k] for k, shp in shapes.items()), \
        [k for k, shp in shapes.items() if k not in man or list(shp) != man[k]][:5]
    extra = set(man) - set(shapes)
    assert any(k.endswith("num_batches_tracked") for k in extra) or name in ("vgg", "alexnet", "squeezenet")
    assert any(k.startswith(("fc.", "classifier.")) for k in extra)


@pytest.mark.parametrize("name", ["resnet50", "vgg", "squeezenet", "densenet121"])
def test_load_state_dict_digests_a_full_torchvision_checkpoint(name, tmp_path, monkeypatch):
    man = json.load(open(tv_manifest.PATH))[NAMES[name]]
    g = graphs.build(name, (224, 224))
    want = weights.synthetic_state_dict(g, 7)
    sd = {}
    for k, shp in man.items():        # every torchvision key; the unused ones as 1-element expanded tensors (vgg16's classifier is 400 MB)
        sd[k] = want[k].clone() if k in want else (torch.zeros(()).expand(shp) if shp else torch.tensor(0))
    torch.save(sd, tmp_path / f"{g.arch}.pth")
    monkeypatch.setenv("I2V_WEIGHTS_DIR", str(tmp_path))
    monkeypatch.delenv("I2V_SYNTHETIC_WEIGHTS", raising=False)
    got = weights.load_state_dict(g)                     # no seed, no opt-in: must come from the file
    assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want)
    assert weights.SOURCES[g.arch].endswith(f"{g.arch}.pth")
    # nested {'state_dict': ...} checkpoints load too; a missing or mis-shaped parameter is an error
    torch.save({"state_dict": sd}, tmp_path / f"{g.arch}.pth")
    assert set(weights.load_state_dict(g)) == set(want)
    k0 = next(iter(want))
    bad = dict(sd); bad.pop(k0)
    torch.save(bad, tmp_path / f"{g.arch}.pth")
    with pytest.raises(KeyError):
        weights.load_state_dict(g)
    bad = dict(sd); bad[k0] = torch.zeros(3)
    torch.save(bad, tmp_path / f"{g.arch}.pth")
    with pytest.raises(ValueError):
        weights.load_state_dict(g)


def test_synthetic_weights_need_an_explicit_opt_in(tmp_path, monkeypatch):
    """a missing checkpoint must not silently become random weights (the CLIs would write valid-looking
    `*-adv.npy` files optimised against noise)."""
    g = graphs.build_tiny("resnet", (64, 64))
    monkeypatch.setenv("I2V_WEIGHTS_DIR", str(tmp_path))
    monkeypatch.delenv("I2V_SYNTHETIC_WEIGHTS", raising=False)
    with pytest.raises(weights.MissingWeights):
        weights.load_state_dict(g)
    assert set(weights.load_state_dict(g, 3)) == set(g.param_shapes())          # explicit seed
    monkeypatch.setenv("I2V_SYNTHETIC_WEIGHTS", "1")
    assert set(weights.load_state_dict(g)) == set(g.param_shapes())             # explicit environment opt-in
    from i2v_amd import attacks
    monkeypatch.delenv("I2V_SYNTHETIC_WEIGHTS", raising=False)
    from tests.hostsim_util import hostsim_engine
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=2, step_size=0.005, steps=1, engine=hostsim_engine(),
                                              graph_builder=graphs.build_tiny)
    with pytest.raises(weights.MissingWeights):
        atk(torch.zeros(1, 3, 2, 64, 64), torch.zeros(1, dtype=torch.long), ["v"])


@pytest.mark.parametrize("model_type", ["i3d_resnet50", "slowfast_resnet50"])
def test_gluoncv_key_converter(model_type):
    g = graphs.build_video(model_type, (32, 224, 224))
    want = weights.synthetic_state_dict(g, 1)
    ckpt = {"module." + k: v for k, v in want.items()}
    for k in list(want):
        if k.endswith("running_mean"):
            ckpt["module." + k.replace("running_mean", "num_batches_tracked")] = torch.tensor(0)
    ckpt["module.fc.weight"] = torch.zeros(400, 2048)                               # head: kept
    ckpt["module.res_layers.9.0.conv1.weight"] = torch.zeros(1)                     # a stage the graph does not build: dropped
    got = weights.convert_gluoncv_state_dict(g, {"state_dict": ckpt})
    assert set(got) == set(want) | {"fc.weight"} and all(torch.equal(got[k], want[k]) for k in want)
    got.pop("fc.weight")
    # a parameter INSIDE a stage the graph builds that the graph does not read (an i3d_nl5 non-local block
    # offered to the plain I3D graph) is a different network, not an ignorable extra
    stage = next(k for k in want if k.count(".") >= 3).rsplit(".", 2)[0]
    with pytest.raises(KeyError, match="different network"):
        weights.convert_gluoncv_state_dict(g, dict(ckpt, **{"module." + stage + ".nl_theta.weight": torch.zeros(4, 4, 1, 1, 1)}))
    # a checkpoint whose names differ needs rules; without them the converter fails loudly
    if "i3d" in model_type:
        odd = {k.replace("res_layers.", "layer"): v for k, v in want.items()}
    else:
        odd = {k.replace("fast_res2.", "fast.res2."): v for k, v in want.items()}
    with pytest.raises(KeyError):
        weights.convert_gluoncv_state_dict(g, odd)
    rules = [(r"^layer(\d+)\.", r"res_layers.\1.")] if "i3d" in model_type else [(r"^fast\.res2\.", "fast_res2.")]
    assert set(weights.convert_gluoncv_state_dict(g, odd, rules)) == set(want)
    # the converted dict is what VideoModel takes
    from i2v_amd import video
    vm = video.VideoModel(model_type, (32, 224, 224), state_dict=got)
    assert vm.state_dict_for(g) is got


def test_classifier_head_travels_with_its_backbone(tmp_path, monkeypatch):
    """`attack.py --model_factory native` builds VideoModel(..., num_classes=K) without a state_dict; the head
    must come from the same `$I2V_WEIGHTS_DIR/<arch>.pth` as the backbone, and a real backbone is never paired with a
    synthetic `fc`."""
    from i2v_amd import video
    vm = video.VideoModel("i3d_resnet50", (8, 32, 32), tiny=True, num_classes=5)
    g = vm.graph_for((8, 32, 32))
    sd = weights.synthetic_state_dict(g, 4)
    C_ = sum(g.tensors[t].C for t in vm.classifier_hook(g))
    monkeypatch.setenv("I2V_WEIGHTS_DIR", str(tmp_path))
    monkeypatch.delenv("I2V_SYNTHETIC_WEIGHTS", raising=False)
    torch.save(sd, tmp_path / f"{g.arch}.pth")                       # backbone only
    with pytest.raises(KeyError, match="never paired"):
        video.VideoModel("i3d_resnet50", (8, 32, 32), tiny=True, num_classes=5).head_weights(g)
    fcw, fcb = torch.randn(5, C_), torch.randn(5)
    torch.save(dict(sd, **{"fc.weight": fcw, "fc.bias": fcb}), tmp_path / f"{g.arch}.pth")
    W, b = video.VideoModel("i3d_resnet50", (8, 32, 32), tiny=True, num_classes=5).head_weights(g)
    assert torch.equal(W, fcw) and torch.equal(b, fcb)
    # synthetic backbone (explicit seed, no checkpoint): synthetic head is allowed
    monkeypatch.setenv("I2V_WEIGHTS_DIR", str(tmp_path / "none"))
    W, b = video.VideoModel("i3d_resnet50", (8, 32, 32), tiny=True, num_classes=5, weight_seed=2).head_weights(g)
    assert tuple(W.shape) == (5, C_)
