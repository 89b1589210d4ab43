"""CPU: DenseNet-121 (extension backbone, BASELINE.json configs[2]).  The reference cannot hook DenseNet
(`_find_target_layer` has no branch, image_attacks.py:260-271), so there is nothing of the reference to pin
against: the graph IR + oracle are checked against torch autograd through an independently written
torchvision-layout module (oracle/tv_models.py:DenseNet), hook = `features.denseblock{d}` output."""
import pytest
import torch

from i2v_amd import graphs, weights
from oracle import restate, tv_models


@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_densenet_oracle_matches_autograd(depth):
    g = graphs.build_tiny("densenet121", (64, 64))
    sd = weights.synthetic_state_dict(g, 2)
    m = tv_models.load_backbone_weights(tv_models.make("densenet121", True), sd).double().eval()
    feats = []
    getattr(m.features, f"denseblock{depth}").register_forward_hook(lambda mod, i, o: feats.append(o))
    x = torch.randn(3, 3, 64, 64, dtype=torch.float64, requires_grad=True)
    m(x)
    onet = restate.OracleNet(g, sd, [g.hooks[depth]], dtype=torch.float64)
    of = onet.forward(x.detach())[0]
    assert torch.allclose(of, feats[0].detach(), rtol=1e-9, atol=1e-10)
    hg = torch.randn_like(of)
    (feats[0] * hg).sum().backward()
    assert torch.allclose(onet.backward([hg]), x.grad, rtol=1e-8, atol=1e-10)


def test_densenet121_shapes():
    g = graphs.build("densenet121")
    assert [(g.tensors[g.hooks[d]].C, g.tensors[g.hooks[d]].H) for d in (1, 2, 3, 4)] == [(256, 56), (512, 28), (1024, 14), (1024, 7)]
    g = graphs.build("densenet161")
    assert [(g.tensors[g.hooks[d]].C, g.tensors[g.hooks[d]].H) for d in (1, 2, 3, 4)] == [(384, 56), (768, 28), (2112, 14), (2208, 7)]
    with pytest.raises(AttributeError):
        graphs.build("densenet")          # the reference's own name keeps the reference's behaviour
