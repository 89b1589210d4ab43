"""CPU, build container only: re-run the IMPORTED reference classes live (oracle/ref_shim.py)
and compare with the oracle.  Skipped wherever /root/reference is absent (the GPU box)."""
import numpy as np
import pytest
import torch

from oracle import ref_shim, restate
from tests import golden_util as gu

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference checkout not present")


def _clip(seed, b, f, hw, dtype):
    from oracle.make_golden import make_clip, normalise
    return normalise(make_clip(seed, b, f, hw), dtype)


@pytest.mark.parametrize("model,depth", [("resnet", 1), ("resnet", 4), ("vgg", 3), ("vgg", 4),
                                         ("alexnet", 1), ("alexnet", 4), ("squeezenet", 1),
                                         ("squeezenet", 4)])
def test_i2v_live_f64(model, depth):
    """Hook depths not covered by the committed fixtures."""
    ref_shim.FACTORY.tiny, ref_shim.FACTORY.seed = True, 5
    ref_shim.FACTORY.in_hw, ref_shim.FACTORY.dtype = (64, 64), torch.float64
    ia = ref_shim.import_reference("image_attacks")
    vid = _clip(31, 1, 2, 64, torch.float64)
    with ref_shim.quiet():
        atk = ia.ImageGuidedFMDirection_Adam([model], depth=depth, step_size=0.004, steps=3)
        with ref_shim.AdamTap() as tap:
            adv = atk(vid.clone(), torch.zeros(1, dtype=torch.long), ["v"])
    fx = dict(models=[model], depth=depth, hw=64, wseed=5)
    (g, sd, hooks), = gu.hook_lists(fx)
    net = restate.OracleNet(g, sd, hooks, dtype=torch.float64)
    out = restate.run_attack([net], vid, steps=3, step_size=0.004, trace=True)
    ref_cost = np.array([float(atk.loss_info["v"][i]["cost"]) for i in range(3)])
    np.testing.assert_allclose(out["costs"], ref_cost, rtol=2e-6)
    assert (tap.grad0.double() - out["grad0"]).abs().max() <= 2e-6 * tap.grad0.abs().max()
    assert (tap.deltas[-1].double() - out["deltas"][-1]).abs().max() < 5e-6
    assert (adv.detach() - out["adv"]).abs().max() < 5e-5
    assert adv.shape == vid.shape                                  # (b,3,f,h,w), image_attacks.py:362-363


def test_aens_coeffs_persist_across_calls():
    """`self.coeffs` lives on the object (TPAMI_attack.py:165,265)."""
    ref_shim.FACTORY.tiny, ref_shim.FACTORY.seed = True, 0
    ref_shim.FACTORY.in_hw, ref_shim.FACTORY.dtype = (64, 64), torch.float64
    tp = ref_shim.import_reference("TPAMI_attack")
    vid = _clip(41, 1, 2, 64, torch.float64)
    depths = {"resnet": [2, 3], "squeezenet": [2, 3]}
    with ref_shim.quiet():
        atk = tp.AENS_I2V_MF(["resnet", "squeezenet"], depths, step_size=0.005, momentum=1.0, steps=2)
        atk(vid.clone(), torch.zeros(1, dtype=torch.long), ["v"])
        w1 = np.stack(atk.weights)
        atk(vid.clone(), torch.zeros(1, dtype=torch.long), ["v"])
        w2 = np.stack(atk.weights)
    fx = dict(models=["resnet", "squeezenet"], depth=depths, hw=64, wseed=0)
    nets = [restate.OracleNet(g, sd, h, dtype=torch.float64) for g, sd, h in gu.hook_lists(fx)]
    c = torch.ones(4, dtype=torch.float64)
    o1 = restate.run_attack(nets, vid, steps=2, step_size=0.005, mode="aens", coeffs=c, momentum=1.0)
    o2 = restate.run_attack(nets, vid, steps=2, step_size=0.005, mode="aens", coeffs=o1["coeffs"],
                            momentum=1.0)
    np.testing.assert_allclose(np.stack(o1["weights"]), w1, rtol=1e-6)
    np.testing.assert_allclose(np.stack(o2["weights"]), w2, rtol=1e-6)


def test_aens_squeezenet_hooks_whole_fire_module_live():
    """With LIST depths the adaptive attack hooks `features[idx]` itself -- cat(expand1x1, expand3x3) -- not the 3x3
    branch the scalar-depth lookup takes (TPAMI_attack.py:195-199).  Gradient and output of the imported reference
    class against the oracle on the product's graph (`Graph.hook_for(d, whole_module=True)`); the coefficient
    weights alone would not notice the difference while cos ~ 1."""
    ref_shim.FACTORY.tiny, ref_shim.FACTORY.seed = True, 3
    ref_shim.FACTORY.in_hw, ref_shim.FACTORY.dtype = (64, 64), torch.float64
    tp = ref_shim.import_reference("TPAMI_attack")
    vid = _clip(52, 2, 2, 64, torch.float64)
    depths = {"squeezenet": [2, 3], "alexnet": [2, 3]}
    with ref_shim.quiet():
        atk = tp.AENS_I2V_MF(["squeezenet", "alexnet"], depths, step_size=0.005, momentum=0.5, steps=3)
        with ref_shim.AdamTap() as tap:
            adv, _, cost_saved = atk(vid.clone(), torch.zeros(2, dtype=torch.long), ["a", "b"])
    shapes = [tuple(a.shape[1:]) for a in atk.activations["value"]]          # last model's hooks: alexnet
    fx = dict(models=["squeezenet", "alexnet"], depth=depths, hw=64, wseed=3)
    lists = gu.hook_lists(fx)
    g, _, hooks = lists[0]
    e1 = g.tensors[g.hooks[2]].C
    assert [g.tensors[h].C for h in hooks] == [2 * e1, 2 * g.tensors[g.hooks[3]].C]      # both Fire branches
    nets = [restate.OracleNet(gg, sd, h, dtype=torch.float64) for gg, sd, h in lists]
    out = restate.run_attack(nets, vid, steps=3, step_size=0.005, mode="aens", coeffs=torch.ones(4, dtype=torch.float64),
                             momentum=0.5, trace=True)
    np.testing.assert_allclose(out["costs"], cost_saved, rtol=2e-6)
    assert (tap.grad0.double() - out["grad0"]).abs().max() <= 2e-6 * tap.grad0.abs().max()
    assert (adv.detach() - out["adv"]).abs().max() < 5e-5
    np.testing.assert_allclose(np.stack(out["weights"]), np.stack(atk.weights), rtol=1e-6)
    assert len(shapes) == 2
