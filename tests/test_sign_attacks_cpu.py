"""CPU (host simulation backend): the BIM / MI-FGSM drop-in classes against what the reference's
classes returned for the same toy video model (fixture `sign_step.npz`, oracle/make_golden.py),
and ILAF live against the imported reference."""
import numpy as np
import pytest
import torch

from oracle import ref_shim
from tests import golden_util as gu
from tests.hostsim_util import hostsim_engine


def toy_video_model():
    torch.manual_seed(3)                      # same construction as oracle/make_golden.py:run_sign_step
    return torch.nn.Sequential(torch.nn.Conv3d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool3d(1),
                               torch.nn.Flatten(), torch.nn.Linear(4, 5))


@pytest.mark.parametrize("cls", ["BIM", "MIFGSM"])
def test_sign_attack_classes_match_reference(cls):
    import base_attacks
    fx = gu.load("sign_step")
    vid = gu.videos_of(fx)
    atk = getattr(base_attacks, cls)(toy_video_model(), epsilon=16 / 255, steps=4, engine=hostsim_engine())
    adv = atk(vid.clone(), torch.tensor([2]))
    ref = torch.from_numpy(fx[cls + "_adv"])
    # identical update arithmetic; the model's own gradient is torch on both sides
    assert (adv - ref).abs().max() < 1e-6
    assert (adv != ref).float().mean() < 1e-3


@pytest.mark.skipif(not ref_shim.available(), reason="reference checkout not present")
def test_ilaf_matches_reference_live():
    ia = ref_shim.import_reference("image_attacks")
    import i2v_amd.sign_attacks as sa
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv3d(3, 6, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv3d(6, 4, 3, padding=1),
                                torch.nn.AdaptiveAvgPool3d(1), torch.nn.Flatten())
    model.layer2 = model[1]                   # 'tpn' branch of _find_target_layer (image_attacks.py:518-519)
    gen = torch.Generator().manual_seed(9)
    ori = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, 4, 16, 16), generator=gen, dtype=torch.uint8).numpy()})
    adv0 = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, 4, 16, 16), generator=gen, dtype=torch.uint8).numpy()})
    adv0 = ori + 0.1 * (adv0 - ori)
    with ref_shim.quiet():
        ref_atk = ia.ILAF(model, "tpn", step_size=0.005, steps=3)
        ref = ref_atk(adv0.clone(), ori.clone(), torch.zeros(1, dtype=torch.long), ["v"]).detach()
    ours = sa.ILAF(model, "tpn", step_size=0.005, steps=3, engine=hostsim_engine())
    got = ours(adv0.clone(), ori.clone(), torch.zeros(1, dtype=torch.long), ["v"]).detach()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 1e-6
    assert ours.loss_info["v"][0]["cost"] == ref_atk.loss_info["v"][0]["cost"]
    c_ours = np.array([float(ours.loss_info["v"][i]["cost"]) for i in range(3)])
    c_ref = np.array([float(ref_atk.loss_info["v"][i]["cost"]) for i in range(3)])
    np.testing.assert_allclose(c_ours, c_ref, rtol=1e-3)
