"""GPU (-m gpu): the DI / TI device kernels (`i2v_resample_nearest_f32`, `i2v_resample_nearest_bwd_f32`, `i2v_dwconv1d_f32`) through the
C ABI on the MI355X, bit for bit against their scalar restatement (tests/hostsim) at the full clip size, and the attack classes
end to end on the HIP engine against the reference fixture."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from i2v_amd import attacks, sign_attacks as sa  # noqa: E402
from tests import golden_util as gu  # noqa: E402
from tests.hostsim_util import hostsim_engine  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return attacks.get_engine("cuda:0")


def test_resample_and_smoothing_kernels_bit_exact(eng):
    host = hostsim_engine()
    x = torch.randn(1, 3, 8, 224, 224, generator=torch.Generator().manual_seed(1))
    my, ylo, yhi = sa.diversity_maps(224, 239, 4)
    mx, xlo, xhi = sa.diversity_maps(224, 239, 9)
    t = torch.from_numpy
    d = lambda a: t(a).to("cuda:0")                                          # noqa: E731
    got = eng.resample_nearest(x.to("cuda:0"), d(my), d(mx)).cpu()
    assert torch.equal(got, host.resample_nearest(x, t(my), t(mx)))
    g = torch.randn(got.shape, generator=torch.Generator().manual_seed(2))
    assert torch.equal(eng.resample_nearest_bwd(g.to("cuda:0"), (224, 224), (d(ylo), d(yhi), d(xlo), d(xhi))).cpu(),
                       host.resample_nearest_bwd(g, (224, 224), (t(ylo), t(yhi), t(xlo), t(xhi))))
    taps = sa.gaussian_taps()
    v = torch.randn(1, 3, 32, 56, 56, generator=torch.Generator().manual_seed(3))
    for axis in (4, 3, 2):
        assert torch.equal(eng.dwconv1d(v.to("cuda:0"), taps, axis).cpu(), host.dwconv1d(v, taps, axis)), axis


@pytest.mark.parametrize("cls,kw", [("DIFGSM", {}), ("TIFGSM", {"momentum": True}), ("TIFGSM3D", {}), ("SIM", {"momentum": True})])
def test_sign_family_on_the_hip_engine(eng, cls, kw):
    import random
    import base_attacks
    from tests.test_sign_attacks_cpu import toy_video_model
    fx = gu.load("sign_family")
    key = cls + ("_m" if kw.get("momentum") else "")
    vid = gu.videos_of({"clip_u8": fx["clip_di_u8"] if cls == "DIFGSM" else fx["clip_u8"]})
    model = toy_video_model().to("cuda:0")
    random.seed(11); torch.manual_seed(11)
    adv = getattr(base_attacks, cls)(model, epsilon=16 / 255, steps=int(fx["steps"]), engine=eng, **kw)(vid.clone(), torch.tensor([2])).cpu()
    ref = torch.from_numpy(fx[key + "_adv"])
    # the toy model's gradient comes from torch on the GPU here (cuDNN-class kernels, not the CPU's): held to sign agreement
    assert float((adv != ref).float().mean()) < 2e-2
    assert np.isfinite(adv.numpy()).all()
