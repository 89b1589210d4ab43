"""N4, UCF-101 half: the loader's validation transform (`/root/reference/dataset_ucf101.py:113-126`: PIL BILINEAR `Scale(224)` ->
`CornerCrop(224,'c')` -> `ToTensor` -> `Normalize`) as ONE device kernel (`i2v_clip_resample_crop_u8_f32`).  The arithmetic is
Pillow's resampler; Pillow is installed, so the oracle (`oracle/restate.ucf101_transform`) runs `Image.resize` itself and the
comparison is BIT-EXACT: host tables (`clips.pil_resample_table`) + kernel vs Pillow."""
import numpy as np
import pytest
import torch

from i2v_amd import clips
from oracle import ref_shim, restate

SHAPES = [(240, 320), (256, 340), (100, 60), (224, 300), (37, 53), (480, 360)]      # UCF-101 frames are 240 x 320


def frames(shape, b=1, t=3, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (b, t, shape[0], shape[1], 3), dtype=np.uint8)


@pytest.mark.parametrize("shape", SHAPES)
def test_hostsim_kernel_equals_pillow(shape):
    from tests.hostsim_util import hostsim_engine
    f = frames(shape, b=2, t=2, seed=shape[0])
    crop = min(224, min(clips.scale_sizes(shape[0], shape[1], 224)))
    got = hostsim_engine().clip_resample_crop(torch.from_numpy(f), 224, crop)
    want = restate.ucf101_transform(f, 224, crop)
    assert got.shape == want.shape == (2, 3, 2, crop, crop)
    assert torch.equal(got, want)


def test_upscale_and_identity_sizes():
    """A short side of exactly `size` is left alone (transforms_ucf101.py:280-281); smaller frames are upscaled (plain two-tap
    bilinear: the filter is only widened when shrinking)."""
    from tests.hostsim_util import hostsim_engine
    eng = hostsim_engine()
    for shape, size, crop in (((64, 80), 64, 64), ((20, 33), 48, 40)):
        f = frames(shape, t=2, seed=7)
        assert torch.equal(eng.clip_resample_crop(torch.from_numpy(f), size, crop), restate.ucf101_transform(f, size, crop))


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference is absent")
def test_oracle_equals_the_reference_transform_classes():
    """The oracle against the reference's OWN Scale / CornerCrop / ToTensor / Normalize objects composed as `test_transform()`
    composes them (dataset_ucf101.py:113-126) and stacked as `__getitem__` stacks them (:75-79)."""
    from PIL import Image
    tr = ref_shim.import_reference("transforms_ucf101")
    chain = [tr.Scale(224), tr.CornerCrop(224, "c"), tr.ToTensor(), tr.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])]
    f = frames((240, 320), t=4, seed=11)
    clip = []
    for ti in range(4):
        img = Image.fromarray(f[0, ti])
        for op in chain:
            img = op(img)
        clip.append(img)
    want = torch.stack(clip, 0).permute(1, 0, 2, 3)
    assert torch.equal(restate.ucf101_transform(f, 224, 224)[0], want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(240, 320), (100, 60), (480, 360)])
def test_gpu_kernel_equals_pillow(shape):
    from i2v_amd import attacks
    eng = attacks.get_engine("cuda:0")
    f = frames(shape, b=2, t=4, seed=shape[1])
    crop = min(224, min(clips.scale_sizes(shape[0], shape[1], 224)))
    got = eng.clip_resample_crop(torch.from_numpy(f).to("cuda:0"), 224, crop).cpu()
    assert torch.equal(got, restate.ucf101_transform(f, 224, crop))


@pytest.mark.gpu
def test_gpu_full_clip_ucf101_size():
    """A whole 32-frame UCF-101 clip (240 x 320 -> 224 x 298 -> centre 224^2) on the device, bit-exact against Pillow."""
    from i2v_amd import attacks
    f = frames((240, 320), b=1, t=32, seed=3)
    got = attacks.get_engine("cuda:0").clip_resample_crop(torch.from_numpy(f).to("cuda:0")).cpu()
    assert got.shape == (1, 3, 32, 224, 224) and torch.equal(got, restate.ucf101_transform(f))
