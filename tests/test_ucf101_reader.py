"""N4, UCF-101 half, the part in FRONT of the device transform: jpg frame folders -> clip list -> LoopPadding(32) -> Pillow decoding
(`/root/reference/dataset_ucf101.py:14-45,52-99,113-126`, `transforms_ucf101.py:23-40`), `i2v_amd.clips.ucf101_*`, and the whole
loader (reader + `Engine.clip_resample_crop`) against the oracle's transform of independently decoded frames -- bit for bit -- and,
where /root/reference exists, against the reference's OWN `attack_ucf101` dataset object on the same folder."""
import json
import os
import pickle

import numpy as np
import pytest
import torch
from PIL import Image

from i2v_amd import clips
from oracle import ref_shim, restate


def write_dataset(root, durations=(40, 5, 33, 12), hw=(60, 80), seed=3):
    """A folder of synthetic jpg frames in the UCF-101 layout + `test01_setting.txt` + `used_idxs.pkl`; returns
    (image_root, setting, used_idxs path, rows)."""
    rng = np.random.default_rng(seed)
    image_root = os.path.join(root, "jpegs")
    rows = []
    for k, dur in enumerate(durations):
        rel = f"Class{k % 2}/v_Class{k % 2}_g01_c{k:02d}"
        os.makedirs(os.path.join(image_root, rel))
        base = rng.integers(0, 256, (hw[0] // 4, hw[1] // 4, 3), dtype=np.uint8)       # smooth-ish content: jpeg-friendly
        for i in range(1, dur + 1):
            img = Image.fromarray(base).resize((hw[1], hw[0]), Image.BICUBIC)
            arr = np.asarray(img).astype(np.int16) + rng.integers(-20, 21, (hw[0], hw[1], 3))
            Image.fromarray(arr.clip(0, 255).astype(np.uint8)).save(os.path.join(image_root, rel, "image_{:05d}.jpg".format(i)), quality=90)
        rows.append((rel, dur, 10 + k))
    setting = os.path.join(root, "test01_setting.txt")
    with open(setting, "w") as fh:
        fh.writelines(f"{rel} {dur} {lab}\n" for rel, dur, lab in rows)
    used = os.path.join(root, "used_idxs.pkl")
    with open(used, "wb") as fh:
        pickle.dump([2, 0, 1], fh)
    return image_root, setting, used, rows


def independent_decode(image_root, rel, indices):
    return np.stack([np.asarray(Image.open(os.path.join(image_root, rel, "image_{:05d}.jpg".format(i))).convert("RGB")) for i in indices])


def test_loop_padding_quirks():
    """`LoopPadding(32)` (transforms_ucf101.py:23-40): starts at the SECOND frame, cycles through its own selection."""
    assert clips.loop_padding(list(range(1, 166))) == list(range(2, 34))
    assert clips.loop_padding(list(range(1, 34))) == list(range(2, 34))                     # 33 frames: exactly enough
    assert clips.loop_padding(list(range(1, 6)), 10) == [2, 3, 4, 5, 2, 3, 4, 5, 2, 3]
    assert clips.loop_padding([1, 2], 4) == [2, 2, 2, 2]
    assert clips.loop_padding([1], 4) == []                                                   # nothing to cycle through


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference is absent")
def test_loop_padding_equals_the_reference_class():
    tr = ref_shim.import_reference("transforms_ucf101")
    for n in (0, 1, 2, 3, 5, 16, 31, 32, 33, 34, 100):
        for size in (1, 4, 32):
            assert clips.loop_padding(list(range(1, n + 1)), size) == tr.LoopPadding(size)(list(range(1, n + 1))), (n, size)


def test_clip_list_and_index_files(tmp_path):
    image_root, setting, used, rows = write_dataset(str(tmp_path), durations=(3, 2, 4), hw=(16, 20))
    full = clips.ucf101_clip_list(setting, image_root)
    assert full == [(os.path.join(image_root, rel), dur, lab) for rel, dur, lab in rows]
    assert clips.ucf101_clip_list(setting, image_root, used) == [full[2], full[0], full[1]]           # the pickle's order
    assert clips.ucf101_clip_list(setting, image_root, [1]) == [full[1]]
    (tmp_path / "idx.json").write_text(json.dumps([1, 1]))
    (tmp_path / "idx.txt").write_text("0 2\n")
    assert clips.ucf101_clip_list(setting, image_root, str(tmp_path / "idx.json")) == [full[1], full[1]]
    assert clips.ucf101_clip_list(setting, image_root, str(tmp_path / "idx.txt")) == [full[0], full[2]]
    with pytest.raises(RuntimeError, match="doesn't exist"):                                             # dataset_ucf101.py:83
        clips.ucf101_clip_list(str(tmp_path / "nope.txt"), image_root)
    (tmp_path / "bad.txt").write_text("only two\n")
    with pytest.raises(RuntimeError, match="missing one or more element"):                               # :91
        clips.ucf101_clip_list(str(tmp_path / "bad.txt"), image_root)
    # an index pickle is data, not code: anything but plain integers is refused without being executed
    with open(tmp_path / "evil.pkl", "wb") as fh:
        pickle.dump([os.path.join], fh)
    with pytest.raises(pickle.UnpicklingError):
        clips.ucf101_clip_list(setting, image_root, str(tmp_path / "evil.pkl"))


def test_reader_decodes_what_pillow_decodes(tmp_path):
    image_root, setting, used, rows = write_dataset(str(tmp_path))
    got = list(clips.ucf101_batches(1, setting, image_root, used, frames=8, workers=2))
    assert [int(l) for _, lab, _ in got for l in lab] == [12, 10, 11] and got[0][2] == ["v_Class0_g01_c02"]
    for (frames, _, _), k in zip(got, (2, 0, 1)):
        rel, dur, _ = rows[k]
        want = independent_decode(image_root, rel, clips.loop_padding(list(range(1, dur + 1)), 8))
        assert frames.dtype == torch.uint8 and tuple(frames.shape) == (1, 8, 60, 80, 3)
        assert np.array_equal(frames[0].numpy(), want)
    # a missing frame ends the clip there, as the reference's loader returns what it has (dataset_ucf101.py:36-45)
    os.remove(os.path.join(image_root, rows[0][0], "image_00005.jpg"))
    short = clips.load_frame_folder(os.path.join(image_root, rows[0][0]), [2, 3, 4, 5, 6])
    assert short.shape[0] == 3
    # batches of two: the same clips, stacked, in order, with or without decoder threads
    two = list(clips.ucf101_batches(2, setting, image_root, [1, 3], frames=4, workers=0))
    assert len(two) == 1 and tuple(two[0][0].shape) == (2, 4, 60, 80, 3) and two[0][1].tolist() == [11, 13]


def whole_loader(eng, tmp_path, frames=6, hw=48):
    image_root, setting, used, rows = write_dataset(str(tmp_path))
    out = []
    for raw, lab, _ in clips.ucf101_batches(1, setting, image_root, used, frames=frames, workers=2):
        clip = eng.clip_resample_crop(raw.to(eng.device), hw, hw).cpu()
        k = int(lab[0]) - 10
        rel, dur, _ = rows[k]
        want = restate.ucf101_transform(independent_decode(image_root, rel, clips.loop_padding(list(range(1, dur + 1)), frames))[None], hw, hw)
        assert clip.shape == want.shape == (1, 3, frames, hw, hw) and torch.equal(clip, want), k
        out.append(clip)
    return out


def test_whole_loader_bit_exact_on_the_host_simulation(tmp_path):
    from tests.hostsim_util import hostsim_engine
    whole_loader(hostsim_engine(), tmp_path)


@pytest.mark.gpu
def test_whole_loader_bit_exact_on_the_device(tmp_path):
    from i2v_amd import attacks
    eng = attacks.get_engine("cuda:0")
    assert eng.capi.i2v_backend() == b"hip:gfx950"
    whole_loader(eng, tmp_path, frames=32, hw=224)


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference is absent")
def test_reader_equals_the_reference_dataset_object(tmp_path, monkeypatch):
    """The reference's own `attack_ucf101` (`dataset_ucf101.py:52-99`) with its `test_transform()` (:113-126), run in a directory
    holding `./test01_setting.txt` and `./used_idxs.pkl` as it expects, on the same jpg folders: item by item the clip tensor and the
    label equal reader + oracle transform (and so, by the tests above, reader + device kernel)."""
    image_root, setting, used, rows = write_dataset(str(tmp_path), durations=(36, 7, 33), hw=(240, 320))
    ds_mod = ref_shim.import_reference("dataset_ucf101")
    monkeypatch.setattr(ds_mod, "UCF_IMAGE_ROOT", image_root)
    monkeypatch.chdir(tmp_path)
    spa, tem = ds_mod.test_transform()
    ds = ds_mod.attack_ucf101(spatial_transform=spa, temporal_transform=tem, get_loader=lambda: __import__("functools").partial(
        ds_mod.video_loader, image_loader=ds_mod.pil_loader))
    ours = list(clips.ucf101_batches(1, setting, image_root, used, frames=32, workers=0))
    assert len(ds) == len(ours) == 3
    for i in range(3):
        clip, target = ds[i]
        assert target == int(ours[i][1][0])
        assert torch.equal(restate.ucf101_transform(ours[i][0].numpy(), 224, 224)[0], clip), i


def test_image_main_ucf101_reads_frame_folders(tmp_path, monkeypatch):
    """`image_main_ucf101.py --frame_dir ... --setting ... --used_idxs ...`: the reference's own inputs end to end on the tiny engine."""
    import importlib
    from tests.hostsim_util import hostsim_engine
    from i2v_amd import attacks, graphs
    monkeypatch.setitem(attacks._ENGINES, attacks.default_device(), hostsim_engine())
    monkeypatch.setattr(graphs, "build", graphs.build_tiny)
    monkeypatch.setenv("I2V_SYNTHETIC_WEIGHTS", "1")
    image_root, setting, used, rows = write_dataset(str(tmp_path / "data"), durations=(5, 9, 4))
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    import image_main
    import image_main_ucf101
    importlib.reload(image_main); importlib.reload(image_main_ucf101)
    image_main_ucf101.main(["--attack_method", "ImageGuidedFMDirection_Adam", "--step", "2", "--step_size", "0.005", "--depth", "2",
                            "--direction_image_model", "resnet", "--frame_dir", image_root, "--setting", setting, "--used_idxs", used,
                            "--frames", "4", "--hw", "48", "--file_prefix", "f", "--group_clips", "1"])
    out = tmp_path / "Image-ImageGuidedFMDirection_Adam-2-f"
    assert sorted(os.listdir(out)) == ["10-adv.npy", "11-adv.npy", "12-adv.npy", "loss_info_1.json"]
    adv = np.load(out / "11-adv.npy")
    clean = restate.ucf101_transform(independent_decode(image_root, rows[1][0], clips.loop_padding(list(range(1, 10)), 4))[None], 48, 48)[0].numpy()
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1, 1)
    assert adv.shape == clean.shape == (3, 4, 48, 48) and 0 < np.abs((adv - clean) * std).max() <= 16 / 255 + 1e-5
