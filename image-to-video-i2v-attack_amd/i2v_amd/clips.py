"""Clip sources for the CLI.  The reference decodes Kinetics-400 mp4s with decord
(`/root/reference/datasets.py`, out of scope: SURVEY.md 8(f) N4); here a batch is either

  * synthetic: 8-bit uniform-noise clips seeded by the sample's row index, labelled
    from the reference's sample list format `path,gt_label,clip_index`, or
  * a directory of `{label}-ori.npy` float32 (3,32,224,224) normalised clips (the format the
    reference's own `attack.py` writes next to `{label}-adv.npy`).

Each item mirrors the reference's validation item `(clip, label, name)` (`datasets.py:138-150`)."""
import csv
import glob
import math
import os

import numpy as np
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def synthetic_clip(seed: int, frames=32, hw=224) -> torch.Tensor:
    gen = torch.Generator().manual_seed(1000 + seed)
    u8 = torch.randint(0, 256, (3, frames, hw, hw), generator=gen, dtype=torch.uint8)
    mean = torch.tensor(MEAN).view(3, 1, 1, 1)
    std = torch.tensor(STD).view(3, 1, 1, 1)
    return (u8.float() / 255 - mean) / std


def sample_list(csv_path=None, n=400):
    """[(name, label)] -- from a `path,gt_label,clip_index` csv when given, else n synthetic rows."""
    if csv_path and os.path.exists(csv_path):
        with open(csv_path) as fh:
            return [(r["path"], int(r["gt_label"])) for r in csv.DictReader(fh)][:n]
    return [(f"synthetic/{i:03d}.mp4", i) for i in range(n)]


def batches(batch_size, csv_path=None, clip_dir=None, frames=32, hw=224, n=400, workers=0):
    """Yields (val_batch (b,3,f,h,w), val_label (b,), video_names) like the reference's DataLoader.  A `clip_dir` of
    `{label}-raw.npy` files -- DECODED uint8 frames (t,H,W,3), what decord hands the reference's loader
    (datasets.py:226-244) -- yields uint8 batches (b,t,H,W,3) instead: the caller runs the validation transform on the
    device (`Engine.clip_resize_crop`, datasets.py:86-93).  Raw clips of one batch must share their frame size."""
    raw = sorted(glob.glob(os.path.join(clip_dir, "*-raw.npy"))) if clip_dir else []
    if raw:
        items = [(os.path.basename(p), int(os.path.basename(p).split("-")[0]), p) for p in raw]
    elif clip_dir:
        files = sorted(glob.glob(os.path.join(clip_dir, "*-ori.npy")))
        items = [(os.path.basename(p), int(os.path.basename(p).split("-")[0]), p) for p in files]
    else:
        items = [(name, label, None) for name, label in sample_list(csv_path, n)]
    def load(s):
        chunk = items[s:s + batch_size]
        clips = [torch.from_numpy(np.load(p)) if p else synthetic_clip(s + i, frames, hw)
                 for i, (_, _, p) in enumerate(chunk)]
        return torch.stack(clips), torch.tensor([c[1] for c in chunk]), [c[0] for c in chunk]
    # `workers` loader threads, a window of 2 x workers batches in flight, results in order (np.load and the synthetic
    # generator release the GIL): what the reference's DataLoader workers are for
    yield from _prefetched(load, list(range(0, len(items), batch_size)), workers)


def num_batches(batch_size, csv_path=None, clip_dir=None, n=400):
    if clip_dir:
        cnt = len(glob.glob(os.path.join(clip_dir, "*-raw.npy"))) or len(glob.glob(os.path.join(clip_dir, "*-ori.npy")))
    else:
        cnt = len(sample_list(csv_path, n))
    return (cnt + batch_size - 1) // batch_size


# ---------------------------------------------------------------------------------------------------------------
# geometry of the reference's validation transform (datasets.py:86-93; gluoncv.torch.data.video_transforms), host side
# of `Engine.clip_resize_crop`.  gluoncv and OpenCV are third-party to the reference and not installed here: restated
# from their public sources, parity UNPINNED (oracle/restate.py:resize_center_crop_normalise is the CPU twin).
# ---------------------------------------------------------------------------------------------------------------
def resize_sizes(h, w, short_side):
    """`video_transforms.Resize(size: int)`: the short side becomes `size`, the other int(size * long / short);
    an image whose short side already is `size` is left alone."""
    if (w <= h and w == short_side) or (h <= w and h == short_side):
        return h, w
    if w < h:
        return int(short_side * h / w), short_side
    return short_side, int(short_side * w / h)


def center_crop_origin(h, w, ch, cw):
    """`video_transforms.CenterCrop`: (y1, x1) = round((im - crop) / 2)."""
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))


def resize_table(n_out, n_in):
    """Per output index of a bilinear axis resize as cv::resize (INTER_LINEAR, 8-bit) builds it: source index,
    weight of it and of its right/lower neighbour in INTER_RESIZE_COEF_SCALE = 2048 units (int32 (n_out, 3)).
        fx = (float)((d + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx
        sx < 0 -> sx = 0, fx = 0;  sx >= n_in - 1 -> sx = n_in - 1, fx = 0
        weights = saturate_cast<short>(rint((1 - fx) * 2048)), saturate_cast<short>(rint(fx * 2048))"""
    scale = np.float64(n_in) / np.float64(n_out)
    d = np.arange(n_out, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo, hi = sx < 0, sx >= n_in - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, n_in - 1, sx))
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    return np.stack([sx, a0, a1], 1).astype(np.int32)


# ---------------------------------------------------------------------------------------------------------------
# geometry of the UCF-101 loader's validation transform (`dataset_ucf101.py:113-126`, `transforms_ucf101.py`):
# Scale(224) = PIL `Image.resize(..., BILINEAR)` to short side 224, CornerCrop(224, 'c'), ToTensor (/255), Normalize.
# Pillow IS installed here, so the restatement of its resampler (libImaging/Resample.c: antialiased two-pass convolution,
# 8-bit path with 22-bit fixed-point coefficients) is pinned bit for bit against the library itself
# (tests/test_pil_resample.py); host side of `Engine.clip_resample_crop`.
# ---------------------------------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2


def scale_sizes(h, w, size):
    """`transforms_ucf101.Scale(size: int)` (:271-289): the smaller edge becomes `size`, the other int(size * long / short)."""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def corner_crop_center_origin(h, w, size):
    """`CornerCrop(size, 'c')` (:343-348): x1 = int(round((W - size) / 2.)), y1 likewise (Python 3 banker's rounding)."""
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def pil_resample_table(n_in, n_out):
    """Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` for the BILINEAR filter over the full axis: per output index the
    first source index, the number of taps, and the taps as 22-bit fixed-point integers.  Returns (bounds int32 (n_out, 2),
    coeffs int32 (n_out, ksize), ksize).  Downscaling widens the triangle filter by the scale factor (antialiasing)."""
    scale = float(n_in) / float(n_out)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                         # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((n_out, 2), np.int32)
    coeffs = np.zeros((n_out, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(n_out):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)              # C cast: truncation
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > n_in:
            xmax = n_in
        xmax -= xmin
        w = []
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - t if t < 1.0 else 0.0)
        ww = sum(w)                                     # left-to-right double sum, as the C loop
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            coeffs[xx, x] = int(-0.5 + k * (1 << PIL_PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs, ksize


# ---------------------------------------------------------------------------------------------------------------
# UCF-101 jpg-frame folders (`/root/reference/dataset_ucf101.py`): the clip list, the temporal selection and the
# decoding -- everything in front of the loader's spatial transform, which runs on the device (`Engine.clip_resample_crop`).
# Decoding is Pillow's (installed here, and what the reference calls, :14-18), so the reader is pinned by construction.
# (The Kinetics-400 half of N4 -- decord reading mp4s, `datasets.py:188-244` -- is NOT buildable in this image: no decord,
# PyAV, OpenCV or ffmpeg.  `--clip_dir` of `{label}-raw.npy` decoded frames is that half's entry point.)
# ---------------------------------------------------------------------------------------------------------------
def loop_padding(frame_indices, size=32):
    """`transforms_ucf101.LoopPadding(size)` (:23-40), quirks included: the selection starts at the SECOND entry
    (`frame_indices[1:size+1]`), and a clip shorter than `size + 1` frames is padded by cycling through that selection (the
    reference appends to the list it is iterating over, which walks the growing list from its start)."""
    out = list(frame_indices[1:size + 1])
    k = 0
    while len(out) < size and k < len(out):
        out.append(out[k])
        k += 1
    return out


def _load_index_list(path):
    """The reference's `used_idxs.pkl` (a pickled list of ints, `dataset_ucf101.py:59-63`) -- read with an unpickler that
    admits NO globals (a list of ints needs none; anything else in the file is refused) --, or a .json / whitespace-separated
    text file holding the same list."""
    if path.endswith(".pkl"):
        import pickle

        class _NoGlobals(pickle.Unpickler):
            def find_class(self, module, name):
                raise pickle.UnpicklingError(f"{path}: only a plain list of integers is accepted (found {module}.{name})")
        with open(path, "rb") as fh:
            idx = _NoGlobals(fh).load()
    elif path.endswith(".json"):
        import json
        with open(path) as fh:
            idx = json.load(fh)
    else:
        with open(path) as fh:
            idx = fh.read().split()
    return [int(i) for i in idx]


def ucf101_clip_list(setting, image_root, used_idxs=None):
    """`attack_ucf101._make_dataset` (:81-99) + the index selection (:59-64): [(frame directory, duration, label)] from the
    lines `video_path duration label` of `test01_setting.txt`, reduced to the rows `used_idxs` names (a path to the index list,
    a list of ints, or None = every row)."""
    if not os.path.exists(setting):
        raise RuntimeError("Setting file %s doesn't exist. Check opt.train-list and opt.val-list. " % setting)      # :83
    rows = []
    with open(setting) as fh:
        for line in fh.readlines():
            info = line.split()
            if len(info) < 3:
                raise RuntimeError("Video input format is not correct, missing one or more element. %s" % line)     # :91
            rows.append((os.path.join(image_root, info[0]), int(info[1]), int(info[2])))
    if used_idxs is None:
        return rows
    idx = _load_index_list(used_idxs) if isinstance(used_idxs, str) else [int(i) for i in used_idxs]
    return [rows[i] for i in idx]


def load_frame_folder(directory, frame_indices):
    """`video_loader` + `pil_loader` (:14-18, 36-45): `image_{:05d}.jpg` of every index, decoded by Pillow and converted to
    RGB; the list ENDS at the first missing file (as the reference's loader returns what it has).  uint8 (t,H,W,3)."""
    from PIL import Image
    frames = []
    for i in frame_indices:
        path = os.path.join(directory, "image_{:05d}.jpg".format(i))
        if not os.path.exists(path):
            break
        with open(path, "rb") as fh:
            with Image.open(fh) as img:
                frames.append(np.asarray(img.convert("RGB")))
    if not frames:
        raise FileNotFoundError(f"no frames under {directory} (expected image_{{:05d}}.jpg)")
    return np.stack(frames)


def ucf101_batches(batch_size, setting, image_root, used_idxs=None, frames=32, workers=0):
    """The UCF-101 loader (`attack_genearte_dataeset`, :103-110) up to its spatial transform: yields (uint8 (b,t,H,W,3) decoded
    frames, labels (b,), names) in list order; the caller runs Scale / CornerCrop / ToTensor / Normalize on the device.  `workers`
    decoder threads (Pillow releases the GIL while decoding) with 2 x workers batches in flight -- the reference's DataLoader
    `num_workers=9`.  Clips of one batch must share their frame size (UCF-101: 240 x 320) and length (`frames`, LoopPadding)."""
    rows = ucf101_clip_list(setting, image_root, used_idxs)

    def load(s):
        chunk = rows[s:s + batch_size]
        clips_ = [load_frame_folder(d, loop_padding(list(range(1, dur + 1)), frames)) for d, dur, _ in chunk]     # :66-72
        return (torch.from_numpy(np.stack(clips_)), torch.tensor([t for _, _, t in chunk]),
                [os.path.basename(d) for d, _, _ in chunk])
    starts = list(range(0, len(rows), batch_size))
    yield from _prefetched(load, starts, workers)


def ucf101_num_batches(batch_size, setting, image_root, used_idxs=None):
    return (len(ucf101_clip_list(setting, image_root, used_idxs)) + batch_size - 1) // batch_size


def _prefetched(load, starts, workers):
    """`load(start)` for every start, in order, on `workers` threads with a window of 2 x workers results in flight."""
    if workers <= 0:
        for s in starts:
            yield load(s)
        return
    import collections
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as ex:
        window = collections.deque()
        it = iter(starts)
        for s in it:
            window.append(ex.submit(load, s))
            if len(window) >= 2 * workers:
                break
        while window:
            out = window.popleft().result()
            nxt = next(it, None)
            if nxt is not None:
                window.append(ex.submit(load, nxt))
            yield out
