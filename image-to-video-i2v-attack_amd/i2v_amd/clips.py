"""Clip sources for the CLI.  The reference decodes Kinetics-400 mp4s with decord
(`/root/reference/datasets.py`, out of scope: SURVEY.md 8(f) N4); here a batch is either

  * synthetic: 8-bit uniform-noise clips seeded by the sample's row index, labelled
    from the reference's sample list format `path,gt_label,clip_index`, or
  * a directory of `{label}-ori.npy` float32 (3,32,224,224) normalised clips (the format the
    reference's own `attack.py` writes next to `{label}-adv.npy`).

Each item mirrors the reference's validation item `(clip, label, name)` (`datasets.py:138-150`)."""
import csv
import glob
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


def batches(batch_size, csv_path=None, clip_dir=None, frames=32, hw=224, n=400):
    """Yields (val_batch (b,3,f,h,w), val_label (b,), video_names) like the reference's DataLoader."""
    if clip_dir:
        files = sorted(glob.glob(os.path.join(clip_dir, "*-ori.npy")))
        items = [(os.path.basename(p), int(os.path.basename(p).split("-")[0]), p) for p in files]
    else:
        items = [(name, label, None) for name, label in sample_list(csv_path, n)]
    for s in range(0, len(items), batch_size):
        chunk = items[s:s + batch_size]
        clips = [torch.from_numpy(np.load(p)) if p else synthetic_clip(s + i, frames, hw)
                 for i, (_, _, p) in enumerate(chunk)]
        yield torch.stack(clips), torch.tensor([c[1] for c in chunk]), [c[0] for c in chunk]


def num_batches(batch_size, csv_path=None, clip_dir=None, n=400):
    if clip_dir:
        cnt = len(glob.glob(os.path.join(clip_dir, "*-ori.npy")))
    else:
        cnt = len(sample_list(csv_path, n))
    return (cnt + batch_size - 1) // batch_size
