#!/usr/bin/env python3
"""Round 6 (GPU box): what the launch overlap of `mark_overlap` (projection shortcuts on a side stream) is worth -- frames/s of the
headline attack (ResNet-50 layer3, 10 steps, 32 x 224^2 clips) for 1 and 4 clips per call with I2V_OVERLAP_MAX_FRAMES = 0 / 32 / 128,
each in a fresh attack object (the knob is read per plan), alternated twice.     python tools/overlap_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-to-video-i2v-attack_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from i2v_amd import attacks  # noqa: E402


def run(clips, knob, reps):
    os.environ["I2V_OVERLAP_MAX_FRAMES"] = knob
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10, weight_seed=0)
    atk.clip_lanes = 1
    vid = torch.randn(clips, 3, 32, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
    lab = torch.zeros(clips, dtype=torch.long)
    names = [f"c{i}" for i in range(clips)]
    atk(vid, lab, names); atk(vid, lab, names)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        atk(vid, lab, names)
    torch.cuda.synchronize()
    return reps * clips * 32 / (time.perf_counter() - t0)


if __name__ == "__main__":
    for rnd in range(2):
        for clips, knobs in ((1, ("0", "32")), (4, ("0", "128"))):
            for k in knobs:
                print(f"round {rnd} clips {clips} I2V_OVERLAP_MAX_FRAMES={k}: {run(clips, k, 10 if clips == 1 else 5):.1f} frames/s", flush=True)
