#!/usr/bin/env python
"""Developer tool: A/B the headline attack (4 clips x 32 x 224^2, ResNet-50 layer3, 10 steps) between engine variants
selected by PLAN-TIME environment knobs (e.g. I2V_GATES=0), in ONE process with interleaved rounds (cdna guide rule 24:
separate invocations add cross-process / cross-device variance that looks like a kernel property).

    python tools/ab_bench.py "I2V_GATES=1" "I2V_GATES=0" [--rounds 6] [--lanes 1]
"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "image-to-video-i2v-attack_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from i2v_amd import attacks  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+", help='each: "K=V,K2=V2" environment set while that variant PLANS its nets')
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--lanes", type=int, default=1)
ap.add_argument("--clips", type=int, default=4)
args = ap.parse_args()
dev = "cuda:0"
torch.cuda.set_device(0)
eng = attacks.get_engine(dev)
vid = bench.synthetic_clips(args.clips).to(dev)
lab = torch.zeros(args.clips, dtype=torch.long)
names = [f"c{i}" for i in range(args.clips)]
atks = []
for v in args.variants:
    env = dict(kv.split("=") for kv in v.split(",") if kv)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    a = attacks.ImageGuidedFMDirection_Adam([bench.MODEL], depth=bench.DEPTH, step_size=0.005, steps=bench.ATTACK_STEPS, engine=eng, weight_seed=0)
    a.clip_lanes = args.lanes
    a(vid, lab, names)          # plans (and autotunes) under this variant's environment
    torch.cuda.synchronize()
    for k, o in old.items():
        if o is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = o
    atks.append(a)
times = [[] for _ in atks]
for r in range(args.rounds):
    for i, a in enumerate(atks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a(vid, lab, names)
        torch.cuda.synchronize()
        times[i].append(time.perf_counter() - t0)
frames = args.clips * bench.FRAMES
for v, t in zip(args.variants, times):
    print(f"{v:40s} median {frames / statistics.median(t):8.1f} fps   best {frames / min(t):8.1f}   all {[round(frames / x, 1) for x in t]}")
