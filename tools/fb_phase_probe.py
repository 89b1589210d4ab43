#!/usr/bin/env python3
"""Round 6 probe (GPU box): where a fused fast-pathway block spends its time -- the plan-time autotuner's own timing of `fast_block2_kernel`
(printed under I2V_FUSE_DEBUG) with parts of the kernel switched off by the probe bits of I2V_FB_DELAY (results are then wrong: timing only;
nothing is executed after the plan).  The probe bits exist in the EXPERIMENTAL build only:
    python __graft_entry__.py --experimental && I2V_LIB=image-to-video-i2v-attack_amd/i2v_amd/libi2v_hip_exp.so python tools/fb_phase_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "image-to-video-i2v-attack_amd")); sys.path.insert(0, %r)
import torch
from i2v_amd import sign_attacks, video
m = video.VideoModel("slowfast_resnet50", (32, 224, 224), weight_seed=0)
a = sign_attacks.ILAF(m, "slowfast_resnet50")
a.independent_clips = True
a.plan_for(torch.empty(4, 3, 32, 224, 224))
''' % (ROOT, ROOT)
for flags, what in ((0, "whole kernel"), (0x100, "no stage-A rows"), (0x200, "no stage-B rows"), (0x400, "no stage C"), (0x300, "no A, no B rows"),
                    (0x500, "no A rows, no C"), (0x600, "no B rows, no C"), (0x700, "decode + epilogues + gates only")):
    env = dict(os.environ, I2V_FB_DELAY=str(flags), I2V_FUSE_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    rows = sorted(set(l[l.index("[i2v fastblock]"):] for l in r.stderr.splitlines() if "[i2v fastblock]" in l and (" 128 frames" in l or " 64 frames" in l)))
    print(f"== I2V_FB_DELAY={flags:#x}: {what}")
    for l in rows:
        print("   ", l.split("] ", 1)[1])
    if r.returncode or not rows:
        print(r.stdout[-1500:], r.stderr[-1500:])
