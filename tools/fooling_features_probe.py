#!/usr/bin/env python3
"""Probe behind the calibrated evaluator head of `tools/fooling_parity.py` (round 6): the pooled pre-`fc` features of the native
I3D-NL / SlowFast classifiers on (a) all 400 CLEAN CSV-keyed clips, (b) the HIP attack's clip of every row, (c) the fp32 oracle's clip
of the first `--oracle_rows` rows and (d) the float64 oracle's of the first `--f64` rows -- written to one npz so that the head's
ridge parameter can be studied offline (`tools/fooling_head_study.py`).  Test infrastructure; nothing here is product.

    python tools/fooling_features_probe.py --oracle_rows 64 --f64 8 --out gpurun_out/r6_fool/features.npz
"""
import argparse
import csv
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "image-to-video-i2v-attack_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=400)
    ap.add_argument("--oracle_rows", type=int, default=64)
    ap.add_argument("--f64", type=int, default=8)
    ap.add_argument("--models", default="i3d_resnet50,slowfast_resnet50")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r6_fool", "features.npz"))
    args = ap.parse_args(argv)
    t0 = time.time()
    from oracle import size_parity
    work = tempfile.mkdtemp(prefix="fool_feat_")
    ora_dir = os.path.join(work, "oracle_rows")
    STEPS, LR = 10, 0.005
    procs = size_parity.start_oracle_workers(list(range(args.oracle_rows)), ora_dir, steps=STEPS, lr=LR, f64_rows=list(range(args.f64)))
    try:
        import numpy as np
        import torch
        import reference as ev
        from i2v_amd import attacks
        os.environ["I2V_SYNTHETIC_WEIGHTS"] = "1"
        with open(os.path.join(ROOT, "tests", "golden", "kinetics400_attack_samples.csv")) as fh:
            rows = list(csv.DictReader(fh))[:args.rows]
        labels = [int(r["gt_label"]) for r in rows]
        eng = attacks.get_engine("cuda:0")
        assert eng.capi.i2v_backend() == b"hip:gfx950"
        atk = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=LR, steps=STEPS, weight_seed=0)
        models = [m for m in args.models.split(",") if m]
        clf = {m: ev.native(m) for m in models}
        out = {"labels": np.asarray(labels)}
        feats = {(m, k): [] for m in models for k in ("clean", "hip", "oracle", "oracle64")}
        costs_err = []
        for r0 in range(0, args.rows, 8):
            rs = list(range(r0, min(r0 + 8, args.rows)))
            vids = torch.cat([size_parity.synthetic_clip(1000 + r) for r in rs])
            adv = atk(vids, torch.tensor([labels[r] for r in rs]), [rows[r]["path"] for r in rs])
            for m in models:
                feats[(m, "clean")].append(clf[m].pooled_features(vids).cpu().numpy())
                feats[(m, "hip")].append(clf[m].pooled_features(adv).cpu().numpy())
            if r0 % 80 == 0:
                print(f"[probe] hip rows {r0}.. done, {time.time() - t0:.0f} s", flush=True)
        for r in range(args.oracle_rows):
            ora = size_parity.wait_oracle_row(ora_dir, r, procs, timeout=3600, lr=LR)
            for m in models:
                feats[(m, "oracle")].append(clf[m].pooled_features(ora["adv"]).cpu().numpy())
            os.remove(os.path.join(ora_dir, f"{r}-oracle-adv.npy"))
            costs_err.append(ora["costs"])
        for r in range(args.f64):
            o64 = size_parity.wait_oracle_row(ora_dir, r, procs, timeout=3600, lr=LR, tag="oracle64")
            for m in models:
                feats[(m, "oracle64")].append(clf[m].pooled_features(o64["adv"]).cpu().numpy())
        for (m, k), v in feats.items():
            if v:
                out[f"{m}__{k}"] = np.concatenate(v, 0)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        np.savez_compressed(args.out, **out)
        print("[probe] wrote", args.out, {k: v.shape for k, v in out.items()}, f"{time.time() - t0:.0f} s")
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            p.wait()
        import shutil
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
