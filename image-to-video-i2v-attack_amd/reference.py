#!/usr/bin/env python
"""Drop-in for the evaluation harness `/root/reference/reference.py`: loads the
saved `*adv*` clips of an attack run, scores them with black-box video models and writes
`top1_acc_all_models.json` + `results_all_models_prediction.csv` next to them.  Fooling rate =
100 - top-1, because every evaluated clip was classified correctly before the attack.

Kept from the reference: `--adv_path/--gpu/--batch_size`, file discovery (`'adv' in name`), label =
`int(fname.split('-')[0])` (`reference.py:43,96-97`), top-1 (`:28-36`), the prediction re-ordering by
label (`:116-119`) and both output files (`:127-129`).  Not kept: joining `adv_path` with `OPT_PATH`
twice (`:21-25`) -- `$I2V_OPT_PATH` is joined once.

The six gluoncv Kinetics-400 models of the reference are not vendored, so models come from a factory:
`--model_factory pkg.module:function`, a callable `name -> torch.nn.Module` mapping a normalised
(b,3,f,h,w) batch to logits; `--models a,b,c` are the names handed to it (default: the reference's
six names).  `--model_factory native` scores with the NATIVE classifiers (`i2v_amd.video.NativeClassifier`: I3D / SlowFast graph IR +
head behind the C ABI).  The built-in factory `proxy` is a seeded random 3-D conv classifier: useless as an
accuracy number, but it lets two sets of adversarial clips (reference-oracle vs HIP) be scored by the
same code, optionally against the model's own clean prediction (`--clean_dir`)."""
import argparse
import importlib
import json
import math
import os

import numpy as np
import torch

DEFAULT_MODELS = ["i3d_resnet50", "i3d_resnet101", "slowfast_resnet50", "slowfast_resnet101", "tpn_resnet50",
                  "tpn_resnet101"]          # keys of CONFIG_PATHS, /root/reference/utils.py:8-15


def proxy(name: str, num_classes: int = 400) -> torch.nn.Module:
    seed = sum(ord(c) for c in name)
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv3d(3, 16, (3, 7, 7), stride=(1, 4, 4), padding=(1, 3, 3)), torch.nn.ReLU(),
                               torch.nn.Conv3d(16, 32, 3, stride=(2, 2, 2), padding=1), torch.nn.ReLU(),
                               torch.nn.AdaptiveAvgPool3d(1), torch.nn.Flatten(), torch.nn.Linear(32, num_classes))


def native(name: str, num_classes: int = 400, **model_kwargs):
    """`--model_factory native`: the evaluated video model runs in libi2v_hip.so (`i2v_amd.video.NativeClassifier`: graph IR to
    the last stage + pool + fc, forward only).  Built for the names whose graphs reach their last stage -- the I3D ResNets and
    the whole SlowFast (`video.VideoModel(num_classes=...)`); `tpn_*` raises KeyError (neck / head not built).  Weights:
    `$I2V_WEIGHTS_DIR/<arch>.pth` with its `fc.*`, or -- only under I2V_SYNTHETIC_WEIGHTS=1 -- the seeded synthetic initialiser."""
    from i2v_amd.video import NativeClassifier, VideoModel
    return NativeClassifier(VideoModel(name, num_classes=num_classes, **model_kwargs))


def accuracy(output, target):
    _, pred = output.topk(1, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return correct[:1].reshape(-1).float().sum(0) * (100.0 / target.size(0)), pred.reshape(-1)


def evaluate(model, adv_path, files_batch, device, clean_dir=None):
    predictions, labels, hit, seen = [], [], 0.0, 0
    with torch.no_grad():
        for batch in files_batch:
            clips = torch.stack([torch.from_numpy(np.load(os.path.join(adv_path, f))) for f in batch]).to(device)
            if clean_dir:       # score against the model's own clean prediction
                clean = torch.stack([torch.from_numpy(np.load(os.path.join(clean_dir, f.replace("adv", "ori"))))
                                     for f in batch]).to(device)
                target = model(clean).argmax(1)
            else:
                target = torch.tensor([int(f.split("-")[0]) for f in batch], device=device)
            prec, preds = accuracy(model(clips), target)
            predictions += list(preds.cpu().numpy())
            labels += [int(f.split("-")[0]) for f in batch]
            hit += prec.item() * len(batch)
            seen += len(batch)
    return predictions, labels, hit / max(seen, 1)


def main(argv=None, ucf101=False):
    """`ucf101`: the UCF-101 twin (`/root/reference/reference_ucf101.py`, drop-in module of that name): 101 classes (`:125`) and
    fine-tuned checkpoints from its own directory (`MODEL_TO_CKPTS`, `:24-31`) -- `$I2V_UCF_CKPT_PATH` takes the place of
    `$I2V_WEIGHTS_DIR` for the native factory.  Everything else is the same code."""
    ap = argparse.ArgumentParser(description="")
    ap.add_argument("--adv_path", type=str, default="", help="the path of adversarial examples.")
    ap.add_argument("--gpu", type=str, default="0", help="gpu device.")
    ap.add_argument("--batch_size", type=int, default=16, metavar="N")
    ap.add_argument("--models", type=str, default=",".join(DEFAULT_MODELS))
    ap.add_argument("--model_factory", type=str, default="proxy", help="'proxy', 'native' (i2v_amd.video.NativeClassifier) or pkg.module:function")
    ap.add_argument("--clean_dir", type=str, default="", help="directory of {label}-ori.npy clean clips")
    ap.add_argument("--num_classes", type=int, default=101 if ucf101 else None,
                    help="classes of the built-in factories' heads (default: 400, the UCF-101 twin 101)")
    args = ap.parse_args(argv)
    if ucf101 and os.environ.get("I2V_UCF_CKPT_PATH"):
        os.environ["I2V_WEIGHTS_DIR"] = os.environ["I2V_UCF_CKPT_PATH"]
    adv_path = os.path.join(os.environ.get("I2V_OPT_PATH", ""), args.adv_path)
    device = torch.device(f"cuda:{args.gpu.split(',')[0]}" if torch.cuda.is_available() else "cpu")
    if args.model_factory == "proxy":
        factory = proxy
    elif args.model_factory == "native":
        factory = native
    else:
        mod, fn = args.model_factory.split(":")
        factory = getattr(importlib.import_module(mod), fn)
    files = sorted(f for f in os.listdir(adv_path) if "adv" in f and f.endswith(".npy"))
    nb = math.ceil(len(files) / args.batch_size)
    files_batch = [files[i * args.batch_size:(i + 1) * args.batch_size] for i in range(nb)]
    model_val_acc, columns = {}, {}
    for name in [m for m in args.models.split(",") if m]:
        own = factory in (proxy, native) and args.num_classes is not None
        model = (factory(name, num_classes=args.num_classes) if own else factory(name)).to(device).eval()
        preds, labels, top1 = evaluate(model, adv_path, files_batch, device, args.clean_dir or None)
        predd = np.zeros_like(preds)
        for i, ind in enumerate(np.argsort(labels)):          # reference.py:116-119
            predd[ind] = preds[i]
        columns["gt_label"] = sorted(labels)
        columns[f"{name}-pre"] = predd
        model_val_acc[name] = top1
        print("Model-{}: top-1 {:.2f}%  fooling rate {:.2f}%".format(name, top1, 100 - top1))
    keys = list(columns)
    with open(os.path.join(adv_path, "results_all_models_prediction.csv"), "w") as fh:
        fh.write(",".join(keys) + "\n")
        for r in range(len(files)):
            fh.write(",".join(str(int(columns[k][r])) for k in keys) + "\n")
    with open(os.path.join(adv_path, "top1_acc_all_models.json"), "w") as opt:
        json.dump(model_val_acc, opt)
    return model_val_acc


if __name__ == "__main__":
    main()
