#!/usr/bin/env python
"""Drop-in for `/root/reference/attack.py` restricted to `--attack_type image`: white-box FGSM / BIM / MIFGSM
(`base_attacks.py:236-340`) on a video classifier, writing the pairs ILAF fine-tunes --
`{model}-{method}-{step}-{prefix}/{label}-adv.npy` and `{label}-ori.npy` (`:102-108`).

Kept: the flags that drive this path and their defaults (`--gpu --batch_size --model --attack_method --attack_type
--step --file_prefix`, `:15-34`; the other flags of the reference parameterise `video_attacks.py`, which is outside
the hot path, and are accepted and ignored), the per-batch call `attack_method(val_batch, val_label)` and both
artefacts per clip.  The update rule runs in `libi2v_hip.so` (`i2v_sign_step_f32`); the classifier and its
cross-entropy gradient are PyTorch, exactly as in the reference -- gluoncv's Kinetics-400 models are not vendored, so
the classifier comes from a factory like the evaluator's (`reference.py`): `--model_factory pkg.module:function`,
a callable `name -> torch.nn.Module` on normalised (b,3,f,h,w) clips; default `reference:proxy`.
`OPT_PATH` is `$I2V_OPT_PATH`; clips come from `i2v_amd.clips` (synthetic or `--clip_dir`)."""
import argparse
import importlib
import os

import numpy as np
import torch

import base_attacks
import video_attacks
from i2v_amd import clips

OPT_PATH = os.environ.get("I2V_OPT_PATH", "")
# flags the reference parses but no code path reads (attack.py:24-51 there)
IGNORED = ["--sf_frame", "--cf_frame", "--nsig", "--gamma", "--momentum_weight"]
IGNORED_FLAGS = ["--frame_conv", "--frame_momentum", "--no_iterative_momentum", "--weight_add",
                 "--iterative_first", "--translation_invariant", "--temporal_augmentation", "--TI_First", "--noise",
                 "--shuffle_grads"]


def arg_parse(argv=None):
    parser = argparse.ArgumentParser(description="")
    parser.add_argument("--gpu", type=str, default="0", help="gpu device.")
    parser.add_argument("--batch_size", type=int, default=4, metavar="N")
    parser.add_argument("--model", type=str, default="i3d_resnet101")
    parser.add_argument("--attack_method", type=str, default="TemporalAugmentationMomentum", help="FGSM | BIM | MIFGSM")
    parser.add_argument("--attack_type", type=str, default="image", help="image | video")
    parser.add_argument("--step", type=int, default=10, metavar="N")
    parser.add_argument("--file_prefix", type=str, default="")
    # TemporalTranslation (attack.py:28,32-33,38,51,79 of the reference)
    parser.add_argument("--kernlen", type=int, default=15, metavar="N")
    parser.add_argument("--kernel_mode", type=str, default="gaussian")
    parser.add_argument("--iterative_momentum", action="store_true", default=False)
    parser.add_argument("--augmentation_weight", type=float, default=1.0)
    parser.add_argument("--move_type", type=str, default="adj", help="adj | large | random")
    for f in IGNORED:
        parser.add_argument(f, default=None, help="parsed by the reference, read by nothing")
    for f in IGNORED_FLAGS:
        parser.add_argument(f, action="store_true", default=False, help="parsed by the reference, read by nothing")
    # additions (not in the reference)
    parser.add_argument("--model_factory", type=str, default="native",
                        help="'native' (default): the I3D / SlowFast graphs with a native classifier head, every launch behind the C ABI; or "
                             "pkg.module:function (name -> torch classifier; e.g. reference:proxy): the caller's torch module, its gradient from "
                             "PyTorch autograd, only the update rule native -- the attack object prints which path it runs on")
    parser.add_argument("--num_classes", type=int, default=400)
    parser.add_argument("--anno", type=str, default=os.environ.get("I2V_ANNO", ""))
    parser.add_argument("--clip_dir", type=str, default="")
    parser.add_argument("--num_clips", type=int, default=400)
    parser.add_argument("--frames", type=int, default=32)
    parser.add_argument("--hw", type=int, default=224)
    args = parser.parse_args(argv)
    args.adv_path = os.path.join(OPT_PATH, "{}-{}-{}-{}".format(args.model, args.attack_method, args.step, args.file_prefix))
    os.makedirs(args.adv_path, exist_ok=True)
    return args


def main(argv=None):
    args = arg_parse(argv)
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = args.gpu.split(",")[0]
    print(args)
    if args.attack_type not in ("image", "video"):
        raise UnboundLocalError("local variable 'attack_method' referenced before assignment")      # as the reference ends up (:75-82)
    dev = torch.device(f"cuda:{os.environ['LOCAL_RANK']}" if torch.cuda.is_available() else "cpu")
    if args.model_factory == "native":        # graph IR + classifier head: the whole white-box gradient behind the C ABI
        from i2v_amd.video import VideoModel
        model = VideoModel(args.model, (args.frames, args.hw, args.hw), num_classes=args.num_classes)
    else:
        mod, fn = args.model_factory.split(":")
        model = getattr(importlib.import_module(mod), fn)(args.model).to(dev)
    if args.attack_type == "image":
        attack_method = getattr(base_attacks, args.attack_method)(model, steps=args.step)   # AttributeError for the default name, as in the reference (:22)
    else:
        if args.attack_method == "TemporalTranslation":                                         # :78-79
            spe_params = {"kernlen": args.kernlen, "momentum": args.iterative_momentum, "weight": args.augmentation_weight,
                          "move_type": args.move_type, "kernel_mode": args.kernel_mode}
        print("Used Params")
        print(spe_params)                                                                       # UnboundLocalError for any other name, as there
        attack_method = getattr(video_attacks, args.attack_method)(model, params=spe_params, steps=args.step)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    total = clips.num_batches(args.batch_size, args.anno, args.clip_dir, args.num_clips)
    for step, (val_batch, val_label, _) in enumerate(clips.batches(args.batch_size, args.anno, args.clip_dir, args.frames,
                                                                    args.hw, args.num_clips)):
        if step % world != rank:
            continue
        print("Running {}, {}/{}".format(args.attack_method, step + 1, total))
        val_batch, val_label = val_batch.to(dev), val_label.to(dev)
        adv_batches = attack_method(val_batch, val_label % args.num_classes)
        val_batch = val_batch.detach()
        for ind, label in enumerate(val_label):
            np.save(os.path.join(args.adv_path, "{}-adv".format(label.item())), adv_batches[ind].cpu().numpy())
            np.save(os.path.join(args.adv_path, "{}-ori".format(label.item())), val_batch[ind].cpu().numpy())
    return args.adv_path


if __name__ == "__main__":
    main()
