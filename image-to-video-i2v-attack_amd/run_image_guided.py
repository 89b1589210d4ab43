#!/usr/bin/env python
"""Drop-in for `/root/reference/run_image_guided.py`: the paper's experiment sweeps as child
processes of `image_main.py` (same flags `--gpu --batch_size`).  The evaluation half of every pair
(`reference.py ...`) needs the gluoncv video models and is only run when
`$I2V_EVAL_CMD` names an evaluator taking `--gpu G --adv_path P`.  `--dry_run` prints the commands."""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MAIN = [sys.executable, os.path.join(HERE, "image_main.py")]


def i2v(gpu, step, lr, model, depth, prefix, bs=1):
    return MAIN + ["--gpu", gpu, "--attack_method", "ImageGuidedFMDirection_Adam", "--step", str(step), "--step_size",
                   str(lr), "--direction_image_model", model, "--depth", str(depth), "--batch_size", str(bs),
                   "--batch_nums", "1", "--batch_index", "1", "--file_prefix", prefix]


def plan(gpu, bs):
    jobs = []
    for step in (20, 40, 60, 80, 100):                                  # Figure 4 (run_image_guided.py:46-52)
        for lr in (0.001, 0.0025, 0.0050, 0.0075, 0.010):
            jobs.append((i2v(gpu, step, lr, "resnet", 1, f"resnet_step_size_{lr}_paper_study", bs),
                         f"Image-ImageGuidedFMDirection_Adam-{step}-resnet_step_size_{lr}_paper_study"))
    for model in ("resnet", "squeezenet", "vgg", "alexnet"):            # Table 2 / Figure 5 (:55-60)
        for depth in (1, 2, 3, 4):
            jobs.append((i2v(gpu, 60, 0.005, model, depth, f"{model}-step_size-0.005-depth-{depth}_paper_study"),
                         f"Image-ImageGuidedFMDirection_Adam-60-{model}-step_size-0.005-depth-{depth}_paper_study"))
    for model in ("squeezenet", "vgg", "alexnet", "resnet"):            # Table 3 (:63-80)
        depth = 2 if model in ("resnet", "squeezenet") else 3
        jobs.append((i2v(gpu, 60, 0.005, model, depth, f"{model}-depth-{depth}_paper_per_com"),
                     f"Image-ImageGuidedFMDirection_Adam-60-{model}-depth-{depth}_paper_per_com"))
        std = i2v(gpu, 60, 0.005, model, depth, f"{model}-depth-{depth}_paper_per_com")
        std[std.index("ImageGuidedFMDirection_Adam")] = "ImageGuidedStd_Adam"
        jobs.append((std, f"Image-ImageGuidedStd_Adam-60-{model}-depth-{depth}_paper_per_com"))
    jobs.append((MAIN + ["--gpu", gpu, "--attack_method", "ImageGuidedFML2_Adam_MultiModels", "--step", "60",
                         "--step_size", "0.005", "--file_prefix", "paper_per_com"],
                 "Image-ImageGuidedFML2_Adam_MultiModels-60-paper_per_com"))
    return jobs


def main(argv=None):
    ap = argparse.ArgumentParser(description="")
    ap.add_argument("--gpu", type=str, default="0", help="gpu device.")
    ap.add_argument("--batch_size", type=int, default=1, help="")
    ap.add_argument("--dry_run", action="store_true")
    args, extra = ap.parse_known_args(argv)
    evaluator = os.environ.get("I2V_EVAL_CMD", "")
    for cmd, adv_path in plan(args.gpu, args.batch_size):
        cmd = cmd + extra
        print(" ".join(cmd))
        if not args.dry_run:
            subprocess.run(cmd, check=False)
            if evaluator:
                subprocess.run(evaluator.split() + ["--gpu", args.gpu, "--adv_path", adv_path], check=False)


if __name__ == "__main__":
    main()
