#!/usr/bin/env python
"""Drop-in for `/root/reference/run_image_guided.py`: the paper's experiment sweeps as child processes, attack then
evaluator for every cell, in the reference's order and with the reference's flags (same flags `--gpu --batch_size`):

  * Figure 4  (`:44-52`)  step count x step size, ResNet, `image_main.py` + `reference.py`
  * Table 2 / Figure 5 (`:54-60`)  4 image models x depths 1-4
  * Table 3   (`:62-80`)  Kinetics-400: I2V and Std per model (resnet / squeezenet depth 2, vgg / alexnet depth 3), then ENS-I2V
  * Table 4   (`:82-100`) UCF-101: the same three over `image_main_ucf101.py` + `reference_ucf101.py`

`plan()` returns the (attack argv, evaluator argv) pairs; `tests/test_cli_and_dist_cpu.py` formats the reference's own
command templates (`:5-29`) and compares them with this list flag for flag.  The evaluator leg is this directory's
`reference.py` / `reference_ucf101.py` as in the reference (`:51-52`); `$I2V_EVAL_CMD` names another program taking
`--gpu G --adv_path P`, `$I2V_EVAL_ARGS` appends flags (e.g. `--model_factory native`), `--no_eval` skips the leg.
`--dry_run` prints the commands; flags this script does not know are handed on to the attack CLIs (`--clip_dir ...`)."""
import argparse
import os
import shlex
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _script(name):
    return [sys.executable, os.path.join(HERE, name)]


def _attack(script, gpu, method, flags, prefix):
    argv = _script(script) + ["--gpu", gpu, "--attack_method", method]
    for k, v in flags:
        argv += ["--" + k, str(v)]
    return argv + ["--file_prefix", prefix]


def _cell(script, evaluator, gpu, method, step, flags, prefix):
    """One (attack, evaluation) pair; the adversarial directory is the attack CLI's own `Image-{method}-{step}-{prefix}`."""
    return (_attack(script, gpu, method, flags, prefix),
            _script(evaluator) + ["--gpu", gpu, "--adv_path", f"Image-{method}-{step}-{prefix}"])


def _per_com(script, evaluator, gpu, models):
    """Tables 3 and 4: I2V then Std per image model, then the ensemble (which the reference runs on the CLI's own --step / --step_size
    handling: image_main.py:80 ignores them, image_main_ucf101.py:75 passes --step on)."""
    jobs = []
    for model in models:
        depth = 2 if model in ("resnet", "squeezenet") else 3
        for method in ("ImageGuidedFMDirection_Adam", "ImageGuidedStd_Adam"):
            jobs.append(_cell(script, evaluator, gpu, method, 60,
                              [("step", 60), ("step_size", 0.005), ("direction_image_model", model), ("depth", depth)],
                              f"{model}-depth-{depth}_paper_per_com"))
    jobs.append(_cell(script, evaluator, gpu, "ImageGuidedFML2_Adam_MultiModels", 60, [("step", 60), ("step_size", 0.005)],
                      "paper_per_com"))
    return jobs


def plan(gpu, bs=1):
    """`bs` (`--batch_size`) is parsed and then not used, exactly as in the reference (`:50` formats `batch_size=1` whatever the flag says;
    the other templates carry no --batch_size at all).  `--loader_batch N` (an addition) appends `--batch_size N` to every attack
    command -- argparse keeps the last occurrence."""
    fmd = "ImageGuidedFMDirection_Adam"
    jobs = []
    for step in (20, 40, 60, 80, 100):                                  # Figure 4 (run_image_guided.py:46-52)
        for lr in (0.001, 0.0025, 0.0050, 0.0075, 0.010):
            jobs.append(_cell("image_main.py", "reference.py", gpu, fmd, step,
                              [("step", step), ("step_size", lr), ("direction_image_model", "resnet"), ("batch_size", 1),
                               ("batch_nums", 1), ("batch_index", 1)], f"resnet_step_size_{lr}_paper_study"))
    for model in ("resnet", "squeezenet", "vgg", "alexnet"):            # Table 2 / Figure 5 (:55-60)
        for depth in (1, 2, 3, 4):
            jobs.append(_cell("image_main.py", "reference.py", gpu, fmd, 60,
                              [("step", 60), ("step_size", 0.005), ("direction_image_model", model), ("depth", depth)],
                              f"{model}-step_size-0.005-depth-{depth}_paper_study"))
    jobs += _per_com("image_main.py", "reference.py", gpu, ("squeezenet", "vgg", "alexnet", "resnet"))              # Table 3 (:63-80)
    jobs += _per_com("image_main_ucf101.py", "reference_ucf101.py", gpu, ("resnet", "squeezenet", "vgg", "alexnet"))  # Table 4 (:83-100)
    return jobs


def main(argv=None):
    ap = argparse.ArgumentParser(description="")
    ap.add_argument("--gpu", type=str, default="0", help="gpu device.")
    ap.add_argument("--batch_size", type=int, default=1, help="")
    ap.add_argument("--dry_run", action="store_true")
    ap.add_argument("--no_eval", action="store_true", help="attacks only")
    ap.add_argument("--loader_batch", type=int, default=0, help="clips per loader batch of every attack CLI (0: the reference's commands as they are)")
    args, extra = ap.parse_known_args(argv)
    if args.loader_batch:
        extra = ["--batch_size", str(args.loader_batch)] + extra
    evaluator = shlex.split(os.environ.get("I2V_EVAL_CMD", ""))
    eval_args = shlex.split(os.environ.get("I2V_EVAL_ARGS", ""))
    for attack, evaluate in plan(args.gpu, args.batch_size):
        attack = attack + extra
        if evaluator:
            evaluate = evaluator + evaluate[2:]
        evaluate = evaluate + eval_args
        for cmd in (attack,) if args.no_eval else (attack, evaluate):
            print(" ".join(cmd), flush=True)
            if not args.dry_run:
                subprocess.run(cmd, check=False)          # os.system in the reference: a failing cell does not stop the sweep


if __name__ == "__main__":
    main()
