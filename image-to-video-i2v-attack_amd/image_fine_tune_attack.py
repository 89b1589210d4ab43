#!/usr/bin/env python
"""Drop-in for `/root/reference/image_fine_tune_attack.py`: ILAF fine-tuning of pre-generated adversarial clips
against a white-box video model.  Same flags and defaults (`:40-54`), same inputs -- `--used_adv/{id}-adv.npy` with
its clean partner `--used_ori/{id}-ori.npy` (`:16-37`) -- same artefact `--opt_path/{label}-adv.npy` (`:80-82`),
one clip per call (`:73-79`).

Differences: the white-box model is an `i2v_amd.video.VideoModel` (graph IR + weights; the reference builds a gluoncv
module from a YACS config, `:58-66`), so `--white_model` is one of the names `graphs.build_video` knows and the whole
ILAF loop runs in `libi2v_hip.so`; under `torchrun` the file list is dealt round-robin over the ranks (replicas only,
no collective); `--steps` / `--step_size` expose ILAF's constructor defaults (60, 0.005; `image_attacks.py:502`);
`--resume` skips clips whose output exists; `--group_clips K` (default 8) hands K clips to ONE engine call as K independent
one-clip problems (`ILAF.forward_independent`: per-clip loss segments, every `{label}-adv.npy` bit-identical to the
one-clip call -- the reference's one clip per call cannot fill the GPU); `--streams N` (default 3) keeps N such calls in
flight on separate HIP streams (`sign_attacks.run_concurrent`)."""
import argparse
import os

import numpy as np
import torch

import image_attacks
from i2v_amd.sign_attacks import run_concurrent
from i2v_amd.video import VideoModel


class AdvDataset(object):
    """`image_fine_tune_attack.py:16-37`."""

    def __init__(self, used_adv_path, used_ori_path):
        self.used_adv_path, self.used_ori_path = used_adv_path, used_ori_path
        self.files = sorted(f for f in os.listdir(used_adv_path) if "adv" in f)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        file = self.files[idx]
        vid_id = file.split("-")[0]
        vid = torch.from_numpy(np.load(os.path.join(self.used_adv_path, file)))[None]
        ori_vid = torch.from_numpy(np.load(os.path.join(self.used_ori_path, "{}-ori.npy".format(vid_id))))[None]
        label = torch.from_numpy(np.array([int(vid_id)]).astype(np.int32)).long()
        return vid, ori_vid, label


def arg_parse(argv=None):
    parser = argparse.ArgumentParser(description="")
    parser.add_argument("--gpu", type=str, default="0", help="gpu device.")
    parser.add_argument("--batch_size", type=int, default=4, metavar="N")
    parser.add_argument("--attack_method", type=str, default="ILAF", help="")
    parser.add_argument("--opt_path", type=str, default="")
    parser.add_argument("--used_adv", type=str, default="", help="")
    parser.add_argument("--used_ori", type=str, default="", help="")
    parser.add_argument("--white_model", type=str, default="i3d_resnet101",
                        help="i3d_resnet50 | i3d_resnet101 | slowfast_resnet50 | slowfast_resnet101 | tpn_resnet50 | tpn_resnet101")
    parser.add_argument("--dataset", type=str, default="Kinetics-400", help="Kinetics-400 | UCF-101")
    # additions (not in the reference)
    parser.add_argument("--steps", type=int, default=60)
    parser.add_argument("--step_size", type=float, default=0.005)
    parser.add_argument("--resume", action="store_true")
    parser.add_argument("--synthetic_weights", action="store_true",
                        help="run on the seeded synthetic initialiser when no checkpoint lies under $I2V_WEIGHTS_DIR "
                             "(same as I2V_SYNTHETIC_WEIGHTS=1); without it a missing checkpoint is an error")
    parser.add_argument("--streams", type=int, default=3, help="engine calls in flight on separate HIP streams")
    parser.add_argument("--group_clips", type=int, default=8,
                        help="clips per engine call, attacked as independent one-clip problems (1 = the reference's call pattern)")
    args = parser.parse_args(argv)
    if args.synthetic_weights:
        os.environ["I2V_SYNTHETIC_WEIGHTS"] = "1"
    return args


def main(argv=None, model_kwargs=None):
    args = arg_parse(argv)
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = args.gpu.split(",")[0]
    from i2v_amd import affinity
    affinity.pin_rank()              # this rank's CPU cores (under a launcher), before anything touches the GPU
    print(args)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dataset = AdvDataset(args.used_adv, args.used_ori)
    os.makedirs(args.opt_path, exist_ok=True)
    todo = []
    for step in range(rank, len(dataset), world):
        vid_id = dataset.files[step].split("-")[0]
        if args.resume and os.path.exists(os.path.join(args.opt_path, "{}-adv.npy".format(int(vid_id)))):
            continue
        todo.append(step)
    if not todo:
        return None
    shape = tuple(dataset[todo[0]][0].shape[2:])            # the clip shape decides the plan

    def make_attack():
        model = VideoModel(args.white_model, shape, **(model_kwargs or {}))
        atk = getattr(image_attacks, args.attack_method)(model, args.white_model, step_size=args.step_size, steps=args.steps)
        atk.independent_clips = True          # a grouped call = its clips' one-clip calls (image_fine_tune_attack.py:73-79)
        return atk

    group = max(1, args.group_clips)

    def items():
        for g0 in range(0, len(todo), group):
            got = []
            for step in todo[g0:g0 + group]:
                print("Running {}, {}/{}".format(args.attack_method, step + 1, len(dataset)))
                vid, ori_vid, label = dataset[step]
                if torch.equal(vid, ori_vid):        # no perturbation to fine-tune (ILAF's direction would be 0/0): skip, loudly
                    print("Skipping {}: its adversarial clip equals its original".format(dataset.files[step]))
                    continue
                got.append((vid, ori_vid, label))
            if not got:
                continue
            yield (torch.cat([g[0] for g in got]), torch.cat([g[1] for g in got]), torch.cat([g[2] for g in got]), ["..."] * len(got))
    def save(_i, item, adv_batches):
        for ind, label in enumerate(item[2]):
            np.save(os.path.join(args.opt_path, "{}-adv".format(label.item())), adv_batches[ind].detach().cpu().numpy())
    _, attacks_ = run_concurrent(make_attack, items(), args.streams, on_result=save)
    return attacks_[0]


if __name__ == "__main__":
    main()
