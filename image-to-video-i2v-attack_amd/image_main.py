#!/usr/bin/env python
"""Drop-in for `/root/reference/image_main.py`: same flags and defaults (`:15-48`), same shard
window (`:61-63`), same artefacts -- `OPT_PATH/Image-{method}-{step}-{prefix}/{label}-adv.npy`
(float32 (3,32,224,224), normalised) and `loss_info_{batch_index}.json` (`:45,90-95`).

Differences: `OPT_PATH` comes from `$I2V_OPT_PATH` (the reference hard-codes an empty constant,
`utils.py:21`); clips come from `i2v_amd.clips` (synthetic or `--clip_dir`), not decord; under
`torchrun` the shard defaults to (WORLD_SIZE, RANK+1) so 8 GPUs need no manual `--batch_index`;
`--resume` skips clips whose `{label}-adv.npy` already exists."""
import argparse
import json
import os
import queue
import threading

import numpy as np
import torch

import image_attacks
import TPAMI_attack
from i2v_amd import clips

OPT_PATH = os.environ.get("I2V_OPT_PATH", "")


def arg_parse(argv=None, ucf101=False):
    parser = argparse.ArgumentParser(description="")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    parser.add_argument("--batch_nums", type=int, default=world)
    parser.add_argument("--batch_index", type=int, default=rank + 1)
    parser.add_argument("--gpu", type=str, default="0", help="gpu device.")
    parser.add_argument("--batch_size", type=int, default=1, metavar="N")
    parser.add_argument("--attack_method", type=str, default="ImageGuidedAttentionMap", help="")
    parser.add_argument("--step", type=int, default=10 if ucf101 else 60, metavar="N")      # image_main_ucf101.py:26
    parser.add_argument("--file_prefix", type=str, default="")
    parser.add_argument("--depth", type=int, default=1, help="1,2,3,4")
    parser.add_argument("--lamb", type=float, default=0.1, help="")
    parser.add_argument("--mode", type=str, default="direction", help="diff_norm\\direction")
    parser.add_argument("--step_size", type=float, default=0.004, help="")
    parser.add_argument("--dropout", type=float, default=0.1, help="")
    parser.add_argument("--direction_image_model", type=str, default="resnet",
                        help="resnet, densenet, squeezenet, vgg, alexnet (resnet50 added)")
    # additions (not in the reference)
    parser.add_argument("--anno", type=str, default=os.environ.get("I2V_ANNO", ""),
                        help="sample list csv `path,gt_label,clip_index` (the reference's kinetics400_attack_samples.csv, utils.py:29); without it 400 synthetic names with labels 0..399 are used")
    parser.add_argument("--clip_dir", type=str, default="", help="directory of {label}-ori.npy clips")
    parser.add_argument("--num_clips", type=int, default=400, help="use only the first N rows of the sample list")
    parser.add_argument("--frames", type=int, default=32)
    parser.add_argument("--hw", type=int, default=224)
    parser.add_argument("--resume", action="store_true")
    if ucf101:           # the UCF-101 loader's inputs (dataset_ucf101.py:52-64, utils.UCF_IMAGE_ROOT): jpg frame folders + clip list
        parser.add_argument("--frame_dir", type=str, default=os.environ.get("I2V_UCF_IMAGE_ROOT", ""),
                            help="UCF_IMAGE_ROOT: <frame_dir>/<video_path>/image_00001.jpg ...; decoded by Pillow on the --workers threads")
        parser.add_argument("--setting", type=str, default="./test01_setting.txt", help="lines `video_path duration label`")
        parser.add_argument("--used_idxs", type=str, default="./used_idxs.pkl",
                            help="rows of --setting to attack (the reference's pickled list of ints, or .json / text); '' = all rows")
    parser.add_argument("--pin_cpus", action="store_true",
                        help="hand-started shards only (no launcher): confine this process to slot batch_index-1 of batch_nums "
                             "of the host's cores -- for a node running ALL batch_nums shards at once (run_image_guided.py)")
    parser.add_argument("--workers", type=int, default=2, help="loader threads (0: load in the reader thread itself)")
    parser.add_argument("--group_clips", type=int, default=4,
                        help="attack up to this many clips of READY loader batches in one engine call (I2V / ENS-I2V only; same clips, "
                             "same logged costs as batch-by-batch; 1 = one call per loader batch as in the reference)")
    parser.add_argument("--synthetic_weights", action="store_true",
                        help="run on the seeded synthetic initialiser when no checkpoint lies under $I2V_WEIGHTS_DIR "
                             "(same as I2V_SYNTHETIC_WEIGHTS=1); without it a missing checkpoint is an error")
    args = parser.parse_args(argv)
    if args.synthetic_weights:
        os.environ["I2V_SYNTHETIC_WEIGHTS"] = "1"
    args.adv_path = os.path.join(OPT_PATH, "{}-{}-{}-{}".format("Image", args.attack_method, args.step, args.file_prefix))
    os.makedirs(args.adv_path, exist_ok=True)
    return args


def build_attack(args, ucf101=False):
    if ucf101 and args.attack_method == "ImageGuidedFML2_Adam_MultiModels":       # the UCF-101 twin passes --step on (image_main_ucf101.py:75)
        names = ["resnet", "vgg", "squeezenet", "alexnet"]
        return image_attacks.ImageGuidedFML2_Adam_MultiModels(names, depths={"resnet": 2, "vgg": 3, "squeezenet": 2, "alexnet": 3}, steps=args.step)
    if ucf101 and args.attack_method == "AENS_I2V_MF":                            # ... and has no adaptive branch (:62-75)
        return getattr(image_attacks, args.attack_method)
    if args.attack_method in ("ImageGuidedStd_Adam", "ImageGuidedFMDirection_Adam"):
        return getattr(image_attacks, args.attack_method)([args.direction_image_model], depth=args.depth,
                                                          step_size=args.step_size, steps=args.step)
    if args.attack_method == "ImageGuidedFML2_Adam_MultiModels":
        # the reference ignores --step/--step_size here (image_main.py:80): 60 steps, lr 0.005
        names = ["resnet", "vgg", "squeezenet", "alexnet"]
        return image_attacks.ImageGuidedFML2_Adam_MultiModels(names, depths={"resnet": 2, "vgg": 3, "squeezenet": 2, "alexnet": 3})
    if args.attack_method == "AENS_I2V_MF":
        names = ["resnet", "vgg", "squeezenet", "alexnet"]
        return TPAMI_attack.AENS_I2V_MF(names, depths={n: [2, 3] for n in names}, step_size=args.step_size, steps=args.step)
    return getattr(image_attacks, args.attack_method)          # AttributeError, as in the reference


def main(argv=None, ucf101=False):
    """`ucf101`: the behaviour of the reference's `image_main_ucf101.py` (drop-in module of that name): --step defaults to 10, the
    ensemble class receives it, names are `str(val_label)` (:83), and decoded uint8 clips go through the UCF-101 loader's transform
    (PIL `Scale(224)` + `CornerCrop`, `Engine.clip_resample_crop`) instead of the Kinetics one."""
    args = arg_parse(argv, ucf101)
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = args.gpu.split(",")[0]
    # This shard's CPU cores, before anything touches the GPU; reader / writer / lane threads inherit them.  Only where the
    # placement is KNOWN: under a launcher (LOCAL_RANK / LOCAL_WORLD_SIZE are the ranks of this node), or by hand with
    # `--pin_cpus` (`--batch_nums 8 --batch_index k --gpu k-1`, run_image_guided.py: the node's shards, slot = batch_index - 1).
    # A lone hand-started shard (`--batch_nums 8 --batch_index 3 --gpu 0`) is NOT confined to an eighth of the host.
    from i2v_amd import affinity
    pinned = None
    if "LOCAL_WORLD_SIZE" in os.environ and os.environ["LOCAL_RANK"].isdigit():
        lr, lw = int(os.environ["LOCAL_RANK"]), int(os.environ["LOCAL_WORLD_SIZE"])
        pinned = affinity.pin_rank(lr, lw) if lr < lw else None
    elif args.pin_cpus:
        pinned = affinity.pin_rank(args.batch_index - 1, args.batch_nums) if 1 <= args.batch_index <= args.batch_nums else None
    if pinned:
        print("cpu affinity:", pinned)
    print(args)
    frame_dir = getattr(args, "frame_dir", "")
    if frame_dir:        # UCF-101 jpg frame folders: clip list + LoopPadding(32) + Pillow decoding on the host, transform on the device
        used = args.used_idxs or None
        total = clips.ucf101_num_batches(args.batch_size, args.setting, frame_dir, used)

        def source():
            return clips.ucf101_batches(args.batch_size, args.setting, frame_dir, used, args.frames, workers=args.workers)
    else:
        total = clips.num_batches(args.batch_size, args.anno, args.clip_dir, args.num_clips)

        def source():
            return clips.batches(args.batch_size, args.anno, args.clip_dir, args.frames, args.hw, args.num_clips, workers=args.workers)
    nums_contained = int(total / args.batch_nums)                      # int(400 / batch_nums), :61
    left = (args.batch_index - 1) * nums_contained
    right = args.batch_index * nums_contained
    attack_method = build_attack(args, ucf101)
    cuda = torch.cuda.is_available()
    if cuda:
        from i2v_amd import attacks as _attacks
        device = torch.device(_attacks.default_device())
        torch.cuda.set_device(device)          # events, pinned copies and the attack's stream all live on THIS rank's device

    # I/O off the critical path: clips are produced by a reader thread one batch ahead (synthetic generation
    # or np.load), results are copied to pinned host memory asynchronously and written by a writer thread,
    # so the GPU goes straight from one batch to the next (the reference loads, attacks and np.saves serially,
    # image_main.py:82-92).
    group = max(1, args.group_clips)
    if group > args.batch_size and hasattr(attack_method, "reserve") and getattr(attack_method, "_mode", "") == "i2v":
        attack_method.reserve(max(group, args.batch_size), args.frames, (args.hw, args.hw))     # plan once, for the largest group
    todo = queue.Queue(maxsize=max(2, group))
    done = queue.Queue(maxsize=4)

    reader_error = []

    def reader():
        try:
            if cuda:
                torch.cuda.set_device(device)      # per-thread state: pin_memory() would otherwise create a context on device 0
            for step, item in enumerate(source()):
                if not (left <= step < right):
                    continue
                if args.resume and all(os.path.exists(os.path.join(args.adv_path, f"{l.item()}-adv.npy")) for l in item[1]):
                    continue
                batch = item[0].pin_memory() if cuda else item[0]
                todo.put((step, batch, item[1], item[2]))
        except BaseException as e:             # a clip that cannot be read must end the run, not leave the main loop waiting
            reader_error.append(e)
        finally:
            todo.put(None)

    def writer():
        failed = False
        while True:
            item = done.get()
            if item is None:
                return
            if failed:
                continue                     # keep draining so that the main loop's done.put() never blocks
            try:
                labels, host, event = item
                if event is not None:
                    event.synchronize()
                for ind, label in enumerate(labels):
                    np.save(os.path.join(args.adv_path, "{}-adv".format(label.item())), host[ind].numpy())
            except BaseException as e:       # disk full, unwritable directory: end the run with that error
                writer_error.append(e)
                failed = True

    threads = [threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)]
    for t in threads:
        t.start()
    ended = False
    carry = None                            # a loader batch taken from the queue that would have overshot the group
    writer_error = []
    while not ended:
        if writer_error:
            break
        # Loader batches that are READY are attacked in one engine call, up to --group_clips clips (frames are independent in
        # I2V / ENS-I2V: every clip and every logged cost is what the batch's own call produces, `forward_grouped`); the
        # reference's default `--batch_size 1` alone would leave the GPU a third empty.  Nothing waits for a batch that is
        # not there yet.
        items, nclips = [], 0
        while nclips < group:
            if carry is not None:
                item, carry = carry, None
            else:
                try:
                    item = todo.get() if not items else todo.get_nowait()
                except queue.Empty:
                    break
            if item is None:
                ended = True
                break
            if items and nclips + int(item[1].shape[0]) > max(group, args.batch_size):
                carry = item                # the plan was reserved for `group` clips: never collect more (a larger group would re-plan)
                break
            items.append(item)
            nclips += int(item[1].shape[0])
        if not items:
            break
        batches = []
        for step, val_batch, val_label, video_names in items:
            print("Running {}, {}/{}".format(args.attack_method, step + 1, total))
            if val_batch.dtype == torch.uint8:          # decoded frames (b,t,H,W,3): the loader's Resize/CenterCrop/ToTensor/Normalize on the device
                from i2v_amd import attacks as _attacks
                eng = _attacks.get_engine()
                raw = val_batch.to(eng.device, non_blocking=True).contiguous()
                val_batch = eng.clip_resample_crop(raw, args.hw, args.hw) if ucf101 else eng.clip_resize_crop(raw, crop=args.hw)
            if ucf101:
                video_names = str(val_label)            # image_main_ucf101.py:83 -- the attack then iterates the CHARACTERS of this string
            batches.append((val_batch, val_label, video_names))
        if len(batches) > 1 and hasattr(attack_method, "forward_grouped"):
            outs = attack_method.forward_grouped(batches)
        else:
            outs = [attack_method(*bt) for bt in batches]
        for (_, val_label, _), adv_batches in zip(batches, outs):
            if isinstance(adv_batches, tuple):                             # AENS returns (adv, time, costs)
                adv_batches = adv_batches[0]
            adv_batches = adv_batches.detach()
            if adv_batches.is_cuda:
                host = torch.empty(adv_batches.shape, dtype=adv_batches.dtype, pin_memory=True)
                host.copy_(adv_batches, non_blocking=True)
                event = torch.cuda.Event()
                event.record(torch.cuda.current_stream(adv_batches.device))
            else:
                host, event = adv_batches.contiguous(), None
            done.put((val_label, host, event))
    done.put(None)
    threads[1].join()
    if writer_error:
        raise writer_error[0]
    if reader_error:
        raise reader_error[0]
    with open(os.path.join(args.adv_path, "loss_info_{}.json".format(args.batch_index)), "w") as opt:
        json.dump(attack_method.loss_info, opt)
    threads[0].join(timeout=60)
    if cuda and __name__ == "__main__":
        del attack_method
        from i2v_amd import attacks as _a
        _a.shutdown()


if __name__ == "__main__":
    main()
