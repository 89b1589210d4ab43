"""TEST INFRASTRUCTURE -- CPU worker of the fooling-rate parity measurement (`tools/fooling_parity.py`,
`tests/test_gpu_size_parity.py`): the fp32 oracle's WHOLE I2V attack (`oracle/size_parity.oracle_attack`: ResNet-50 layer3,
32 x 224^2, 10 Adam steps, `/root/reference/image_attacks.py:294-364`) on the clips keyed to rows of
`kinetics400_attack_samples.csv` -- row r is the synthetic clip of seed 1000 + r, attacked under the name / label of that row.

    python -m oracle.fooling_worker --rows 3,11,19 --threads 32 --slot 3 --out DIR

A worker is a child PROCESS of its caller (started with subprocess, never an exec of the caller itself), touches no GPU, and
confines itself to `threads` of the host's CPUs (slot k takes the k-th group of that many allowed CPUs).  Per row it writes
`DIR/{row}-oracle-adv.npy` (float32 (3,32,224,224), the file the evaluator scores) and then `DIR/{row}-oracle.npz` (costs,
mean|delta_10|, seconds) -- the npz appears last and atomically (rename), so its presence means the row is complete.
`--f64_rows` adds the float64 oracle's run of those rows (`{row}-oracle64*`: the yardstick of `oracle/size_parity.py`).

How many workers of how many threads: `oracle/size_parity.start_oracle_workers` (one thread each, as many as the host's CPU quota).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pin(slot, threads):
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if slot is None or threads * (slot + 1) > len(allowed):
        return None
    mine = allowed[slot * threads:(slot + 1) * threads]
    os.sched_setaffinity(0, mine)
    return mine


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="", help="comma-separated row indices of the sample list")
    ap.add_argument("--f64_rows", default="", help="rows to attack with the FLOAT64 oracle as well (the yardstick run; done first)")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--slot", type=int, default=None)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--tiny", action="store_true", help="the tiny backbone (CPU self-test of the plumbing)")
    args = ap.parse_args(argv)
    pin(args.slot, args.threads)
    os.environ["OMP_NUM_THREADS"] = str(args.threads)
    for p in (os.path.join(ROOT, "image-to-video-i2v-attack_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch
    torch.set_num_threads(args.threads)
    from i2v_amd import graphs, weights
    from oracle import restate, size_parity
    g = graphs.build_tiny("resnet", (args.hw, args.hw)) if args.tiny else graphs.build("resnet50", (args.hw, args.hw))
    net = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[3]])
    os.makedirs(args.out, exist_ok=True)
    jobs = [(int(r), "oracle64") for r in args.f64_rows.split(",") if r != ""] + [(int(r), "oracle") for r in args.rows.split(",") if r != ""]
    net64 = None
    for row, tag in jobs:
        done = os.path.join(args.out, f"{row}-{tag}.npz")
        if os.path.exists(done):
            continue
        t0 = time.time()
        vid = size_parity.synthetic_clip(1000 + row, args.frames, args.hw)
        if tag == "oracle64":
            net64 = net64 or restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[3]], dtype=torch.float64)
            ora = size_parity.oracle_attack(net64, vid.double(), steps=args.steps, lr=args.lr)
        else:
            ora = size_parity.oracle_attack(net, vid, steps=args.steps, lr=args.lr)
        np.save(os.path.join(args.out, f"{row}-{tag}-adv.npy"), ora["adv"][0].float().numpy())
        tmp = os.path.join(args.out, f".{row}-{tag}.tmp.npz")
        np.savez(tmp, costs=np.asarray(ora["costs"], np.float64), mean_abs_delta=float(ora["delta"].abs().mean()), seconds=time.time() - t0)
        os.replace(tmp, done)


if __name__ == "__main__":
    main()
