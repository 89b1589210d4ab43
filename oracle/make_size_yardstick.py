"""TEST INFRASTRUCTURE -- writes `tests/golden/size_parity_f64_seed1000.npz`: the float64 oracle's run of BASELINE.json configs[0]
(ResNet-50 layer3, clip seed 1000, 1 x 32 x 224^2, 10 steps, lr 0.005) reduced to what `bench.py`'s default `parity_check` needs to
hold the device's perturbed pixels to the yardstick WITHOUT paying for the float64 run (about a minute on the GPU box's host, ten
here): the ten costs, mean|delta_10|, and every STRIDE-th element of the perturbed clip (float32; 4.8 M / 41 = 117 k values).

    python -m oracle.make_size_yardstick          (from the repo root; ~10 min on 8 cores)

The float64 run is reproducible to far below fp32 resolution whatever the host (its own rounding noise is 1e-16; the chaotic
amplification that makes fp32 runs differ by 2*lr on 12 % of the pixels needs 1e-7), so one committed run serves every box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "image-to-video-i2v-attack_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

STRIDE = 41
OUT = os.path.join(ROOT, "tests", "golden", "size_parity_f64_seed1000.npz")


def main():
    from i2v_amd import graphs, weights
    from oracle import restate, size_parity
    g = graphs.build("resnet50", (224, 224))
    net64 = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[3]], dtype=torch.float64)
    vid = size_parity.synthetic_clip(1000)
    ora = size_parity.oracle_attack(net64, vid.double(), steps=10, lr=0.005)
    adv = ora["adv"].float().reshape(-1)
    np.savez_compressed(OUT, costs=np.asarray(ora["costs"], np.float64), mean_abs_delta=float(ora["delta"].abs().mean()),
                        stride=STRIDE, numel=adv.numel(), adv_sample=adv[::STRIDE].numpy(), seed=1000, steps=10, lr=0.005)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; costs", ora["costs"])


if __name__ == "__main__":
    main()
