"""TEST INFRASTRUCTURE -- writes `tests/golden/ilaf_i3d_full_size_yardstick.npz`: the COST trajectories of ILAF
(`/root/reference/image_attacks.py:534-629`, restated in `oracle/restate.run_ilaf`) on the non-local I3D at BASELINE.json configs[4]'s
shape -- one clip of 32 x 224 x 224, the clip pair of `tests/test_gpu_video.py::test_native_ilaf_full_size_against_oracle` (seed 11) --
run free for STEPS sign steps by the torch-module oracle in float32 AND in float64 (costs, and two scalars about the two final clips:
a few hundred bytes).

Why.  Sign steps move every element of the perturbation by +-0.005 whatever the size of its gradient, and the I3D's five non-local
(softmax attention) blocks amplify the element-wise drift between any two fp32 runs: the native loop stayed within rtol 2e-4 of the
fp32 oracle for three steps and was 7.5e-4 away at the fourth (round 5) -- asserted to be chaos, without a yardstick.  This file is the
yardstick: the distance of the reference arithmetic's OWN fp32 run from exact arithmetic, step by step.  The device is held to
1.25 x that distance from the float64 run (`test_native_ilaf_full_size_against_oracle[i3d_resnet50]`), for STEPS >= 20 steps; if it
drifted faster than the oracle does, the attention path would have a bug.

    python -m oracle.make_ilaf_yardstick [steps]          (from the repo root; about 15 min on 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "image-to-video-i2v-attack_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

OUT = os.path.join(ROOT, "tests", "golden", "ilaf_i3d_full_size_yardstick.npz")
MT, THW, SEED = "i3d_resnet50", (32, 224, 224), 11


def clip_pair():
    """(adv, ori) of the full-size ILAF test: uint8 noise clip, and the same clip +-10 grey levels (an 'existing adversarial clip')."""
    from tests import golden_util as gu
    gen = torch.Generator().manual_seed(SEED)
    ori_u8 = torch.randint(0, 256, (1, 3, *THW), generator=gen, dtype=torch.uint8)
    adv_u8 = (ori_u8.long() + torch.randint(-10, 11, ori_u8.shape, generator=gen)).clamp(0, 255).to(torch.uint8)
    return gu.videos_of({"clip_u8": adv_u8.numpy()}), gu.videos_of({"clip_u8": ori_u8.numpy()})


def main(steps=24):
    from i2v_amd import graphs, weights
    from oracle import restate, video_models as vm
    adv, ori = clip_pair()
    g = graphs.build_video(MT, THW)
    tm = vm.load_weights(vm.make(MT, False), weights.synthetic_state_dict(g, 0))
    out, clips = {}, {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        t0 = time.time()
        tm.to(dt)
        clip, costs, _, _ = restate.run_ilaf(tm, vm.hook_modules(tm, MT), adv.to(dt), ori.to(dt), steps=steps)
        clips[tag] = clip.double()
        out["costs_" + tag] = np.asarray(costs, np.float64)
        print(tag, f"{time.time() - t0:.0f} s", costs, flush=True)
        np.savez(OUT + ".partial.npz", **out)
    rel = np.abs(out["costs_f32"] - out["costs_f64"]) / np.abs(out["costs_f64"])
    print("fp32 oracle vs float64 oracle, relative cost distance per step:", np.array2string(rel, precision=2))
    # the element-wise distance between the two runs' final clips (sign steps are chaotic element by element: DESIGN.md section 6): the
    # yardstick the device's distance from the live fp32 oracle is held to
    diff = (clips["f32"] - clips["f64"]).abs()
    out["mean_abs_out_f32_f64"] = float(diff.mean())
    out["frac_differing_f32_f64"] = float((diff > 0).double().mean())
    print("final clips, fp32 oracle vs float64 oracle: mean|diff|", out["mean_abs_out_f32_f64"], "elements differing", out["frac_differing_f32_f64"])
    np.savez(OUT, model_type=MT, seed=SEED, steps=steps, step_size=0.005, threads=torch.get_num_threads(), **out)
    os.remove(OUT + ".partial.npz")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
