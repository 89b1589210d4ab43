"""CPU: the free-running rung's machinery (`oracle/size_parity.py`: oracle attack, float64 yardstick, statistics, bounds) on a tiny
backbone, with the planner's host simulation standing in for the device -- the `-m gpu` twin at BASELINE size is
tests/test_gpu_size_parity.py.  Rung: `/root/reference/image_attacks.py:325-364`, SURVEY.md 7.3-1 (iv)."""
import os

import numpy as np
import torch

from i2v_amd import attacks, graphs, weights
from oracle import restate, size_parity
from tests.hostsim_util import hostsim_engine


def test_rung_statistics_on_the_host_simulation():
    eng = hostsim_engine()
    vid = size_parity.synthetic_clip(1000, frames=4, hw=64)
    g = graphs.build_tiny("resnet", (64, 64))
    sd = weights.synthetic_state_dict(g, 0)
    net32 = restate.OracleNet(g, sd, [g.hooks[3]])
    net64 = restate.OracleNet(g, sd, [g.hooks[3]], dtype=torch.float64)
    ora32 = size_parity.oracle_attack(net32, vid, steps=6, lr=0.005)
    ref = restate.run_attack([net32], vid, steps=6, step_size=0.005)          # the same arithmetic as the pinned restatement
    assert np.array_equal(ora32["costs"], ref["costs"]) and torch.equal(ora32["adv"], ref["adv"]) and torch.equal(ora32["delta"], ref["delta"])
    ora64, y32 = size_parity.yardstick(net64, vid, ora32, steps=6, lr=0.005)
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=3, step_size=0.005, steps=6, engine=eng,
                                              graph_builder=graphs.build_tiny, weight_seed=0)
    adv = atk(vid, torch.zeros(1, dtype=torch.long), ["c"])
    st32 = size_parity.compare(atk.last_costs, atk._delta, adv, ora32)
    st64 = size_parity.compare(atk.last_costs, atk._delta, adv, ora64)
    ok, bad = size_parity.within_bounds(st32, st64, y32)
    assert ok, (bad, st32, st64, y32)
    # a run that is NOT the same attack fails the well-conditioned bounds
    atk2 = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=3, step_size=0.004, steps=6, engine=eng,
                                               graph_builder=graphs.build_tiny, weight_seed=0)
    adv2 = atk2(vid, torch.zeros(1, dtype=torch.long), ["c"])
    ok2, bad2 = size_parity.within_bounds(size_parity.compare(atk2.last_costs, atk2._delta, adv2, ora32))
    assert not ok2 and bad2


def test_committed_float64_yardstick_fixture():
    """`tests/golden/size_parity_f64_seed1000.npz` (oracle/make_size_yardstick.py): what `bench.py`'s default `parity_check` holds the
    perturbed pixels to.  The file is for clip seed 1000 / 10 steps / lr 0.005 only; its first cost is the 32 frames' cos = 1 at
    delta_0; `compare_sampled` of the run against itself is the identity; another configuration gets no yardstick."""
    import os
    import numpy as np
    import torch
    from oracle import size_parity
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    y = size_parity.load_yardstick(gold)
    assert y is not None and y["costs"].shape == (10,) and abs(float(y["costs"][0]) - 32.0) < 1e-6
    assert np.all(np.diff(y["costs"]) < 0)
    assert int(y["numel"]) == 3 * 32 * 224 * 224 and y["adv_sample"].shape[0] == -(-int(y["numel"]) // int(y["stride"]))
    assert size_parity.load_yardstick(gold, seed=1001) is None and size_parity.load_yardstick(gold, steps=3) is None
    adv = torch.zeros(1, 3, 32, 224, 224)
    adv.reshape(-1)[::int(y["stride"])] = torch.from_numpy(y["adv_sample"])
    st = size_parity.compare_sampled(y["costs"], torch.full((8,), float(y["mean_abs_delta"])), adv, y)
    assert st["max_rel_cost_err"] == 0 and st["mean_abs_adv_diff"] == 0 and st["frac_pixels_within_2lr"] == 1.0
    assert abs(st["mean_abs_delta_ratio"] - 1) < 1e-6
    # the sample's values are normalised pixels of a clamped clip: inside the ImageNet-normalised [0, 1] box
    assert float(y["adv_sample"].min()) >= (0 - 0.485) / 0.229 - 1e-5 and float(y["adv_sample"].max()) <= (1 - 0.406) / 0.225 + 1e-5


def test_oracle_worker_processes_on_the_tiny_backbone(tmp_path):
    """`oracle/fooling_worker.py` through `size_parity.start_oracle_workers` (the plumbing of tests/test_gpu_size_parity.py and
    tools/fooling_parity.py), on the tiny backbone: single-thread fp32 workers dealt round-robin + the float64 worker; a row is complete
    when its npz appears; the fp32 and float64 runs of one row agree to fp32 rounding; `effective_cpus` honours the cgroup quota."""
    from oracle import size_parity
    n = size_parity.effective_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    procs = size_parity.start_oracle_workers([0, 1, 2], str(tmp_path), workers=2, steps=2, extra=["--tiny", "--frames", "4", "--hw", "64"],
                                             f64_rows=[1], f64_threads=2)
    assert len(procs) == 3
    rows = [size_parity.wait_oracle_row(str(tmp_path), r, procs, timeout=300) for r in range(3)]
    o64 = size_parity.wait_oracle_row(str(tmp_path), 1, procs, timeout=300, tag="oracle64")
    assert [p.wait() for p in procs] == [0, 0, 0]
    for o in rows:
        assert o["costs"].shape == (2,) and abs(float(o["costs"][0]) - 4.0) < 1e-5 and o["adv"].shape == (1, 3, 4, 64, 64)
    st = size_parity.compare(rows[1]["costs"], rows[1]["mean_abs_delta"], rows[1]["adv"], o64)
    assert st["max_rel_cost_err"] < 1e-5 and abs(st["mean_abs_delta_ratio"] - 1) < 0.01
