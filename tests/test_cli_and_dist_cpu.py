"""CPU: the drop-in CLI's file contract (image_main.py) and the multi-process path (gloo,
world_size 2) -- both on the host simulation of the kernel backend, tiny backbones."""
import json
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.hostsim_util import hostsim_engine


@pytest.fixture
def tiny_engine(monkeypatch):
    from i2v_amd import attacks, graphs
    eng = hostsim_engine()
    monkeypatch.setitem(attacks._ENGINES, attacks.default_device(), eng)
    monkeypatch.setattr(graphs, "build", graphs.build_tiny)
    return eng


def test_image_main_file_contract(tiny_engine, tmp_path, monkeypatch):
    """`{label}-adv.npy` per clip (float32 (3,f,h,w), normalised), `loss_info_{i}.json`, shard
    window arithmetic of /root/reference/image_main.py:45,61-63,90-95."""
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    import importlib
    import image_main
    importlib.reload(image_main)
    common = ["--attack_method", "ImageGuidedFMDirection_Adam", "--step", "2", "--step_size", "0.005", "--depth", "2",
              "--direction_image_model", "resnet", "--num_clips", "4", "--frames", "2", "--hw", "64",
              "--file_prefix", "t", "--batch_nums", "2"]
    image_main.main(common + ["--batch_index", "2"])
    out = tmp_path / "Image-ImageGuidedFMDirection_Adam-2-t"
    assert sorted(os.listdir(out)) == ["2-adv.npy", "3-adv.npy", "loss_info_2.json"]     # second half of 4 clips
    adv = np.load(out / "2-adv.npy")
    assert adv.dtype == np.float32 and adv.shape == (3, 2, 64, 64)
    info = json.load(open(out / "loss_info_2.json"))
    name = list(info)[0]
    assert list(info[name]) == ["0", "1"] and float(info[name]["0"]["cost"]) > 1.9
    # evaluator-side label parsing (reference.py:43): int(fname.split('-')[0])
    assert int("2-adv.npy".split("-")[0]) == 2
    image_main.main(common + ["--batch_index", "1", "--resume"])
    assert sorted(os.listdir(out)) == ["0-adv.npy", "1-adv.npy", "2-adv.npy", "3-adv.npy", "loss_info_1.json",
                                       "loss_info_2.json"]
    # default attack name does not exist, as in the reference (image_main.py:25)
    with pytest.raises(AttributeError):
        image_main.main(["--num_clips", "1"])


def test_image_main_grouped_batches_are_byte_identical(tiny_engine, tmp_path, monkeypatch):
    """`--group_clips`: ready loader batches share one engine call.  Every `{label}-adv.npy` and every logged cost string must
    equal what one call per loader batch (`--group_clips 1`, the reference's behaviour) writes -- also with clip lanes on."""
    import importlib
    import image_main
    outs = {}
    for tag, extra, lanes in (("one", ["--group_clips", "1"], "1"), ("grp", ["--group_clips", "3"], "1"), ("grp_lanes", ["--group_clips", "4"], "2")):
        monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path / tag))
        monkeypatch.setenv("I2V_CLIP_LANES", lanes)
        (tmp_path / tag).mkdir()
        importlib.reload(image_main)
        image_main.main(["--attack_method", "ImageGuidedFMDirection_Adam", "--step", "3", "--step_size", "0.005", "--depth", "2",
                         "--direction_image_model", "resnet", "--num_clips", "5", "--frames", "2", "--hw", "64", "--file_prefix", "t"] + extra)
        d = tmp_path / tag / "Image-ImageGuidedFMDirection_Adam-3-t"
        outs[tag] = ({f: np.load(d / f) for f in sorted(os.listdir(d)) if f.endswith(".npy")}, json.load(open(d / "loss_info_1.json")))
    files, info = outs["one"]
    assert len(files) == 5 and len(info) == 5
    for tag in ("grp", "grp_lanes"):
        f2, i2 = outs[tag]
        assert sorted(f2) == sorted(files) and all(np.array_equal(f2[k], files[k]) for k in files), tag
        assert i2 == info, tag


def test_image_main_reader_failure_ends_the_run(tiny_engine, tmp_path, monkeypatch):
    """A clip file that cannot be read raises out of `main` (after the clips before it were written) instead of leaving the main
    loop waiting on its queue."""
    import importlib
    import image_main
    clip_dir = tmp_path / "clips"
    clip_dir.mkdir()
    np.save(clip_dir / "1-ori.npy", np.zeros((3, 2, 64, 64), np.float32))
    (clip_dir / "2-ori.npy").write_bytes(b"not a numpy file")
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    importlib.reload(image_main)
    with pytest.raises(Exception):
        image_main.main(["--attack_method", "ImageGuidedFMDirection_Adam", "--step", "1", "--step_size", "0.005", "--depth", "2",
                         "--direction_image_model", "resnet", "--clip_dir", str(clip_dir), "--file_prefix", "e", "--workers", "0", "--group_clips", "1"])
    assert os.path.exists(tmp_path / "Image-ImageGuidedFMDirection_Adam-1-e" / "1-adv.npy")


def test_image_main_ucf101_twin(tiny_engine, tmp_path, monkeypatch):
    """`image_main_ucf101.py`: raw 240 x 320 uint8 clips go through the UCF-101 transform (PIL Scale + centre crop) on the engine;
    --step defaults to 10; names are the characters of `str(val_label)` (image_main_ucf101.py:83 feeds a string to the attack)."""
    import importlib
    import image_main
    import image_main_ucf101
    from oracle import restate
    clip_dir = tmp_path / "clips"
    clip_dir.mkdir()
    rng = np.random.default_rng(5)
    raws = {}
    for label in (4, 9):
        raws[label] = rng.integers(0, 256, (2, 60, 80, 3), dtype=np.uint8)          # (t, H, W, 3) decoded frames
        np.save(clip_dir / f"{label}-raw.npy", raws[label])
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    importlib.reload(image_main); importlib.reload(image_main_ucf101)
    image_main_ucf101.main(["--attack_method", "ImageGuidedFMDirection_Adam", "--step_size", "0.005", "--depth", "2", "--direction_image_model", "resnet",
                            "--clip_dir", str(clip_dir), "--hw", "48", "--file_prefix", "u", "--group_clips", "1"])
    out = tmp_path / "Image-ImageGuidedFMDirection_Adam-10-u"                       # --step defaulted to 10
    assert sorted(os.listdir(out)) == ["4-adv.npy", "9-adv.npy", "loss_info_1.json"]
    adv = np.load(out / "4-adv.npy")
    clean = restate.ucf101_transform(raws[4][None], 48, 48)[0].numpy()            # what the attack started from
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1, 1)
    assert adv.shape == clean.shape == (3, 2, 48, 48) and 0 < np.abs((adv - clean) * std).max() <= 16 / 255 + 1e-5
    info = json.load(open(out / "loss_info_1.json"))
    assert set(info) == set("tensor([9])") | set("tensor([4])") and list(info["t"]) == [str(i) for i in range(10)]


def test_image_fine_tune_attack_file_contract(tiny_engine, tmp_path):
    """`{id}-adv.npy` + `{id}-ori.npy` in, `{label}-adv.npy` out (/root/reference/image_fine_tune_attack.py:16-37,73-82),
    ILAF running natively on a (tiny) I3D graph."""
    import image_fine_tune_attack as ift
    adv_dir, ori_dir, out_dir = tmp_path / "adv", tmp_path / "ori", tmp_path / "out"
    adv_dir.mkdir(); ori_dir.mkdir()
    gen = torch.Generator().manual_seed(1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1, 1)
    for vid in (3, 17):
        u8 = torch.randint(16, 240, (3, 8, 32, 32), generator=gen)
        ori = (u8.float() / 255 - mean) / std
        adv = ((u8 + torch.randint(-8, 9, u8.shape, generator=gen)).float() / 255 - mean) / std
        np.save(ori_dir / f"{vid}-ori.npy", ori.numpy())
        np.save(adv_dir / f"{vid}-adv.npy", adv.numpy())
    argv = ["--used_adv", str(adv_dir), "--used_ori", str(ori_dir), "--opt_path", str(out_dir), "--white_model", "i3d_resnet50",
            "--steps", "2"]
    atk = ift.main(argv, model_kwargs=dict(tiny=True))
    assert sorted(os.listdir(out_dir)) == ["17-adv.npy", "3-adv.npy"]
    out = np.load(out_dir / "3-adv.npy")
    assert out.dtype == np.float32 and out.shape == (3, 8, 32, 32)
    assert list(atk.loss_info["..."].keys()) == [0, 1] and abs(float(atk.loss_info["..."][0]["cost"]) + 1.5) < 1e-4
    # the default groups clips into one engine call (independent one-clip problems): byte-identical to one clip per call
    one = tmp_path / "one_per_call"
    ift.main(argv[:5] + [str(one)] + argv[6:] + ["--group_clips", "1", "--streams", "1"], model_kwargs=dict(tiny=True))
    for f in ("3-adv.npy", "17-adv.npy"):
        assert np.array_equal(np.load(one / f), np.load(out_dir / f)), f
    before = os.path.getmtime(out_dir / "3-adv.npy")
    assert ift.main(argv + ["--resume"], model_kwargs=dict(tiny=True)) is None      # nothing left to do
    assert os.path.getmtime(out_dir / "3-adv.npy") == before
    with pytest.raises(KeyError):
        ift.main(argv[:-2] + ["--white_model", "i3d_nl10_resnet50"])          # (not a reference configuration, utils.py:9-14)


def test_attack_cli_feeds_fine_tune(tiny_engine, tmp_path, monkeypatch):
    """`attack.py --attack_type image` writes `{label}-adv.npy` + `{label}-ori.npy` (/root/reference/attack.py:102-108)
    within eps of each other, and `image_fine_tune_attack.py` consumes exactly that directory."""
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    import importlib
    import attack as attack_cli
    importlib.reload(attack_cli)
    out = attack_cli.main(["--attack_method", "BIM", "--step", "2", "--model", "i3d_resnet50", "--num_clips", "2", "--batch_size", "1",
                           "--frames", "32", "--hw", "16", "--file_prefix", "t", "--kernlen", "15", "--noise",
                           "--model_factory", "reference:proxy"])     # (file contract only: a small torch classifier; the default is 'native')
    assert out == str(tmp_path / "i3d_resnet50-BIM-2-t")
    assert sorted(os.listdir(out)) == ["0-adv.npy", "0-ori.npy", "1-adv.npy", "1-ori.npy"]
    adv, ori = np.load(os.path.join(out, "1-adv.npy")), np.load(os.path.join(out, "1-ori.npy"))
    assert adv.shape == ori.shape == (3, 32, 16, 16) and adv.dtype == np.float32
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1, 1)
    assert 0 < np.abs((adv - ori) * std).max() <= 16 / 255 + 1e-6
    with pytest.raises(AttributeError):
        attack_cli.main(["--num_clips", "1", "--frames", "32", "--hw", "16"])          # the reference's default method name does not exist
    with pytest.raises(UnboundLocalError):          # a video attack other than TemporalTranslation: `spe_params` is never bound (attack.py:78-82)
        attack_cli.main(["--attack_type", "video", "--attack_method", "BIM"])
    import image_fine_tune_attack as ift
    ift.main(["--used_adv", out, "--used_ori", out, "--opt_path", str(tmp_path / "ft"), "--white_model", "slowfast_resnet50",
              "--steps", "1"], model_kwargs=dict(tiny=True))
    assert sorted(os.listdir(tmp_path / "ft")) == ["0-adv.npy", "1-adv.npy"]


def test_image_fine_tune_attack_rank_sharding(tiny_engine, tmp_path, monkeypatch):
    """ILAF shards as replicas only: under torchrun the clip files are dealt round-robin over the
    ranks, no collective; the union over ranks is every clip exactly once and equals the single-process result."""
    import image_fine_tune_attack as ift
    adv_dir, ori_dir = tmp_path / "adv", tmp_path / "ori"
    adv_dir.mkdir(); ori_dir.mkdir()
    gen = torch.Generator().manual_seed(2)
    for vid in (1, 4, 9):
        ori = torch.rand(3, 8, 32, 32, generator=gen)
        np.save(ori_dir / f"{vid}-ori.npy", ori.numpy())
        np.save(adv_dir / f"{vid}-adv.npy", (ori + 0.05 * torch.randn(3, 8, 32, 32, generator=gen)).numpy())
    base = ["--used_adv", str(adv_dir), "--used_ori", str(ori_dir), "--white_model", "slowfast_resnet50", "--steps", "1"]
    ift.main(base + ["--opt_path", str(tmp_path / "single")], model_kwargs=dict(tiny=True))
    monkeypatch.setenv("WORLD_SIZE", "2")
    done = []
    for rank in (0, 1):
        monkeypatch.setenv("RANK", str(rank))
        out = tmp_path / f"r{rank}"
        ift.main(base + ["--opt_path", str(out)], model_kwargs=dict(tiny=True))
        done.append(sorted(os.listdir(out)))
    assert done == [["1-adv.npy", "9-adv.npy"], ["4-adv.npy"]]
    for rank, files in enumerate(done):
        for f in files:
            assert np.array_equal(np.load(tmp_path / f"r{rank}" / f), np.load(tmp_path / "single" / f))


def test_sample_list_fixture():
    """The reference's sample list (data fixture): 400 rows, one clip per class, labels 0..399."""
    from i2v_amd import clips
    rows = clips.sample_list(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kinetics400_attack_samples.csv"))
    assert len(rows) == 400 and sorted(l for _, l in rows) == list(range(400))
    assert rows[0] == ("abseiling/YqTT34PsD5c_000003_000013.mp4", 0)
    assert clips.sample_list(None, 7)[6] == ("synthetic/006.mp4", 6)


def test_run_image_guided_plan():
    import run_image_guided
    jobs = run_image_guided.plan("0", 1)
    assert len(jobs) == 25 + 16 + 9 + 9           # Figure 4, Table 2, Table 3, Table 4
    assert any("ImageGuidedFML2_Adam_MultiModels" in j[0] for j in jobs)
    assert jobs[0][1][-1] == "Image-ImageGuidedFMDirection_Adam-20-resnet_step_size_0.001_paper_study"
    ucf = [j for j in jobs if j[0][1].endswith("image_main_ucf101.py")]
    assert len(ucf) == 9 and all(j[1][1].endswith("reference_ucf101.py") for j in ucf)
    for a, e in jobs:                             # every script the sweep starts exists next to it
        assert os.path.isfile(a[1]) and os.path.isfile(e[1])


def test_run_image_guided_matches_the_reference_commands(monkeypatch, capsys):
    """The reference's own `run_image_guided.py` executed with `os.system` recording instead of running: its formatted
    command lines (templates `:5-29`, loops `:44-100`), in order, against `plan()` flag for flag -- and against what
    `--dry_run` prints."""
    import runpy
    import sys
    from oracle import ref_shim
    path = os.path.join(ref_shim.REFERENCE_DIR, "run_image_guided.py")
    if not os.path.isfile(path):
        pytest.skip("reference not present")
    import run_image_guided
    for gpu, bs in (("0", 1), ("3", 4)):
        issued = []
        monkeypatch.setattr(os, "system", lambda c: issued.append(c.split()) or 0)
        monkeypatch.setattr(sys, "argv", ["run_image_guided.py", "--gpu", gpu, "--batch_size", str(bs)])
        runpy.run_path(path, run_name="__main__")
        monkeypatch.undo()
        ours = [c for pair in run_image_guided.plan(gpu, bs) for c in pair]
        assert len(issued) == len(ours) == 2 * 59
        for ref, mine in zip(issued, ours):
            assert ref[0] == "python" and mine[0] == sys.executable
            assert ref[1] == os.path.basename(mine[1])
            assert ref[2:] == mine[2:], (ref, mine)
        monkeypatch.delenv("I2V_EVAL_CMD", raising=False)
        monkeypatch.delenv("I2V_EVAL_ARGS", raising=False)
        run_image_guided.main(["--gpu", gpu, "--batch_size", str(bs), "--dry_run"])
        printed = [l.split() for l in capsys.readouterr().out.splitlines() if l.strip()]
        assert [p[2:] for p in printed] == [r[2:] for r in issued]


def _aens_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from i2v_amd import attacks, graphs
    from tests import golden_util as gu
    fx = gu.load("aens_2x2_f64")
    vid = gu.videos_of(fx)
    atk = attacks.AENS_I2V_MF(fx["models"], depths=fx["depth"], step_size=fx["lr"], steps=fx["steps"],
                              engine=hostsim_engine(), graph_builder=graphs.build_tiny, **fx["kw"])
    shard = vid[rank:rank + 1]                                            # one clip per rank
    adv, _, costs = atk(shard, torch.zeros(1, dtype=torch.long), [f"c{rank}"])
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), adv=adv.numpy(), costs=costs, weights=np.stack(atk.weights))
    dist.destroy_process_group()


def test_aens_two_ranks_match_single_device(tmp_path):
    """2 ranks x 1 clip with the per-step all-reduce == 1 device x 2 clips (the golden fixture):
    same weight trajectory and cost (TPAMI_attack.py:265,289-297), same adversarial clips."""
    from tests import golden_util as gu
    fx = gu.load("aens_2x2_f64")
    port = 29500 + os.getpid() % 2000
    mp.start_processes(_aens_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    np.testing.assert_allclose(r0["weights"], fx["weights"], rtol=1e-4)
    np.testing.assert_array_equal(r0["weights"], r1["weights"])
    np.testing.assert_allclose(r0["costs"], fx["cost_saved"], rtol=2e-4)
    adv = np.concatenate([r0["adv"], r1["adv"]])
    assert np.abs(adv - fx["adv"]).mean() < 5e-3


def _aens_uneven_worker(rank, world, port, out_dir, counts):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from i2v_amd import attacks, graphs
    from oracle.make_golden import make_clip, normalise
    vid = normalise(make_clip(911, sum(counts), 2, 48), torch.float32)
    lo = sum(counts[:rank])
    atk = attacks.AENS_I2V_MF(["resnet", "vgg"], depths={"resnet": [2, 3], "vgg": [2, 3]}, step_size=0.005, steps=3, momentum=0.5,
                              engine=hostsim_engine(), graph_builder=graphs.build_tiny, weight_seed=0)
    adv, _, costs = atk(vid[lo:lo + counts[rank]], torch.zeros(counts[rank], dtype=torch.long), [f"c{lo + i}" for i in range(counts[rank])])
    np.savez(os.path.join(out_dir, f"u{rank}.npz"), adv=adv.numpy(), costs=costs, weights=np.stack(atk.weights))
    dist.destroy_process_group()


def test_aens_four_ranks_unequal_shards_match_single_device(tmp_path):
    """four ranks holding 2 / 1 / 1 / 1 clips.  The adaptive weights depend on sums over the GLOBAL batch
    (`TPAMI_attack.py:265,293-297`: the inner softmax is not scale invariant), so every rank must see the weights and the cost
    of the single-device run over all five clips, whatever its own share -- against the float64 oracle on the five clips."""
    from oracle import restate
    from oracle.make_golden import make_clip, normalise
    from i2v_amd import graphs, weights
    counts = (2, 1, 1, 1)
    vid = normalise(make_clip(911, sum(counts), 2, 48), torch.float32)
    nets = []
    for m in ("resnet", "vgg"):
        g = graphs.build_tiny(m, (48, 48))
        nets.append(restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hook_for(2, True), g.hook_for(3, True)], dtype=torch.float64))
    ref = restate.run_attack(nets, vid.double(), steps=3, step_size=0.005, mode="aens", momentum=0.5, trace=True,
                             coeffs=torch.ones(4, dtype=torch.float64))
    port = 33500 + os.getpid() % 2000
    mp.start_processes(_aens_uneven_worker, args=(4, port, str(tmp_path), counts), nprocs=4, join=True, start_method="spawn")
    rs = [np.load(tmp_path / f"u{r}.npz") for r in range(4)]
    for r in rs:
        np.testing.assert_array_equal(r["weights"], rs[0]["weights"])          # one global weight trajectory
        np.testing.assert_array_equal(r["costs"], rs[0]["costs"])
    np.testing.assert_allclose(rs[0]["weights"], np.stack(ref["weights"]), rtol=1e-4)
    np.testing.assert_allclose(rs[0]["costs"], ref["costs"], rtol=2e-4)
    adv = np.concatenate([r["adv"] for r in rs])
    assert adv.shape[0] == 5 and np.abs(adv - ref["adv"].float().numpy()).mean() < 5e-3


def test_affinity_plan_is_numa_and_smt_aware():
    """Per-rank CPU placement (i2v_amd/affinity.py): ranks dealt to NUMA nodes in blocks, a node's cores cut evenly, every
    allowed core used at most once, nothing outside the allowed set."""
    from i2v_amd import affinity
    nodes = [list(range(0, 64)), list(range(64, 128))]
    got = [affinity.plan(r, 8, nodes) for r in range(8)]
    assert [(g[0], g[-1], len(g)) for g in got] == [(16 * r, 16 * r + 15, 16) for r in range(8)]
    assert affinity.plan(0, 1, nodes) == nodes[0] + nodes[1]                   # one rank: everything
    assert affinity.plan(1, 2, nodes) == nodes[1]
    three = [affinity.plan(r, 3, [list(range(8))]) for r in range(3)]
    assert sorted(c for g in three for c in g) == list(range(8))
    assert affinity.plan(5, 6, [[0, 1], [2, 3]]) == [2, 3]                      # more ranks than cores on a node: shared whole
    with pytest.raises(ValueError):
        affinity.plan(4, 4, nodes)
    # sysfs parsing: hyper-thread siblings end up next to each other
    assert affinity._parse_cpulist("0-3,8-11\n") == [0, 1, 2, 3, 8, 9, 10, 11]


def _dr_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from i2v_amd import attacks, graphs
    from oracle.make_golden import make_clip, normalise
    vid = normalise(make_clip(77, 2, 2, 64), torch.float32)
    atk = attacks.ImageGuidedStd_Adam(["resnet"], depth=2, step_size=0.005, steps=3, engine=hostsim_engine(),
                                      graph_builder=graphs.build_tiny)
    adv = atk(vid[rank:rank + 1], torch.zeros(1, dtype=torch.long), [f"c{rank}"])
    np.savez(os.path.join(out_dir, f"dr{rank}.npz"), adv=adv.numpy(), costs=atk.last_costs)
    dist.destroy_process_group()


def test_dr_two_ranks_match_single_device(tmp_path):
    """Dispersion-Reduction couples all frames of the batch through one std (image_attacks.py:218):
    2 ranks x 1 clip with the 3-double all-reduce == the oracle on both clips at once."""
    from oracle import restate
    from oracle.make_golden import make_clip, normalise
    from i2v_amd import graphs, weights
    vid = normalise(make_clip(77, 2, 2, 64), torch.float32)
    g = graphs.build_tiny("resnet", (64, 64))
    net = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[2]], dtype=torch.float64)
    ref = restate.run_attack([net], vid.double(), steps=3, step_size=0.005, mode="std")
    port = 31500 + os.getpid() % 2000
    mp.start_processes(_dr_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "dr0.npz"), np.load(tmp_path / "dr1.npz")
    np.testing.assert_allclose(r0["costs"], ref["costs"], rtol=2e-4)
    np.testing.assert_array_equal(r0["costs"], r1["costs"])
    adv = np.concatenate([r0["adv"], r1["adv"]])
    assert np.abs(adv - ref["adv"].float().numpy()).mean() < 5e-3


def label_reader(name):
    """Model factory for the evaluator test: 'reads' the label planted in the clip."""
    class M(torch.nn.Module):
        def forward(self, x):
            idx = x[:, 0, 0, 0, 0].round().long() + (1 if name == "off_by_one" else 0)
            return torch.nn.functional.one_hot(idx.clamp(0, 9), 10).float()
    return M()


def test_evaluator_contract(tmp_path, monkeypatch):
    """reference.py contract: discovers `*adv*` files, label from the file name, top-1 per model,
    `top1_acc_all_models.json` + `results_all_models_prediction.csv`."""
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path))
    d = tmp_path / "run"
    d.mkdir()
    for label in (3, 0, 7, 5):
        clip = np.zeros((3, 2, 4, 4), np.float32)
        clip[0, 0, 0, 0] = label if label != 7 else 2          # clip 7 is "fooled"
        np.save(d / f"{label}-adv.npy", clip)
    (d / "loss_info_1.json").write_text("{}")
    import importlib
    import reference as ev
    importlib.reload(ev)
    acc = ev.main(["--adv_path", "run", "--models", "exact,off_by_one", "--model_factory",
                   "tests.test_cli_and_dist_cpu:label_reader", "--batch_size", "3"])
    assert acc == {"exact": 75.0, "off_by_one": 0.0}
    assert json.load(open(d / "top1_acc_all_models.json")) == acc
    rows = (d / "results_all_models_prediction.csv").read_text().strip().split("\n")
    assert rows[0] == "gt_label,exact-pre,off_by_one-pre" and len(rows) == 5
    assert rows[1:] == ["0,0,1", "3,3,4", "5,5,6", "7,2,3"]
    assert ev.main(["--adv_path", "run", "--models", "i3d_resnet50"])["i3d_resnet50"] >= 0.0     # built-in proxy runs


# ------------------------------------------------------------------ bench.py --gpus N: rank spawning
def _bench(args, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, I2V_QUIET_WEIGHTS="1", I2V_PIN_CPUS="1", **(env or {}))
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_bench_gpus_n_spawns_n_ranks():
    """`python bench.py --gpus 2` must BE two ranks (the flag used to be parsed and ignored): the parent
    starts one process per device before touching the GPU, the ranks prove themselves with an all-reduce of their
    ids, rank 0 prints ONE line with n_gpus == 2 and whole-job frames/s.  Here on the host simulation over gloo; the
    AENS workload puts its per-step all-reduce inside the timed loop and every rank sees the same (global) weights."""
    code, lines, err = _bench(["--gpus", "2", "--selftest-hostsim", "--steps", "1", "--workload", "aens"])
    assert code == 0, err
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and len(out["per_gpu"]) == 2 and out["value"] > 0
    assert out["ranks_proved_by_allreduce"] == {"sum_of_rank_ids": 1, "expected": 1}
    assert "not a measurement" in out["metric"]
    assert abs(sum(out["aens_weights_last"]) - 1) < 1e-5
    assert out["per_gpu_min"] == min(out["per_gpu"]) and out["per_gpu_max"] == max(out["per_gpu"])
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 2:          # rank 0 pinned itself to its half of the allowed cores before doing anything else
        assert out["cpu_affinity_rank0"]["cores"] == ncpu // 2, out["cpu_affinity_rank0"]
    code, lines, err = _bench(["--gpus", "1", "--selftest-hostsim", "--steps", "1", "--n1-value", "1000"])
    assert code == 0 and lines[0]["n_gpus"] == 1
    assert abs(lines[0]["efficiency_vs_n1"] - lines[0]["value"] / 1000) < 1e-3 and lines[0]["cpu_affinity_rank0"] is None


def test_bench_eight_ranks_aens_weights_follow_the_global_batch():
    """`bench.py --gpus 8 --workload aens` as it will run on the 8-GPU node, here on the host simulation over gloo --
    eight ranks x 2 clips, the 2L-float all-reduce of `AENS_I2V_MF._exchange` inside every step (`TPAMI_attack.py:265,293-297`): the
    weight trajectory is bit-identical on all eight ranks and equals the ONE-device run over the global batch of 16 clips (float64
    oracle), because the inner softmax sees the sums over all b*f frames."""
    from oracle import restate
    from i2v_amd import graphs, weights
    import bench
    code, lines, err = _bench(["--gpus", "8", "--selftest-hostsim", "--steps", "1", "--workload", "aens"], env={"OMP_NUM_THREADS": "1"})
    assert code == 0, err[-2000:]
    out = lines[0]
    assert out["n_gpus"] == 8 and len(out["per_gpu"]) == 8
    assert out["ranks_proved_by_allreduce"] == {"sum_of_rank_ids": 28, "expected": 28}
    assert out["aens_weights_identical_on_all_ranks"] is True
    vids = torch.cat([bench.synthetic_clips(1, seed0=1000 + k)[:, :, :2, :32, :32] for k in range(16)]).contiguous()     # selftest_hostsim's clips
    nets = []
    for m in ("resnet", "vgg"):
        g = graphs.build_tiny(m, (32, 32))
        nets.append(restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hook_for(2, True), g.hook_for(3, True)], dtype=torch.float64))
    ref = restate.run_attack(nets, vids.double(), steps=2, step_size=0.005, mode="aens", coeffs=torch.ones(4, dtype=torch.float64))
    np.testing.assert_allclose(np.array(out["aens_weights"]), np.stack(ref["weights"]), rtol=1e-4)


def test_bench_failed_rank_fails_the_run():
    code, lines, err = _bench(["--gpus", "2", "--selftest-hostsim", "--steps", "1"], env={"I2V_BENCH_SELFTEST_FAIL_RANK": "1"})
    assert code != 0 and "rank(s) failed" in err and not lines
