"""CPU: the graph planner/executor of csrc/i2v_engine.cpp (weight packing, k-tables, stride-parity
classes, addend/mask fusion, arena) driven through the C ABI with the scalar host backend of
tests/hostsim/, against the oracle.  The HIP kernels themselves are checked by the -m gpu tests."""
import ctypes
import os

import numpy as np
import pytest
import torch

from i2v_amd import graphs, weights, attacks
from oracle import restate
from tests import golden_util as gu
from tests.hostsim_util import hostsim_engine

CASES = [("resnet", [3], 64), ("resnet", [2, 3], 64), ("resnet", [1], 32), ("vgg", [2], 32), ("vgg", [3], 32),
         ("alexnet", [3], 64), ("alexnet", [2, 4], 64), ("squeezenet", [2], 64), ("squeezenet", [2, 3], 64),
         ("squeezenet", [4], 64), ("densenet121", [1], 64), ("densenet121", [3], 64), ("densenet121", [2, 4], 64)]


def write_hook_grads(net, feats, hg, N):
    """Put d(cost)/d(hook) where the library expects it, gated by the hook's own ReLU (what
    i2v_cossim_fwd_bwd_f32 does on the device)."""
    for i, hi in enumerate(net.hooks):
        gate = (feats[i] > 0).to(hg[i].dtype) if hi.post_relu else torch.ones_like(feats[i])
        flat = (hg[i] * gate).float().reshape(hg[i].shape[0], -1).contiguous()
        for n in range(flat.shape[0]):
            ctypes.memmove(hi.grad + 4 * n * hi.grad_stride, flat[n].data_ptr(), 4 * hi.D)


@pytest.mark.parametrize("model,depths,hw", CASES)
def test_forward_backward_match_oracle(model, depths, hw):
    eng = hostsim_engine()
    g = graphs.build_tiny(model, (hw, hw))
    sd = weights.synthetic_state_dict(g, 3)
    hooks = [g.hooks[d] for d in depths]
    N = 3
    net = eng.build_net(g, sd, hooks, N)
    onet = restate.OracleNet(g, sd, hooks, dtype=torch.float64)
    torch.manual_seed(hw + len(depths))
    x = torch.randn(N, 3, hw, hw)
    feats = onet.forward(x.double())
    net.forward(x)
    for nd in net.graph.nodes:                       # every activation the truncated graph produces
        got, want = net.read_tensor(nd.dst, N).double(), onet.tensor(nd.dst)
        assert torch.allclose(got, want, rtol=1e-4, atol=max(1e-5, 1e-6 * float(want.abs().max()))), nd   # fp32 vs f64
    hg = [torch.randn_like(f) for f in feats]        # well-conditioned random hook gradients
    write_hook_grads(net, feats, hg, N)
    gx = torch.empty(N, 3, hw, hw)
    net.backward(gx)
    ref = onet.backward(hg)
    err = (gx.double() - ref).abs().max() / ref.abs().max()
    assert err < 1e-4, err
    gx2 = gx.clone()
    net.backward(gx2, accumulate=True)
    assert torch.allclose(gx2, 2 * gx, rtol=1e-6, atol=1e-9)


MODE = {"i2v": attacks.ImageGuidedFMDirection_Adam, "std": attacks.ImageGuidedStd_Adam}


@pytest.mark.parametrize("name", ["i2v_resnet_d2_f32", "i2v_vgg_d2_f64", "i2v_alexnet_d3_f64",
                                  "i2v_squeezenet_d2_f64", "std_resnet_d2_f64"])
def test_attack_loop_against_golden(name):
    fx = gu.load(name)
    atk = MODE[fx["kind"]](fx["models"], depth=fx["depth"], step_size=fx["lr"], steps=fx["steps"],
                           engine=hostsim_engine(), graph_builder=graphs.build_tiny, weight_seed=fx["wseed"])
    vid = gu.videos_of(fx)
    adv = atk(vid, torch.zeros(fx["b"], dtype=torch.long), ["clip0"])
    ref_cost = np.array([float(s) for s in fx["cost_str"]])
    np.testing.assert_allclose(atk.last_costs, ref_cost, rtol=2e-4)
    assert adv.shape == vid.shape
    assert list(atk.loss_info["clip0"].keys()) == list(range(fx["steps"]))
    assert atk.loss_info["clip0"][0]["cost"] == str(np.float32(atk.last_costs[0]))
    # L_inf / box invariants of the composed output
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    un = adv * std + mean
    clean = torch.from_numpy(fx["clip_u8"]).float() / 255
    assert (un - clean).abs().max() <= 16 / 255 + 1e-6
    assert un.min() >= -1e-6 and un.max() <= 1 + 1e-6
    assert np.abs(adv.numpy() - fx["adv"]).mean() < 5e-3


def test_ens_and_aens_against_golden():
    fx = gu.load("ens_4models_f64")
    atk = attacks.ImageGuidedFML2_Adam_MultiModels(fx["models"], depths=fx["depth"], steps=fx["steps"],
                                                   engine=hostsim_engine(), graph_builder=graphs.build_tiny)
    adv = atk(gu.videos_of(fx), torch.zeros(fx["b"], dtype=torch.long), ["clip0", "clip1"])
    ref_cost = np.array([float(s) for s in fx["cost_str"]])
    np.testing.assert_allclose(atk.last_costs, ref_cost, rtol=2e-4)
    assert np.abs(adv.numpy() - fx["adv"]).mean() < 5e-3

    for name in ("aens_2x2_f64", "aens_coefce_f64"):
        fx = gu.load(name)
        atk = attacks.AENS_I2V_MF(fx["models"], depths=fx["depth"], step_size=fx["lr"], steps=fx["steps"],
                                  engine=hostsim_engine(), graph_builder=graphs.build_tiny, **fx["kw"])
        adv, used_time, cost_saved = atk(gu.videos_of(fx), torch.zeros(fx["b"], dtype=torch.long), ["c"] * fx["b"])
        np.testing.assert_allclose(cost_saved, fx["cost_saved"], rtol=2e-4)
        np.testing.assert_allclose(np.stack(atk.weights), fx["weights"], rtol=1e-4)
        np.testing.assert_allclose(atk.coeffs.numpy(), fx["coeffs_after"], rtol=1e-4)
        assert cost_saved.dtype == np.float64 and used_time >= 0


@pytest.mark.parametrize("name", ["tf_i2v_resnet_d3_f64", "tf_ens_f64", "tf_aens_f64"])
def test_teacher_forced_steps_against_reference_states(name):
    """The planner/executor + host-simulated kernels, restarted at every step from the reference's optimiser state
    (gu.check_teacher_forced states the tolerances; the same check runs on the HIP kernels in test_gpu_parity.py)."""
    fx = gu.load(name)
    gu.check_teacher_forced(fx, gu.make_attack(fx, attacks, engine=hostsim_engine()))


def test_aens_coefficients_persist_across_calls():
    """`self.coeffs` lives on the attack object and carries over to the next call (TPAMI_attack.py:165,265): two calls
    of the product class against the oracle run twice with the coefficients handed on (the oracle itself is pinned to
    the reference for this in tests/test_oracle_vs_reference.py)."""
    fx = dict(models=["resnet", "squeezenet"], depth={"resnet": [2, 3], "squeezenet": [2, 3]}, hw=64, wseed=0)
    vid = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, 2, 64, 64), generator=torch.Generator().manual_seed(41),
                                                 dtype=torch.uint8).numpy()})
    atk = attacks.AENS_I2V_MF(fx["models"], depths=fx["depth"], step_size=0.02, momentum=1.0, steps=4,
                              engine=hostsim_engine(), graph_builder=graphs.build_tiny)
    nets = [restate.OracleNet(g, sd, h, dtype=torch.float64) for g, sd, h in gu.hook_lists(fx)]
    coeffs = torch.ones(4, dtype=torch.float64)
    first = []
    for call in range(2):
        atk(vid.clone(), torch.zeros(1, dtype=torch.long), ["v"])
        first.append(np.stack(atk.weights)[0].copy())
        o = restate.run_attack(nets, vid.double(), steps=4, step_size=0.02, mode="aens", coeffs=coeffs, momentum=1.0)
        coeffs = o["coeffs"]
        np.testing.assert_allclose(np.stack(atk.weights), np.stack(o["weights"]), rtol=1e-3)     # fp32 engine vs f64 oracle, lr 0.02
        np.testing.assert_allclose(atk.coeffs.numpy(), coeffs.float().numpy(), rtol=1e-3)
    assert np.abs(first[0] - first[1]).max() > 0     # the second call did not start from ones (uniform 0.25)


def test_clip_lanes_are_bit_identical():
    """Frames are independent in I2V / ENS-I2V, so cutting the batch into concurrently executed lanes of whole clips
    (own nets, own threads / streams) must not change a single bit of the perturbed clips; the batch cost is the sum of
    the lanes' costs."""
    gen = torch.Generator().manual_seed(8)
    vid = gu.videos_of({"clip_u8": torch.randint(0, 256, (3, 3, 2, 64, 64), generator=gen, dtype=torch.uint8).numpy()})
    names = ["a", "b", "c"]
    for cls, kw in ((attacks.ImageGuidedFMDirection_Adam, dict(model_name_lists=["resnet"], depth=2, step_size=0.005, steps=3)),
                    (attacks.ImageGuidedFML2_Adam_MultiModels, dict(model_name_lists=["resnet", "alexnet"],
                                                                    depths={"resnet": 2, "alexnet": 3}, steps=2))):
        one = cls(engine=hostsim_engine(), graph_builder=graphs.build_tiny, **kw)
        assert one._lane_count(3) == 1                      # host simulation: lanes are opt-in
        ref = one(vid, torch.zeros(3, dtype=torch.long), names)
        two = cls(engine=hostsim_engine(), graph_builder=graphs.build_tiny, **kw)
        two.clip_lanes = 2
        got = two(vid, torch.zeros(3, dtype=torch.long), names)
        assert torch.equal(got, ref) and torch.equal(two._delta, one._delta)
        assert np.array_equal(two.last_costs, one.last_costs)      # canonical per-clip summation: the split does not show in the log either
        assert list(two.loss_info) == names and two.loss_info["c"][0]["cost"] == str(two.last_costs[0])
        assert len(two._lanes) == 2 and two._lanes[0]._nets[0].max_frames == 2 and two._lanes[1]._nets[0].max_frames == 4
        again = two(vid, torch.zeros(3, dtype=torch.long), names)      # lanes and their nets are reused
        assert torch.equal(again, ref)
    # a single clip is cut along its frames
    long_clip = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, 16, 32, 32), generator=gen, dtype=torch.uint8).numpy()})
    one = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=2, step_size=0.005, steps=2, engine=hostsim_engine(), graph_builder=graphs.build_tiny)
    ref = one(long_clip, torch.zeros(1, dtype=torch.long), ["v"])
    two = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=2, step_size=0.005, steps=2, engine=hostsim_engine(), graph_builder=graphs.build_tiny)
    two.clip_lanes = 2
    assert two._lane_count(1, 16) == 2 and two._lane_count(1, 8) == 1
    got = two(long_clip, torch.zeros(1, dtype=torch.long), ["v"])
    assert got.shape == ref.shape and torch.equal(got, ref) and torch.equal(two._delta, one._delta)
    assert np.array_equal(two.last_costs, one.last_costs)      # canonical per-clip summation: the split does not show in the log either
    dr = attacks.ImageGuidedStd_Adam(["resnet"], depth=2, step_size=0.005, steps=1, engine=hostsim_engine(), graph_builder=graphs.build_tiny)
    dr.clip_lanes = 2
    assert dr._lane_count(3) == 1                           # DR couples the whole batch: never split


def test_teacher_forced_first_step_matches_reference_gradient():
    """First step from delta_0: the sign of the gradient handed to Adam against the reference's
    (f64-backbone fixture => the reference value is accurate)."""
    fx = gu.load("i2v_resnet_d3_f64")
    atk = attacks.ImageGuidedFMDirection_Adam(fx["models"], depth=fx["depth"], step_size=fx["lr"], steps=1,
                                              engine=hostsim_engine(), graph_builder=graphs.build_tiny)
    atk(gu.videos_of(fx), torch.zeros(1, dtype=torch.long), ["c"])
    r0 = fx["grad0"]
    d1 = atk._delta.numpy()
    big = np.abs(r0) > 2e-2 * np.abs(r0).max()
    moved = np.sign(0.01 / 255 - d1)                 # first Adam step moves by -lr*sign(g)
    assert (moved[big] == np.sign(r0[big])).mean() > 0.995
    # |g| is ~1e-7 on many pixels, comparable to Adam's eps=1e-8 and to fp32 gradient noise, so the
    # step g/(|g|+1e-8) is only reproducible where the gradient is well above that floor
    well = np.abs(r0) > 5e-2 * np.abs(r0).max()      # |g| >= ~1e-7 = 10x Adam's eps
    assert (np.abs(d1 - fx["delta_first"])[well] < 1e-4).all()
    assert (np.abs(d1 - fx["delta_first"])[well] < 2e-5).mean() > 0.99
    assert (np.abs(d1 - fx["delta_first"]) < 1e-4).mean() > 0.9


def test_error_paths():
    eng = hostsim_engine()
    with pytest.raises(AttributeError):
        attacks.ImageGuidedFMDirection_Adam(["densenet"], depth=2, step_size=0.005, engine=eng)
    with pytest.raises(UnboundLocalError):
        attacks.ImageGuidedFMDirection_Adam(["inception"], depth=2, step_size=0.005, engine=eng)
    g = graphs.build_tiny("resnet", (32, 32))
    sd = weights.synthetic_state_dict(g, 0)
    net = eng.build_net(g, sd, [g.hooks[2]], 2)
    from i2v_amd.lib import I2VError
    with pytest.raises(I2VError):
        net.forward(torch.zeros(3, 3, 32, 32))          # more frames than planned
    net.close()
    with pytest.raises(I2VError):
        eng.capi.i2v_net_forward  # noqa: B018
        from i2v_amd import lib
        lib.check(eng.capi, eng.capi.i2v_net_workspace_bytes(eng.h, 10 ** 6) or 1)   # bad id -> error text


def test_clip_from_u8_matches_loader_tail():
    """ClipToTensor + Normalize + THWC->CTHW (the tail of /root/reference/datasets.py:88-93)."""
    eng = hostsim_engine()
    u8 = torch.randint(0, 256, (2, 3, 5, 6, 3), generator=torch.Generator().manual_seed(4), dtype=torch.uint8)
    got = eng.clip_from_u8(u8)
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    ref = (u8.permute(0, 4, 1, 2, 3).float() / 255 - mean) / std
    assert torch.equal(got, ref)


def test_resize_crop_normalise_matches_loader_transform():
    """`Resize(256, bilinear) -> CenterCrop(224) -> ClipToTensor -> Normalize` (datasets.py:86-93) fused into
    one kernel (host simulation here, HIP in test_gpu_parity.py) against the oracle's independent restatement of the
    gluoncv / OpenCV arithmetic -- bit for bit (the resize is integer fixed point) -- on landscape, portrait, already-sized
    and square frames; and, as the only pin available offline (cv2 and gluoncv are not installed: parity unpinned), against
    PIL's bilinear resize on an UPscale, where PIL's filter has the same two-tap support: within 1.5/255."""
    eng = hostsim_engine()
    for H, W in ((240, 320), (360, 300), (256, 341), (288, 288), (224, 224)):
        fr = torch.randint(0, 256, (2, 2, H, W, 3), generator=torch.Generator().manual_seed(H + W), dtype=torch.uint8)
        got = eng.clip_resize_crop(fr)
        assert got.shape == (2, 3, 2, 224, 224)
        assert torch.equal(got, restate.resize_center_crop_normalise(fr.numpy()))
    from PIL import Image
    yy, xx = np.mgrid[0:240, 0:320]
    img = np.stack([(yy + xx) % 256, (2 * yy) % 256, (3 * xx) % 256], -1).astype(np.uint8)
    mine = restate.resize_center_crop_normalise(img[None, None])[0, :, 0].permute(1, 2, 0)
    mine = (mine * torch.tensor(restate.STD) + torch.tensor(restate.MEAN)).numpy() * 255
    pil = np.asarray(Image.fromarray(img).resize((341, 256), Image.BILINEAR)).astype(np.float32)
    x1 = int(round((341 - 224) / 2.0))
    wrap = np.abs(np.diff(img.astype(np.int32), axis=0, prepend=0)).max(-1) > 8         # the test pattern's modulo jumps
    ok = ~np.asarray(Image.fromarray((wrap * 255).astype(np.uint8)).resize((341, 256), Image.BILINEAR))[16:240, x1:x1 + 224].astype(bool)
    assert np.abs(pil[16:240, x1:x1 + 224] - mine)[ok].max() <= 1.5


def test_image_main_takes_decoded_frames(tmp_path, monkeypatch):
    """`--clip_dir` with `{label}-raw.npy` decoded uint8 clips: the CLI runs the loader's transform on the engine."""
    from i2v_amd import attacks
    eng = hostsim_engine()
    monkeypatch.setitem(attacks._ENGINES, attacks.default_device(), eng)
    monkeypatch.setattr(graphs, "build", graphs.build_tiny)
    monkeypatch.setenv("I2V_OPT_PATH", str(tmp_path / "out"))
    raw = tmp_path / "raw"
    raw.mkdir()
    for label in (4, 9):
        np.save(raw / f"{label}-raw.npy", np.random.RandomState(label).randint(0, 256, (2, 80, 100, 3)).astype(np.uint8))
    import importlib
    import image_main
    importlib.reload(image_main)
    monkeypatch.setattr(eng, "clip_resize_crop", lambda fr, short_side=256, crop=224, _f=eng.clip_resize_crop: _f(fr, short_side=72, crop=crop))
    image_main.main(["--attack_method", "ImageGuidedFMDirection_Adam", "--step", "1", "--depth", "2", "--direction_image_model", "resnet",
                     "--clip_dir", str(raw), "--hw", "64", "--batch_size", "2", "--file_prefix", "raw"])
    out = tmp_path / "out" / "Image-ImageGuidedFMDirection_Adam-1-raw"
    assert sorted(f for f in os.listdir(out) if f.endswith(".npy")) == ["4-adv.npy", "9-adv.npy"]
    assert np.load(out / "4-adv.npy").shape == (3, 2, 64, 64)


def test_net_ids_are_reused():
    """A long CLI run re-plans whenever its batch grows; the ids it gives back must be handed out again (round 1 capped
    a handle at 4096 backbones ever created)."""
    eng = hostsim_engine()
    seen = set()
    for _ in range(5000):
        nid = ctypes.c_int(-1)
        assert eng.capi.i2v_net_create(eng.h, ctypes.byref(nid)) == 0
        seen.add(nid.value)
        assert eng.capi.i2v_net_destroy(eng.h, nid.value) == 0
    assert len(seen) <= 2
    # and a planned net still works next to the churn
    g = graphs.build_tiny("resnet", (32, 32))
    net = eng.build_net(g, weights.synthetic_state_dict(g, 1), [g.hooks[1]], 2)
    net.forward(torch.zeros(2, 3, 32, 32))
    net.close()


def test_pool_output_plane_is_validated():
    """ADVICE (round 1): the planner accepted any destination plane for a max-pool; a wrong one is an out-of-bounds write
    in the banded kernels.  floor and ceil_mode planes are accepted, anything else is refused."""
    from i2v_amd import lib
    eng = hostsim_engine()
    cap = eng.capi

    def pool_into(h_out):
        nid, b0, b1, t0, t1 = (ctypes.c_int(-1) for _ in range(5))
        assert cap.i2v_net_create(eng.h, ctypes.byref(nid)) == 0
        assert cap.i2v_net_add_buffer(eng.h, nid.value, 4, 14, 14, ctypes.byref(b0)) == 0
        assert cap.i2v_net_add_buffer(eng.h, nid.value, 4, h_out, h_out, ctypes.byref(b1)) == 0
        assert cap.i2v_net_add_tensor(eng.h, nid.value, b0.value, 0, 4, 0, ctypes.byref(t0)) == 0
        assert cap.i2v_net_add_tensor(eng.h, nid.value, b1.value, 0, 4, 0, ctypes.byref(t1)) == 0
        rc = cap.i2v_net_add_maxpool(eng.h, nid.value, ctypes.byref(lib.PoolDesc(t0.value, t1.value, 3, 2, 0)))
        cap.i2v_net_destroy(eng.h, nid.value)
        return rc
    assert pool_into(6) == 0            # floor((14 - 3) / 2) + 1
    assert pool_into(7) == 0            # ceil_mode
    assert pool_into(8) != 0 and b"pool output plane" in cap.i2v_last_error()
    assert pool_into(5) != 0


def test_device_guard_is_per_engine():
    """the pointer guard checks the OWNING engine's device -- a host-simulation (cpu) engine that exists, or once
    existed, in the process must not let CPU tensors through to a HIP engine."""
    import types
    from i2v_amd import engine as _engine, lib as _lib
    eng = hostsim_engine()
    t = torch.zeros(4)
    assert _engine._ptr(t, eng).value == t.data_ptr()
    hip_like = types.SimpleNamespace(device=torch.device("cuda:0"))
    with pytest.raises(_lib.I2VError, match="passed to an engine"):
        _engine._ptr(t, hip_like)
    assert _engine._DEVICE_TYPES.get("cpu", 0) >= 1
    assert isinstance(eng.plan_lock, type(__import__("threading").RLock()))


@pytest.mark.parametrize("model", ["resnet", "vgg"])
def test_mid_trajectory_teacher_forced_step_hostsim(model):
    """The full-size GPU check (`golden_util.check_mid_trajectory_step`) on the tiny backbones through the host simulation:
    3 free steps, then one engine iteration vs one float64 oracle iteration from the same optimiser state."""
    hw = 48
    g = graphs.build_tiny(model, (hw, hw))
    onet = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[2]], dtype=torch.float64)
    u8 = torch.randint(0, 256, (2, 3, 4, hw, hw), generator=torch.Generator().manual_seed(77), dtype=torch.uint8)
    vid = gu.videos_of({"clip_u8": u8.numpy()})
    eng = hostsim_engine()
    mk = lambda steps: attacks.ImageGuidedFMDirection_Adam([model], depth=2, step_size=0.005, steps=steps, weight_seed=0,   # noqa: E731
                                                           engine=eng, graph_builder=graphs.build_tiny)
    o32 = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[2]], dtype=torch.float32)
    gu.check_mid_trajectory_step(mk, [onet], vid, [1, 6], t=3, lr=0.005, tag=f"hostsim tiny {model}",
                                 fp32_nets=[o32] if model == "vgg" else None)     # both assertion modes


def test_segment_timing_reports_the_same_totals_as_per_launch_timing():
    """`i2v_timing_enable(h, 2)` (round 4): one event pair per SEGMENT -- a run of consecutive launches of one kind in a launch list --
    instead of one per launch (what `bench.py`'s timed region uses: the per-launch records cost 1.7 % of the figure they measured).  The
    per-kind launch counts, algorithmic flops and algorithmic bytes must be exactly those of the per-launch mode; only the low-intensity
    split needs per-launch records.  Fused pairs (forced) count as one launch carrying both halves' flops in either mode."""
    from i2v_amd import attacks
    eng = hostsim_engine()
    vid = torch.randn(1, 3, 2, 32, 32, generator=torch.Generator().manual_seed(5))
    totals = {}
    for mode in (True, "segments"):
        atk = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=3, step_size=0.005, steps=2, engine=eng, graph_builder=graphs.build_tiny,
                                                  weight_seed=0)
        atk(vid, torch.zeros(1, dtype=torch.long), ["w"])            # plan outside the timed call
        eng.timing_enable(mode)
        atk(vid, torch.zeros(1, dtype=torch.long), ["a"])
        kt = eng.timing_collect()
        eng.timing_enable(False)
        totals[mode] = {k: (int(v["launches"]), v["flops"], v["bytes"]) for k, v in kt.items()}
        if mode == "segments":
            assert all(v["lowi_launches"] == 0 for v in kt.values())
    assert totals[True] == totals["segments"], totals
    assert totals[True]["conv_igemm_fwd"][0] > 0 and totals[True]["conv_igemm_dgrad"][0] > 0


def test_fusable_pairs_of_resnet50_are_found_and_change_nothing(monkeypatch):
    """Round 4, planner side of the fused 3x3 -> pointwise pair (`mark_fusable`): on ResNet-50 -> layer3 the forward list holds six pairs
    (conv2 -> conv3 of layer1's three and layer2's three stride-1 bottlenecks; the first bottleneck's shortcut convolution, planned
    between conv2 and conv3, is swapped out of the way) and the backward list six (the input gradients of conv2 -> conv1).  A pair is
    admitted only if nothing else touches the intermediate: a hook on a conv2 output removes that pair.  Forcing the fusion executes
    the pairs through `k_conv_fused` (on the host: the two convolutions one after the other) -- features and gradient unchanged."""
    eng = hostsim_engine()
    g = graphs.build("resnet50", (32, 32))
    sd = weights.synthetic_state_dict(g, 0)
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    outs = []
    for force in ("0", "1"):
        monkeypatch.setenv("I2V_FORCE_FUSE", force)
        net = eng.build_net(g, sd, [g.hooks[3]], 2)
        info = net.fusion_info()
        assert info[:2] == (6, 6) and info[2:] == ((6, 6) if force == "1" else (0, 0)), info
        net.forward(x)
        f = net.save_hook(0, 2).clone()
        write_hook_grads(net, [f], [torch.randn(f.shape, generator=torch.Generator().manual_seed(1))], 2)
        gx = torch.empty(2, 3, 32, 32)
        net.backward(gx)
        outs.append((f, gx.clone()))
        net.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert bool(torch.isfinite(outs[1][0]).all()) and bool(torch.isfinite(outs[1][1]).all())   # the host backend POISONS the unstored intermediate
    # a hook on layer1.0.conv2's output: that tensor now has another reader
    monkeypatch.setenv("I2V_FORCE_FUSE", "1")
    t_conv2 = next(i for i, t in enumerate(g.tensors) if (t.name or "").endswith("layer1.0.conv2"))
    # read-back of a fused pair's intermediate (activation or gradient view) is refused by name; every other tensor still reads
    net = eng.build_net(g, sd, [g.hooks[3]], 2)
    net.forward(x)
    t_conv1 = next(i for i, t in enumerate(g.tensors) if (t.name or "").endswith("layer1.0.conv1"))
    for tid, grad in ((t_conv2, False), (t_conv1, True)):       # forward pair conv2 -> conv3; backward pair dgrad(conv2) -> dgrad(conv1)
        with pytest.raises(RuntimeError, match="never stored"):
            net.read_tensor(tid, 2, grad=grad)
    t_conv3 = next(i for i, t in enumerate(g.tensors) if (t.name or "").endswith("layer1.0.out"))
    assert bool(torch.isfinite(net.read_tensor(t_conv3, 2)).all())
    net.close()
    net = eng.build_net(g, sd, [g.hooks[3], t_conv2], 2)
    assert net.fusion_info()[0] == 5
    net.close()


def test_fast_pathway_blocks_are_found_and_change_nothing(monkeypatch):
    """Round 6, planner side of the fused fast-pathway block (`mark_fusable` -> `k_fastblock`): in SlowFast's fast res2 stage every
    bottleneck's forward launches (conv1 -> conv2 -> [projection] -> conv3) form a group, and so do the input gradients of conv3 and
    conv2 in the backward list (a first block's projection gradient, planned between them, is moved in front).  Forcing the groups
    executes them through `k_fastblock` (on the host: the member convolutions one after the other, intermediates POISONED afterwards)
    -- hooked features and input gradient unchanged; an intermediate cannot be read back; a hook on one removes its group."""
    eng = hostsim_engine()
    mt = "slowfast_resnet50"
    g = graphs.build_video_tiny(mt, (8, 32, 32))
    sd = weights.synthetic_state_dict(g, 0)
    hooks = graphs.video_hooks(g, mt)
    T = g.tensors[g.input].T
    x = torch.randn(2 * T, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    outs, ran = [], []
    for force in ("0", "1"):
        monkeypatch.setenv("I2V_FORCE_FASTBLOCK", force)
        net = eng.build_net(g, sd, hooks, 2 * T)
        before = eng.capi.i2v_backend_stat(b"fastblock_launches")
        net.forward(x)
        feats = [net.save_hook(i, 2 * hi.T).clone() for i, hi in enumerate(net.hooks)]
        write_hook_grads(net, feats, [torch.randn(f.shape, generator=torch.Generator().manual_seed(7 + i)) for i, f in enumerate(feats)], 2 * T)
        gx = torch.empty(2 * T, 3, 32, 32)
        net.backward(gx)
        ran.append(eng.capi.i2v_backend_stat(b"fastblock_launches") - before)
        outs.append((feats, gx.clone()))
        if force == "1":
            names = {(t.name or ""): i for i, t in enumerate(g.tensors)}
            for nm, grad in (("fast_res2.1.conv1", False), ("fast_res2.1.conv2", False), ("fast_res2.0.downsample", False), ("fast_res2.1.conv2", True)):
                with pytest.raises(RuntimeError, match="never stored"):
                    net.read_tensor(names[nm], 2 * g.tensors[names[nm]].T, grad=grad)
            assert bool(torch.isfinite(net.read_tensor(names["fast_res2.1.out"], 2 * g.tensors[names["fast_res2.1.out"]].T)).all())
        net.close()
    assert ran == [0, 4], ran                 # two blocks: forward (one with, one without the projection) and backward
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][1], outs[1][1]) and bool(torch.isfinite(outs[1][1]).all()) and float(outs[1][1].abs().max()) > 0
    # a hook on block 1's conv2 output: that tensor has another reader now -- its forward group is gone, the other three stay
    monkeypatch.setenv("I2V_FORCE_FASTBLOCK", "1")
    names = {(t.name or ""): i for i, t in enumerate(g.tensors)}
    net = eng.build_net(g, sd, hooks + [names["fast_res2.1.conv2"]], 2 * T)
    before = eng.capi.i2v_backend_stat(b"fastblock_launches")
    net.forward(x)
    assert eng.capi.i2v_backend_stat(b"fastblock_launches") - before == 1
    net.close()


@pytest.mark.parametrize("model,depths,hw", [("resnet", [3], 64), ("resnet", [2, 3], 64), ("resnet", [4], 64), ("squeezenet", [2, 3], 64),
                                             ("densenet121", [2, 4], 64), ("vgg", [3], 32)])
def test_launch_overlap_keeps_every_bit(model, depths, hw, monkeypatch):
    """Round 6, `mark_overlap` / `run_list`: launches that do not depend on their predecessors (a first
    bottleneck's projection shortcut, its input gradient) are ISSUED early, on the net's side stream.  The host simulation executes
    every launch synchronously in issue order, so a hoisted launch really runs before the launches it was moved over: the dependency
    analysis is right iff every activation and the input gradient stay bit-identical to the in-order run."""
    eng = hostsim_engine()
    g = graphs.build_tiny(model, (hw, hw))
    sd = weights.synthetic_state_dict(g, 3)
    hooks = [g.hooks[d] for d in depths]
    N = 2
    torch.manual_seed(7)
    x = torch.randn(N, 3, hw, hw)
    hgs = None
    out = {}
    for mode in ("0", "1000"):
        monkeypatch.setenv("I2V_OVERLAP_MAX_FRAMES", mode)
        net = eng.build_net(g, sd, hooks, N)
        before = eng.capi.i2v_backend_stat(b"overlap_launches")
        net.forward(x)
        acts = [net.read_tensor(nd.dst, N).clone() for nd in net.graph.nodes]
        feats = [net.read_tensor(h, N) for h in hooks]
        if hgs is None:
            hgs = [torch.randn_like(f) for f in feats]
        write_hook_grads(net, feats, hgs, N)
        gx = torch.empty(N, 3, hw, hw)
        net.backward(gx)
        gx2 = gx.clone()
        net.backward(gx2, accumulate=True)
        out[mode] = (acts, gx, gx2, eng.capi.i2v_backend_stat(b"overlap_launches") - before)
    a0, g0, h0, n0 = out["0"]
    a1, g1, h1, n1 = out["1000"]
    assert n0 == 0
    if model == "resnet":
        assert n1 >= 2 * len([1 for _ in range(max(depths))]), n1      # every projection shortcut, both passes (+ the extra backward call)
    for u, v in zip(a0, a1):
        assert torch.equal(u, v)
    assert torch.equal(g0, g1) and torch.equal(h0, h1)
