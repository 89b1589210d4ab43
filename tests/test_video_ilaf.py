"""CPU: ILAF on video backbones.  (1) The oracle's restatement of the loop against what the UNMODIFIED reference
class produced on the same torch modules (fixtures `ilaf_*.npz`, oracle/make_golden.py).  (2) The native path
(`ILAF(VideoModel)`, whole loop behind the C ABI, host simulation backend here) against the same fixtures."""
import numpy as np
import pytest
import torch

from i2v_amd import graphs, sign_attacks, video, weights
from oracle import restate, video_models as vm
from tests import golden_util as gu
from tests.hostsim_util import hostsim_engine

FIX = ["ilaf_i3d_f64", "ilaf_i3d_f32", "ilaf_slowfast_f64", "ilaf_tpn_f64"]


def load(name):
    z = np.load(gu.GOLDEN + "/" + name + ".npz")
    fx = {k: z[k] for k in z.files}
    fx["model_type"], fx["prec"] = str(fx["model_type"]), str(fx["prec"])
    fx["steps"], fx["b"], fx["wseed"] = int(fx["steps"]), int(fx["b"]), int(fx["wseed"])
    fx["thw"] = tuple(int(v) for v in fx["thw"])
    fx["cost"] = np.array([float(s) for s in fx["cost_str"]])
    return fx


def clips(fx, dtype=torch.float32):
    return gu.videos_of({"clip_u8": fx["adv_u8"]}, dtype), gu.videos_of({"clip_u8": fx["ori_u8"]}, dtype)


@pytest.mark.parametrize("name", FIX)
def test_restatement_matches_reference_class(name):
    fx = load(name)
    dtype = torch.float64 if fx["prec"] == "f64" else torch.float32
    g = graphs.build_video_tiny(fx["model_type"], fx["thw"])
    model = vm.load_weights(vm.make(fx["model_type"], True), weights.synthetic_state_dict(g, fx["wseed"])).to(dtype)
    adv, ori = clips(fx, dtype)
    # the reference hooks in ITS order; the loss is a sum over layers, so the order is immaterial
    out, costs, grad0, _ = restate.run_ilaf(model, vm.hook_modules(model, fx["model_type"]), adv, ori, steps=fx["steps"])
    tol = 1e-9 if fx["prec"] == "f64" else 2e-3
    np.testing.assert_allclose(costs, fx["cost"], rtol=tol)
    if fx["prec"] == "f64":
        assert np.abs(grad0.numpy() - fx["grad0"]).max() <= 1e-6 * np.abs(fx["grad0"]).max()
        assert np.abs(out.numpy() - fx["out"]).max() < 1e-6
    else:
        assert np.abs(out.numpy() - fx["out"]).mean() < 5e-3


@pytest.mark.parametrize("name", FIX)
def test_native_ilaf_against_reference_fixture(name):
    fx = load(name)
    adv, ori = clips(fx)
    model = video.VideoModel(fx["model_type"], fx["thw"], weight_seed=fx["wseed"], tiny=True)
    atk = sign_attacks.ILAF(model, fx["model_type"], step_size=0.005, steps=fx["steps"], engine=hostsim_engine())
    out = atk(adv.clone(), ori.clone(), torch.zeros(fx["b"], dtype=torch.long), ["v"])
    assert out.shape == fx["out"].shape
    rtol = 2e-4 if fx["prec"] == "f64" else 5e-3          # fp32 engine vs f64 / f32 reference trajectories
    np.testing.assert_allclose(atk.last_costs, fx["cost"], rtol=rtol)
    assert atk.loss_info["v"][0]["cost"] == str(np.float32(atk.last_costs[0]))
    assert list(atk.loss_info["v"].keys()) == list(range(fx["steps"]))
    assert np.abs(out.numpy() - fx["out"]).mean() < 5e-3
    # box / L_inf invariants in the reference's (scrambled) output layout: undo it first
    b, c, f, h, w = out.shape
    un = out.permute(0, 2, 1, 3, 4).reshape(b, c, f, h, w) * torch.tensor(gu.STD).view(1, 3, 1, 1, 1) + torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    clean = torch.from_numpy(fx["ori_u8"]).float() / 255
    assert (un - clean).abs().max() <= 16 / 255 + 1e-6 and un.min() >= -1e-6 and un.max() <= 1 + 1e-6


@pytest.mark.parametrize("name", ["ilaf_i3d_f64", "ilaf_i3d_f32"])
def test_native_first_step_follows_reference_gradient_sign(name):
    fx = load(name)
    adv, ori = clips(fx)
    model = video.VideoModel(fx["model_type"], fx["thw"], weight_seed=fx["wseed"], tiny=True)
    atk = sign_attacks.ILAF(model, fx["model_type"], step_size=0.005, steps=1, engine=hostsim_engine())
    atk(adv.clone(), ori.clone(), torch.zeros(1, dtype=torch.long), ["v"])
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    m0 = (adv * std + mean) - (ori * std + mean)                         # (b,3,f,h,w)
    m1 = atk._modifier.reshape(1, fx["thw"][0], 3, *fx["thw"][1:]).permute(0, 2, 1, 3, 4)
    moved = torch.sign(m0 - m1).numpy()                                   # = sign of the gradient the engine saw
    g = fx["grad0"]
    big = np.abs(g) > 2e-2 * np.abs(g).max()
    assert (moved[big] == np.sign(g[big])).mean() > 0.995
    assert (moved[g == 0] == 0).all()                                     # clamp-masked pixels do not move
    assert (moved[:, :, -1] == 0).all()                                   # T=8: the last frame is never read (stride 2 twice)


def test_video_model_errors():
    with pytest.raises(KeyError):
        video.VideoModel("i3d_nl10_resnet50")          # (the reference names i3d_nl5 only, utils.py:9-10)
    assert video.VideoModel("i3d_nl5_resnet50").graph_for((32, 224, 224)).arch == video.VideoModel("i3d_resnet50").graph_for((32, 224, 224)).arch
    m = video.VideoModel("i3d_resnet50", (8, 32, 32), tiny=True)
    assert m.cuda() is m and m.eval() is m


def test_concurrent_clip_streams_match_sequential():
    """`run_concurrent`: worker threads with their own attack objects (own planned nets) give, clip by clip, exactly
    what one attack object gives sequentially; results come back in item order, or through the callback."""
    from i2v_amd.sign_attacks import run_concurrent
    eng = hostsim_engine()
    fx = load("ilaf_i3d_f32")
    adv, ori = clips(fx)
    gen = torch.Generator().manual_seed(3)
    items = []
    for k in range(5):
        noise = 0.02 * torch.randn(adv.shape, generator=gen)
        items.append((adv + noise, ori, torch.zeros(1, dtype=torch.long), [f"clip{k}"]))

    def make():
        return sign_attacks.ILAF(video.VideoModel(fx["model_type"], fx["thw"], weight_seed=fx["wseed"], tiny=True),
                                 fx["model_type"], step_size=0.005, steps=2, engine=eng)
    seq = make()
    want = [seq(*it).clone() for it in items]
    got, workers = run_concurrent(make, items, streams=3, device="cpu")
    assert len(workers) == 3 and len(got) == 5
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    names = sorted(n for a in workers for n in a.loss_info)
    assert names == [f"clip{k}" for k in range(5)]
    seen = {}
    run_concurrent(make, iter(items), streams=2, device="cpu", on_result=lambda i, item, res: seen.__setitem__(i, res))
    assert sorted(seen) == list(range(5)) and all(torch.equal(seen[i], want[i]) for i in range(5))
    with pytest.raises(ZeroDivisionError):
        run_concurrent(make, items, streams=2, device="cpu", on_result=lambda i, item, res: 1 / 0)


@pytest.mark.parametrize("name", ["ilaf_i3d_f32", "ilaf_slowfast_f64"])
def test_independent_clips_in_one_call_match_one_clip_calls(name):
    """K clips batched into ONE launch list with per-clip loss segments (`ILAF.forward_independent`,
    `i2v_ilaf_*_seg_f32`) are K independent calls: every clip's output and every logged cost is BIT-identical to the one-clip
    call (the reference fine-tunes one clip per call, image_fine_tune_attack.py:73-79, and its norms run over that clip only)."""
    eng = hostsim_engine()
    fx = load(name)
    adv, ori = clips(fx)
    gen = torch.Generator().manual_seed(11)
    advs = torch.cat([adv + 0.02 * torch.randn(adv.shape, generator=gen) for _ in range(3)])
    oris = torch.cat([ori + 0.01 * torch.randn(ori.shape, generator=gen) for _ in range(3)])

    def make():
        return sign_attacks.ILAF(video.VideoModel(fx["model_type"], fx["thw"], weight_seed=fx["wseed"], tiny=True),
                                 fx["model_type"], step_size=0.005, steps=3, engine=eng)
    one = make()
    want = [one(advs[k:k + 1].float(), oris[k:k + 1].float(), torch.zeros(1, dtype=torch.long), [f"c{k}"]).clone() for k in range(3)]
    many = make()
    got = many.forward_independent(advs.float(), oris.float(), torch.zeros(3, dtype=torch.long), ["c0", "c1", "c2"])
    for k in range(3):
        assert torch.equal(got[k:k + 1], want[k]), k
        assert many.loss_info[f"c{k}"] == one.loss_info[f"c{k}"]
    # the coupled form (the reference's semantics for a b > 1 call) is a different computation
    coupled = make()(advs.float(), oris.float(), torch.zeros(3, dtype=torch.long), ["c0", "c1", "c2"])
    assert not torch.equal(coupled, got)
    # a clip whose given adversarial equals its original has |adv0 - ori| = 0 at every hook: refused by name, not a NaN clip
    advs2 = advs.clone().float()
    advs2[1] = oris[1].float()
    with pytest.raises(ValueError, match="c1"):
        many.forward_independent(advs2, oris.float(), torch.zeros(3, dtype=torch.long), ["c0", "c1", "c2"])
