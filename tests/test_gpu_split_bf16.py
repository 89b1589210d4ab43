"""GPU (-m gpu): the opt-in `I2V_MATH=bf16x3` mode (round 5) -- convolutions on three-term bf16 operands (six bf16 MFMAs per 16 K rows,
fp32 accumulation) instead of fp32-input MFMAs.  It is NOT the default and is never what `bench.py`'s `value` measures; these tests pin
what it is: fp32-grade arithmetic held to the SAME tolerances as the fp32 path, deterministic, and independent of the tile choice.

  * every activation and the input gradient of ResNet-50 -> layer3 at 224^2 against the oracle, at the fp32 path's own tolerances, and
    against the FLOAT64 oracle next to the fp32 path's error: the mode's mean error must not exceed twice the fp32-MFMA path's
    (measured: about the same -- every product term down to 2^-26 |w||x| is kept, the accumulation is fp32 in both);
  * the headline attack (configs[0], 10 steps) against the committed float64 yardstick: cost of every step within rtol 2e-4, perturbed
    pixels within the margins `bench.py`'s parity_check uses;
  * the four tile configurations (and two chunks per barrier) forced in turn: bit-identical results;
  * the launches really ran on the split-bf16 K loop (`i2v_backend_stat("bf3_launches")`).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from i2v_amd import attacks, graphs, weights  # noqa: E402
from oracle import restate, size_parity  # noqa: E402
from tests import golden_util as gu  # noqa: E402
from tests.test_gpu_parity import write_hook_grads  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    e = attacks.get_engine("cuda:0")
    assert e.capi.i2v_backend() == b"hip:gfx950"
    return e


@pytest.fixture(autouse=True)
def _needs_the_experimental_library(experimental_build):
    """Every test of this module plans nets with I2V_MATH=bf16x3, which the product library refuses (`i2v_net_plan`)."""


def dev(t):
    return t.to("cuda:0").contiguous()


def _run_layers(eng, g, sd, hooks, x, hg):
    N = x.shape[0]
    net = eng.build_net(g, sd, hooks, N)
    before = eng.capi.i2v_backend_stat(b"bf3_launches")
    net.forward(dev(x))
    acts = {nd.dst: net.read_tensor(nd.dst, N).cpu() for nd in net.graph.nodes}
    feats = [acts[h] for h in hooks]
    write_hook_grads(net, feats, hg, N)
    gx = torch.empty(N, 3, x.shape[2], x.shape[3], device="cuda:0")
    net.backward(gx)
    ran = eng.capi.i2v_backend_stat(b"bf3_launches") - before
    net.close()
    return acts, gx.cpu(), ran


def test_split_bf16_full_size_layers_against_both_oracles(eng, monkeypatch):
    g = graphs.build("resnet50", (224, 224))
    sd = weights.synthetic_state_dict(g, 0)
    hooks = [g.hooks[3]]
    N = 2
    x = gu.videos_of({"clip_u8": torch.randint(0, 256, (1, 3, N, 224, 224), generator=torch.Generator().manual_seed(1), dtype=torch.uint8).numpy()})
    x = restate.flatten_frames(x).contiguous()
    o64 = restate.OracleNet(g, sd, hooks, dtype=torch.float64)
    o64.forward(x.double())
    hg = [torch.randn(o64.tensor(h).shape, generator=torch.Generator().manual_seed(2)) for h in hooks]
    monkeypatch.delenv("I2V_MATH", raising=False)
    a32, g32, ran32 = _run_layers(eng, g, sd, hooks, x, hg)
    monkeypatch.setenv("I2V_MATH", "bf16x3")
    a3, g3, ran3 = _run_layers(eng, g, sd, hooks, x, hg)
    assert ran32 == 0 and ran3 > 60, (ran32, ran3)          # ResNet-50 -> layer3: 40 forward + 39 input-gradient launches, all but the stem pair
    worst = 0.0
    for nd in g.truncated(hooks).nodes:
        ref = o64.tensor(nd.dst)
        e3 = float((a3[nd.dst].double() - ref).abs().max() / ref.abs().max())
        e32 = float((a32[nd.dst].double() - ref).abs().max() / ref.abs().max())
        m3 = float((a3[nd.dst].double() - ref).abs().mean()); m32 = float((a32[nd.dst].double() - ref).abs().mean())
        worst = max(worst, e3)
        assert e3 <= 2e-4, (nd, e3)                                   # the fp32 path's own tolerance (test_full_size_layers)
        assert m3 <= 2.0 * m32 + 1e-9, (nd, m3, m32, e3, e32)         # no worse than twice the fp32-MFMA path's distance from float64
    # input gradient: both modes gated by their own activations, against the float64 oracle gated by the SAME activations
    errs = {}
    for tag, acts, gx in (("fp32", a32, g32), ("bf16x3", a3, g3)):
        o64.adopt_activations({k: v.double() for k, v in acts.items()})
        ref = o64.backward([h.double() for h in hg])
        errs[tag] = (float((gx.double() - ref).abs().max() / ref.abs().max()), float((gx.double() - ref).abs().mean() / ref.abs().mean()))
    print(f"\\nsplit-bf16 vs float64 oracle: worst activation error {worst:.2e} of max; input gradient (max/max, mean/mean): {errs}")
    assert errs["bf16x3"][0] <= 2e-4 and errs["bf16x3"][1] <= 2.0 * errs["fp32"][1] + 1e-9, errs


def test_split_bf16_tiles_are_bit_identical_and_deterministic(eng, monkeypatch):
    monkeypatch.setenv("I2V_MATH", "bf16x3")
    monkeypatch.setenv("I2V_AUTOTUNE", "0")
    g = graphs.Graph("bf3_test", (28, 28))
    x = g.new_tensor(3, 28, 28, False, "input")
    g.input = x
    a = g.conv(x, 64, 3, 1, 1, "a.weight", bn="a_bn", relu=True)              # K = 27: per-row gather, stays on the fp32 path
    c = g.conv(a, 96, 1, 1, 0, "c.weight", bn="c_bn", relu=True)              # pointwise K = 64, 96 rows (partial tiles)
    d = g.conv(c, 64, 3, 1, 1, "d.weight", bn="d_bn", relu=True)              # 3x3, K = 864 (54 chunks)
    e = g.conv(d, 64, 1, 1, 0, "e.weight", bn="e_bn", relu=True, residual=a)  # residual epilogue
    f = g.conv(e, 128, 3, 2, 1, "f.weight", bn="f_bn", relu=True)             # stride 2: parity-class input gradients (scalar epilogue)
    h = g.conv(f, 160, 1, 1, 0, "h.weight", bn="h_bn", relu=True)             # K = 128, 160 rows
    g.hooks[1] = h
    sd = weights.synthetic_state_dict(g, 0)
    frames = 5
    xin = dev(torch.randn(frames, 3, 28, 28, generator=torch.Generator().manual_seed(0)))
    outs = []
    for cfg in (3, 3, 2, 1, 0, 0 | 64, 3 | 64):          # (0 | 64: the software-pipelined loop of the 128x128 tile)
        monkeypatch.setenv("I2V_FORCE_CFG", str(cfg))
        net = eng.build_net(g, sd, [h], frames)
        before = eng.capi.i2v_backend_stat(b"bf3_launches")
        net.forward(xin)
        assert eng.capi.i2v_backend_stat(b"bf3_launches") - before == 5
        ft = net.save_hook(0, frames).cpu()
        hgrad = torch.randn(ft.shape, generator=torch.Generator().manual_seed(1))
        write_hook_grads(net, [ft], [hgrad], frames)
        gx = torch.empty(frames, 3, 28, 28, device="cuda:0")
        net.backward(gx)
        outs.append((ft, gx.cpu()))
        net.close()
    for ft, gx in outs[1:]:
        assert torch.equal(ft, outs[0][0]) and torch.equal(gx, outs[0][1])
    # ... and close to the fp32 path (not bit-identical to it: another summation order)
    monkeypatch.delenv("I2V_MATH")
    monkeypatch.setenv("I2V_FORCE_CFG", "3")
    net = eng.build_net(g, sd, [h], frames)
    net.forward(xin)
    f32 = net.save_hook(0, frames).cpu()
    net.close()
    assert not torch.equal(f32, outs[0][0])
    assert float((f32 - outs[0][0]).abs().max()) <= 2e-5 * float(f32.abs().max())


def test_split_bf16_headline_attack_against_the_float64_yardstick(eng, monkeypatch):
    """configs[0] (clip seed 1000, ResNet-50 layer3, 10 steps) in the bf16x3 mode against the committed float64 run: the same bounds
    `bench.py`'s parity_check puts on the default path, with the fp32 CPU oracle's own measured distance from the float64 run (0.0202
    mean |adv - adv64|, 0.8805 of the pixels within 2 lr: DESIGN.md section 6) as the yardstick."""
    monkeypatch.setenv("I2V_MATH", "bf16x3")
    yard = size_parity.load_yardstick(os.path.join(HERE, "golden"))
    assert yard is not None
    vid = size_parity.synthetic_clip(1000)
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10, weight_seed=0)
    before = eng.capi.i2v_backend_stat(b"bf3_launches")
    adv = atk(vid, torch.zeros(1, dtype=torch.long), ["clip0"]).cpu()
    assert eng.capi.i2v_backend_stat(b"bf3_launches") - before > 1000
    st = size_parity.compare_sampled(atk.last_costs, atk._delta.cpu(), adv, yard)
    print("\\nbf16x3 attack vs the float64 oracle:", st, "\\ncosts", atk.last_costs)
    y32 = {"mean_abs_adv_diff": 0.0202, "frac_pixels_within_2lr": 0.8805}
    ok, bad = size_parity.within_bounds(st, st, y32)
    assert ok, bad
    un = adv * size_parity.STD + size_parity.MEAN
    clean = vid * size_parity.STD + size_parity.MEAN
    assert float((un - clean).abs().max()) <= 16 / 255 + 1e-6 and float(un.min()) >= -1e-6 and float(un.max()) <= 1 + 1e-6
    # deterministic: the same call again, bit for bit (a fresh plan: the autotuner's choices do not enter the result)
    atk2 = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10, weight_seed=0)
    assert torch.equal(atk2(vid, torch.zeros(1, dtype=torch.long), ["clip0"]).cpu(), adv)


@pytest.mark.parametrize("mt", ["slowfast_resnet50", "i3d_resnet50"])
def test_split_bf16_video_backbones_ilaf_costs(eng, monkeypatch, mt):
    """The video graphs in the bf16x3 mode (BASELINE.json configs[4] shape: 1 x 32 x 224^2): spatial, pointwise and TEMPORAL tap-uniform
    launches run on the split-bf16 loop (`conv_igemm_bf3<..., VID>`), stems / attention products / fast-pathway layers stay fp32.  Three
    ILAF steps against the oracle's torch-module run: every cost within rtol 2e-4, as in the default mode."""
    from i2v_amd import sign_attacks, video
    from oracle import video_models as vm
    monkeypatch.setenv("I2V_MATH", "bf16x3")
    thw = (32, 224, 224)
    gen = torch.Generator().manual_seed(11)
    ori_u8 = torch.randint(0, 256, (1, 3, *thw), generator=gen, dtype=torch.uint8)
    adv_u8 = (ori_u8.long() + torch.randint(-10, 11, ori_u8.shape, generator=gen)).clamp(0, 255).to(torch.uint8)
    ori, adv = gu.videos_of({"clip_u8": ori_u8.numpy()}), gu.videos_of({"clip_u8": adv_u8.numpy()})
    model = video.VideoModel(mt, thw)
    atk = sign_attacks.ILAF(model, mt, step_size=0.005, steps=3)
    before = eng.capi.i2v_backend_stat(b"bf3_launches")
    atk(adv.clone(), ori.clone(), torch.zeros(1, dtype=torch.long), ["v"])
    ran = eng.capi.i2v_backend_stat(b"bf3_launches") - before
    g = graphs.build_video(mt, thw)
    tm = vm.load_weights(vm.make(mt, False), weights.synthetic_state_dict(g, 0))
    _, costs, _, _ = restate.run_ilaf(tm, vm.hook_modules(tm, mt), adv, ori, steps=3)
    print(f"\nbf16x3 ILAF {mt}: {ran} split-bf16 launches; costs {atk.last_costs} oracle {costs}")
    assert ran > 50
    np.testing.assert_allclose(atk.last_costs, costs, rtol=2e-4)
