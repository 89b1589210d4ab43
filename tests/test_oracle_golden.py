"""CPU: the oracle (oracle/restate.py) against the golden vectors captured from the imported
reference classes (oracle/make_golden.py).  f64 fixtures pin the semantics tightly; the f32
fixture checks the chaotic-regime contract of SURVEY.md 7.3-1 (cost trajectory, statistics)."""
import numpy as np
import pytest
import torch

from oracle import restate
from tests import golden_util as gu

F64_CASES = ["i2v_resnet_d3_f64", "i2v_vgg_d2_f64", "i2v_alexnet_d3_f64", "i2v_squeezenet_d2_f64",
             "std_resnet_d2_f64", "ens_4models_f64", "aens_2x2_f64", "aens_coefce_f64"]
MODE = {"i2v": "i2v", "ens": "i2v", "std": "std", "aens": "aens"}


def run_oracle(fx, dtype):
    nets = [restate.OracleNet(g, sd, hooks, dtype=dtype) for g, sd, hooks in gu.hook_lists(fx)]
    kw = {}
    if fx["kind"] == "aens":
        kw = dict(coeffs=torch.ones(2 * len(nets), dtype=dtype), momentum=fx["kw"]["momentum"],
                  coef_CE=fx["kw"]["coef_CE"])
    return restate.run_attack(nets, gu.videos_of(fx, dtype), steps=fx["steps"], step_size=fx["lr"],
                              mode=MODE[fx["kind"]], trace=True, **kw)


@pytest.mark.parametrize("name", F64_CASES)
def test_oracle_matches_reference_semantics_f64(name):
    fx = gu.load(name)
    out = run_oracle(fx, torch.float64)
    ref_cost = np.array([float(s) for s in fx["cost_str"]])
    np.testing.assert_allclose(out["costs"], ref_cost, rtol=2e-6)
    g0 = out["grad0"].float().numpy()
    assert np.abs(g0 - fx["grad0"]).max() <= 2e-6 * np.abs(fx["grad0"]).max()
    # the reference keeps delta/Adam in float32 (image_attacks.py:304); the oracle run is all-f64,
    # so agreement is to float32 rounding of the state, not to f64 epsilon
    assert np.abs(out["deltas"][0].float().numpy() - fx["delta_first"]).max() < 2e-6
    assert np.abs(out["deltas"][-1].float().numpy() - fx["delta_last"]).max() < 5e-6
    assert np.abs(out["adv"].float().numpy() - fx["adv"]).max() < 5e-5
    if fx["kind"] == "aens":
        np.testing.assert_allclose(np.stack(out["weights"]), fx["weights"], rtol=1e-5)
        np.testing.assert_allclose(out["coeffs"].float().numpy(), fx["coeffs_after"], rtol=1e-5)
        np.testing.assert_allclose(out["costs"], fx["cost_saved"], rtol=2e-6)


@pytest.mark.parametrize("name", ["tf_i2v_resnet_d3_f64", "tf_ens_f64", "tf_aens_f64"])
def test_oracle_teacher_forced_steps_f64(name):
    """Every step restarted from the reference's (delta, exp_avg, exp_avg_sq) -- and, for the adaptive attack, its
    coefficients: the oracle's step must land on the reference's next state (image_attacks.py:325-358,
    TPAMI_attack.py:258-312).  tf_aens_f64 hooks SqueezeNet with LIST depths, i.e. the whole Fire module."""
    fx = gu.load(name)
    nets = [restate.OracleNet(g, sd, hooks, dtype=torch.float64) for g, sd, hooks in gu.hook_lists(fx)]
    S = fx["steps"]
    states = [None] + [tuple(torch.from_numpy(fx[k][i]) for k in ("tf_delta", "tf_m", "tf_v")) for i in range(S - 1)]
    kw = {}
    if fx["kind"] == "aens":
        L = sum(len(n.hooks) for n in nets)
        kw = dict(coeffs=torch.ones(L, dtype=torch.float64), momentum=fx["kw"]["momentum"], coef_CE=fx["kw"]["coef_CE"],
                  forced_coeffs=[torch.from_numpy(w) for w in fx["weights"]])
    out = restate.run_attack(nets, gu.videos_of(fx, torch.float64), steps=S, step_size=fx["lr"], mode=MODE[fx["kind"]],
                             trace=True, forced_states=states, **kw)
    np.testing.assert_allclose(out["costs"], np.array([float(s) for s in fx["cost_str"]]), rtol=2e-6)
    for i in range(S):
        g, r = out["grads"][i].float().numpy(), fx["tf_grad"][i]
        assert np.abs(g - r).max() <= 2e-6 * np.abs(r).max(), i
        # the oracle keeps delta/Adam in float64 here, the reference in float32: agreement to float32 rounding
        assert np.abs(out["deltas"][i].float().numpy() - fx["tf_delta"][i]).max() < 2e-6, i
    if fx["kind"] == "aens":       # the coefficients the oracle derives itself (from the previous forced step's cosines)
        np.testing.assert_allclose(np.stack(out["weights_own"]), fx["weights"], rtol=1e-5)


def test_oracle_f32_contract():
    """fp32 vs fp32 reference: only summation order differs, yet single pixels diverge
    (SURVEY.md 0.5); what must hold is the parity ladder of 7.3-1."""
    fx = gu.load("i2v_resnet_d2_f32")
    out = run_oracle(fx, torch.float32)
    ref_cost = np.array([float(s) for s in fx["cost_str"]])
    np.testing.assert_allclose(out["costs"], ref_cost, rtol=1e-4)
    g0, r0 = out["grad0"].numpy(), fx["grad0"]
    assert np.abs(g0 - r0).max() <= 1e-2 * np.abs(r0).max()
    big = np.abs(r0) > 1e-2 * np.abs(r0).max()
    assert (np.sign(g0[big]) == np.sign(r0[big])).mean() > 0.999
    # first Adam step: every pixel moves by +-lr; only near-zero gradients may flip
    d1 = out["deltas"][0].numpy()
    assert (np.abs(d1 - fx["delta_first"]) < 1e-4).mean() > 0.98
    # statistics of the final perturbation
    dl, rl = out["deltas"][-1].numpy(), fx["delta_last"]
    assert abs(np.abs(dl).mean() / np.abs(rl).mean() - 1) < 0.01
    adv = out["adv"].numpy()
    assert np.abs(adv - fx["adv"]).mean() < 5e-3


def test_cost_string_format():
    fx = gu.load("i2v_resnet_d2_f32")
    out = run_oracle(fx, torch.float32)
    s = restate.cost_strings(out["costs"])
    assert all(isinstance(x, str) and float(x) == np.float32(float(x)) for x in s)
    assert len(s) == fx["steps"]


def test_cosine_matches_torch():
    torch.manual_seed(0)
    a, b = torch.randn(5, 3, 7, 7), torch.randn(5, 3, 7, 7)
    a.requires_grad_(True)
    ref = torch.nn.functional.cosine_similarity(a.view(5, -1), b.view(5, -1))
    ref.sum().backward()
    cos, gr = restate.cosine_fwd_bwd(a.detach(), b)
    assert torch.allclose(cos, ref.detach(), atol=1e-6)
    assert torch.allclose(gr, a.grad, atol=1e-6)


def test_adam_matches_torch_optim_to_the_ulp():
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.full((4, 3, 5, 5), 0.01 / 255))
    opt = torch.optim.Adam([p], lr=0.005)
    d = p.detach().clone()
    st = restate.AdamState(d, 0.005)
    for _ in range(5):
        g = torch.randn_like(d) * 1e-4
        p.grad = g.clone()
        opt.step()
        st.step(d, g)
        # torch's CPU kernels contract `addcmul` / `addcdiv` to FMA on some hosts, the oracle never does (it is the
        # host-independent formulation the HIP kernel computes bit for bit): agreement to a few ulp
        assert (d - p.detach()).abs().max() <= 8e-9          # |delta| <= 0.02: an ulp is <= 1.9e-9; the states drift apart by ulps
        assert (d == p.detach()).float().mean() > 0.3


def test_sign_step_golden():
    fx = gu.load("sign_step")
    u8 = torch.from_numpy(fx["clip_u8"])
    vid = gu.videos_of(fx)
    mean = torch.tensor(gu.MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(gu.STD).view(1, 3, 1, 1, 1)
    u = vid.clone().mul_(std).add_(mean)
    eps, steps = float(fx["eps"]), int(fx["steps"])
    adv = vid.clone()
    for g in torch.from_numpy(fx["BIM_grads"]):
        adv = restate.sign_step_bim(adv[0], u[0], g[0], eps / steps, eps)[None]
    assert torch.equal(adv, torch.from_numpy(fx["BIM_adv"]))
    assert u8.shape[2] == 32


def test_frame_order_and_output_layout():
    v = torch.arange(2 * 3 * 4 * 2 * 2, dtype=torch.float32).view(2, 3, 4, 2, 2)
    x = restate.flatten_frames(v)
    assert torch.equal(x[1 * 4 + 2], v[1, :, 2])          # n = b_idx*f + f_idx
    assert torch.equal(restate.unflatten_frames(x, 2, 4), v)
