"""TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the reference's attack-loop arithmetic.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`i2v_amd`) never does and fails loudly without its HIP library.

Parity status: the LOOP arithmetic (frame flattening, delta_0, compose/clamp semantics, cosine
loss, cost definition, Adam state evolution, AENS re-weighting, output layout) is PINNED by
golden vectors captured from the imported reference classes (`oracle/make_golden.py`,
`tests/golden/*.npz`, re-checked live by `tests/test_oracle_vs_reference.py` whenever
`/root/reference` is present).  The BACKBONE arithmetic (torchvision 0.10.1 graphs, ATen
conv/BN/pool of pytorch 1.9.1) is third-party to the reference, which holds no tests or golden
vectors for it: that part is pinned only against the locally installed torch 2.10 CPU ops
through the same golden vectors.

Everything here is written WITHOUT autograd: truncated forward to the hook layers, analytic
cosine gradient, input-gradient-only backward (ReLU masks, pool arg-max, residual fan-out),
inclusive clamp masks, and Adam spelled out op by op.
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)     # /root/reference/image_attacks.py:33-34
STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5
COS_EPS = 1e-8                   # F.cosine_similarity default, image_attacks.py:343


def _cvec(v, like):
    return torch.tensor(v, dtype=like.dtype, device=like.device).view(1, 3, 1, 1)


# ---------------------------------------------------------------------------
# backbone: forward to the hooks / backward to the input, interpreting the graph IR
# ---------------------------------------------------------------------------
class OracleNet:
    """Truncated backbone.  Forward follows the reference's call `self.model(x)` with BN
    frozen in eval mode (`image_attacks.py:253-256,318,334`); backward is what
    `cost.backward()` (`:352`) delivers to the input, minus the wasted weight gradients."""

    def __init__(self, graph, state_dict, hook_tensors: Sequence[int], dtype=torch.float32, device="cpu"):
        """`device`: "cpu" for everything the parity ladder uses; `bench.py`'s `gpu_framework_baseline` leg runs the same
        restatement with its tensors on cuda:0 (ATen -> MIOpen), as what PyTorch-ROCm would do with the reference's loop."""
        self.g = graph.truncated(list(hook_tensors))
        self.hooks = list(hook_tensors)
        self.device = torch.device(device)
        self.sd = {k: v.to(dtype).to(self.device) for k, v in state_dict.items()}
        self.dtype = dtype
        self.buf: Dict[int, torch.Tensor] = {}
        self.pool_idx: Dict[int, torch.Tensor] = {}

    def _view(self, store, tid, N=None):
        t = self.g.tensors[tid]
        if t.buf not in store:
            store[t.buf] = torch.zeros(N, self.g.buffers[t.buf], t.H, t.W, dtype=self.dtype, device=self.device)
        return store[t.buf][:, t.c_off:t.c_off + t.C]

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        g, N = self.g, x.shape[0]
        self.buf, self.pool_idx = {}, {}
        self._view(self.buf, g.input, N).copy_(x)
        for i, nd in enumerate(g.nodes):
            src = self._view(self.buf, nd.src, N)
            if nd.op == "conv":
                bias = self.sd[nd.bias] if nd.bias else None
                if nd.pre_bn:       # DenseNet pre-activation: norm -> relu -> conv
                    src = F.relu(F.batch_norm(src, self.sd[nd.pre_bn + ".running_mean"], self.sd[nd.pre_bn + ".running_var"],
                                              self.sd[nd.pre_bn + ".weight"], self.sd[nd.pre_bn + ".bias"], False, 0.1, BN_EPS))
                y = F.conv2d(src, self.sd[nd.weight], bias, nd.stride, nd.pad)
                if nd.bn:
                    y = F.batch_norm(y, self.sd[nd.bn + ".running_mean"],
                                     self.sd[nd.bn + ".running_var"], self.sd[nd.bn + ".weight"],
                                     self.sd[nd.bn + ".bias"], False, 0.1, BN_EPS)
                if nd.residual is not None:
                    y = y + self._view(self.buf, nd.residual, N)
                if nd.relu:
                    y = F.relu(y)
            elif nd.op == "avgpool":
                y = F.avg_pool2d(src, nd.k, nd.stride)
            else:
                y, idx = F.max_pool2d(src, nd.k, nd.stride, nd.pad, ceil_mode=nd.ceil_mode,
                                      return_indices=True)
                self.pool_idx[i] = idx
            self._view(self.buf, nd.dst, N).copy_(y)
        return [self._view(self.buf, h, N) for h in self.hooks]

    def tensor(self, tid):
        return self._view(self.buf, tid)

    def grad_of(self, tid):
        return self._view(self.last_grads, tid)

    def adopt_activations(self, acts: Dict[int, torch.Tensor]):
        """Replace the stored activations (tensor id -> value) by externally computed ones and
        recompute the pool arg-max from them, so that a following `backward` uses exactly the same
        ReLU gates / arg-max as the implementation under test (removes the chaotic gate flips of
        near-zero activations from a gradient comparison)."""
        for tid, val in acts.items():
            self._view(self.buf, tid).copy_(val.to(self.dtype))
        for i, nd in enumerate(self.g.nodes):
            if nd.op == "maxpool":
                src = self._view(self.buf, nd.src)
                _, idx = F.max_pool2d(src, nd.k, nd.stride, nd.pad, ceil_mode=nd.ceil_mode, return_indices=True)
                self.pool_idx[i] = idx

    def backward(self, hook_grads: Sequence[torch.Tensor]) -> torch.Tensor:
        """d(cost)/d(input) given d(cost)/d(hook feature) for every hook (same order)."""
        g = self.g
        N = hook_grads[0].shape[0]
        gb: Dict[int, torch.Tensor] = {}
        for h, hg in zip(self.hooks, hook_grads):
            self._view(gb, h, N).add_(hg)
        self._view(gb, g.input, N)
        for i in range(len(g.nodes) - 1, -1, -1):
            nd = g.nodes[i]
            gd = self._view(gb, nd.dst, N)
            src = self._view(self.buf, nd.src, N)
            if nd.op == "conv":
                dz = gd
                if nd.relu:
                    dz = dz * (self._view(self.buf, nd.dst, N) > 0).to(self.dtype)
                if nd.residual is not None:
                    self._view(gb, nd.residual, N).add_(dz)
                if nd.bn:
                    inv = self.sd[nd.bn + ".weight"] / torch.sqrt(self.sd[nd.bn + ".running_var"] + BN_EPS)
                    dz = dz * inv.view(1, -1, 1, 1)
                gx = torch.nn.grad.conv2d_input(src.shape, self.sd[nd.weight], dz.contiguous(),
                                                nd.stride, nd.pad)
                if nd.pre_bn:       # back through relu(BN(x)): gate on the pre-activation, times the BN scale
                    inv = self.sd[nd.pre_bn + ".weight"] / torch.sqrt(self.sd[nd.pre_bn + ".running_var"] + BN_EPS)
                    sh = self.sd[nd.pre_bn + ".bias"] - self.sd[nd.pre_bn + ".running_mean"] * inv
                    pre = src * inv.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
                    gx = gx * (pre > 0).to(self.dtype) * inv.view(1, -1, 1, 1)
                self._view(gb, nd.src, N).add_(gx)
            elif nd.op == "avgpool":
                gx = gd.repeat_interleave(nd.k, 2).repeat_interleave(nd.k, 3) / (nd.k * nd.k)
                full = torch.zeros_like(src)
                full[:, :, :gx.shape[2], :gx.shape[3]] = gx
                self._view(gb, nd.src, N).add_(full)
            else:
                idx = self.pool_idx[i]
                C = src.shape[1]
                gx = torch.zeros(N, C, src.shape[2] * src.shape[3], dtype=self.dtype, device=self.device)
                gx.scatter_add_(2, idx.reshape(N, C, -1), gd.reshape(N, C, -1))
                self._view(gb, nd.src, N).add_(gx.view_as(src))
        self.last_grads = gb          # buffer id -> gradient (tests read intermediate gradients from it)
        return self._view(gb, g.input, N).clone()


# ---------------------------------------------------------------------------
# elementwise pieces of the loop
# ---------------------------------------------------------------------------
def flatten_frames(videos: torch.Tensor) -> torch.Tensor:
    """(b,c,f,h,w) -> (b*f,c,h,w), frame n = b_idx*f + f_idx (`image_attacks.py:300-301`)."""
    b, c, f, h, w = videos.shape
    return videos.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)


def unflatten_frames(x: torch.Tensor, b: int, f: int) -> torch.Tensor:
    """`image_attacks.py:362-363`."""
    n, c, h, w = x.shape
    return x.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


def unnormalise(x: torch.Tensor) -> torch.Tensor:
    """u = x*std + mean with in-place mul_ then add_ (`image_attacks.py:62,308`)."""
    return x.clone().mul_(_cvec(STD, x)).add_(_cvec(MEAN, x))


def compose(u, delta, eps):
    """x = (clamp(u + clamp(delta,-eps,eps),0,1) - mean)/std (`image_attacks.py:331-332`,
    `:59` -- subtraction then DIVISION by std).  Also returns the inclusive pass-through
    mask of the two clamps (ATen clamp_backward: (x >= min) & (x <= max))."""
    dc = torch.clamp(delta, -eps, eps)
    s = u + dc
    xi = torch.clamp(s, 0, 1)
    xn = (xi - _cvec(MEAN, u)) / _cvec(STD, u)
    mask = ((delta >= -eps) & (delta <= eps) & (s >= 0) & (s <= 1)).to(u.dtype)
    return xn, mask


def compose_backward(gx, mask):
    """d/d(delta) of compose: divide by std (div_ backward), gate by both clamps."""
    return gx / _cvec(STD, gx) * mask


def cosine_fwd_bwd(a: torch.Tensor, b: torch.Tensor):
    """Per-frame cosine similarity over the flattened feature and its gradient w.r.t. `a`
    (`image_attacks.py:341-343`; b is detached `:322`).  Returns (cos[N], dcos/da)."""
    N = a.shape[0]
    af, bf = a.reshape(N, -1), b.reshape(N, -1)
    n1 = af.norm(dim=1, keepdim=True).clamp_min(COS_EPS)
    n2 = bf.norm(dim=1, keepdim=True).clamp_min(COS_EPS)
    cos = ((af / n1) * (bf / n2)).sum(1)
    grad = bf / (n1 * n2) - cos.view(N, 1) * af / (n1 * n1)
    return cos, grad.view_as(a)


class AdamState:
    """`torch.optim.Adam([delta], lr)` with defaults betas=(0.9,0.999), eps=1e-8 (`image_attacks.py:306`), single-tensor path
    of torch/optim/adam.py -- `lerp_` for m, `mul_` + `addcmul_` for v, `sqrt / bc2_sqrt + eps`, `addcdiv_`, host-side bias
    corrections in double -- written as SEPARATE single-rounding operations (lerp = one fused multiply-add).  torch's own CPU
    kernels are not a bit-level reference: ATen's vectorised `addcmul` / `addcdiv` contract to FMA on some hosts and not on
    others, an ulp of difference; this formulation is host-independent and is what the HIP kernel and its scalar restatement
    compute bit for bit."""

    def __init__(self, like, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.m = torch.zeros_like(like)
        self.v = torch.zeros_like(like)
        self.t = 0
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps

    def step(self, delta, grad):
        self.t += 1
        w1, w2 = 1 - self.b1, 1 - self.b2
        if delta.dtype == torch.float32:        # fma(w1, g - m, m): the product of two floats is exact in double
            w1f = float(np.float32(w1))
            self.m = (self.m.double() + (grad - self.m).double() * w1f).float()
        else:
            self.m = self.m + (grad - self.m) * w1
        self.v = self.v * self.b2 + (grad * w2) * grad
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        step_size = self.lr / bc1
        denom = self.v.sqrt() / (bc2 ** 0.5) + self.eps
        delta.add_((self.m / denom) * (-step_size))


def sign_step_bim(adv_norm, u, grad, step_size, eps):
    """BIM-style update (`/root/reference/base_attacks.py:289-293`): un-normalise, sign step,
    project delta to +-eps, clamp to [0,1], re-normalise.  5-D (b,3,f,h,w) tensors."""
    mean = torch.tensor(MEAN, dtype=u.dtype).view(3, 1, 1, 1)
    std = torch.tensor(STD, dtype=u.dtype).view(3, 1, 1, 1)
    a = adv_norm.clone().mul_(std).add_(mean)
    a = a + step_size * grad.sign()
    d = torch.clamp(a - u, -eps, eps)
    a = torch.clamp(u + d, 0, 1)
    return a.sub_(mean).div_(std)


def sign_step_ilaf(delta, grad, step_size):
    """ILAF update `modifier.data -= step_size * grad.sign()` (`image_attacks.py:617`)."""
    return delta - step_size * grad.sign()


def aens_coeffs(prev, coeffs, momentum):
    """`coeffs = softmax(softmax(prev) + momentum*coeffs)` (`TPAMI_attack.py:265`)."""
    return torch.softmax(torch.softmax(prev, 0) + momentum * coeffs, 0)


# ---------------------------------------------------------------------------
# the attack loops
# ---------------------------------------------------------------------------
def run_attack(nets: Sequence[OracleNet], videos: torch.Tensor, *, steps: int, step_size: float,
               epsilon: float = 16 / 255, mode: str = "i2v", coeffs: Optional[torch.Tensor] = None,
               momentum: float = 0.0, coef_CE: bool = False, trace: bool = False,
               forced_states: Optional[Sequence] = None, forced_coeffs: Optional[Sequence[torch.Tensor]] = None,
               first_step: int = 0):
    """Restates `ImageGuidedFMDirection_Adam.forward` (`image_attacks.py:294-364`, one net, one
    hook), `ImageGuidedFML2_Adam_MultiModels.forward` (`:426-496`, several nets) [mode 'i2v'],
    `AENS_I2V_MF.forward` (`TPAMI_attack.py:223-320`) [mode 'aens'] and
    `ImageGuidedStd_Adam.forward` (`image_attacks.py:187-234`) [mode 'std'].

    Teacher forcing: `forced_states[i] = (delta, exp_avg, exp_avg_sq)`, when given and not None, replaces the
    optimiser state at the START of step i (i.e. the reference's state after step i-1; the Adam step count is i
    either way), and `forced_coeffs[i]` replaces the AENS coefficients used in step i (the self-computed ones are
    still returned in `weights_own`).  `first_step` skips the iterations before it (only meaningful together with a
    forced state for that step: a mid-trajectory step at full size without paying for the steps that led there); the
    trace lists then hold the executed steps only.
    Returns a dict: adv (b,3,f,h,w), costs[steps] (float32), and with trace=True per-step
    delta (after the update), per-step gradients, cos per (step, layer, frame), weights.
    """
    dt = nets[0].dtype
    b, c, f, h, w = videos.shape
    x = flatten_frames(videos.to(dt)).contiguous()
    N = b * f
    delta = torch.full((N, c, h, w), 0.01 / 255, dtype=dt)          # image_attacks.py:304
    opt = AdamState(delta, step_size)
    u = unnormalise(x)                                              # :308
    init = None
    if mode != "std":
        init = [[t.clone() for t in net.forward(x)] for net in nets]   # :318-323 (raw x, not norm(u))
    L = sum(len(net.hooks) for net in nets)
    out = {"costs": np.zeros(steps, np.float32), "deltas": [], "cos": [], "weights": [], "weights_own": [], "grad0": None,
           "grads": []}
    prev = torch.ones(L, dtype=dt) if mode == "aens" else None
    for i in range(first_step, steps):
        if forced_states is not None and forced_states[i] is not None:
            fd, fm, fv = forced_states[i]
            delta, opt.m, opt.v, opt.t = fd.clone().to(dt), fm.clone().to(dt), fv.clone().to(dt), i
        if mode == "aens":
            coeffs = aens_coeffs(prev, coeffs, momentum)            # TPAMI_attack.py:265
            out["weights_own"].append(coeffs.clone().numpy())
            if forced_coeffs is not None and forced_coeffs[i] is not None:
                coeffs = forced_coeffs[i].clone().to(dt)
            out["weights"].append(coeffs.clone().numpy())
        xn, mask = compose(u, delta, epsilon)
        gx = torch.zeros_like(xn)
        cos_all = []
        l = 0
        cost = torch.zeros((), dtype=dt)
        for n, net in enumerate(nets):
            feats = net.forward(xn)
            hgrads = []
            for k, a in enumerate(feats):
                if mode == "std":
                    # cost += a.std() over the WHOLE tensor, unbiased (image_attacks.py:218)
                    cnt = a.numel()
                    mu = a.mean()
                    sd = a.std()
                    cost = cost + sd
                    hgrads.append((a - mu) / ((cnt - 1) * sd))
                    cos_all.append(torch.full((N,), float(sd), dtype=dt))
                    continue
                cs, gr = cosine_fwd_bwd(a, init[n][k])
                cos_all.append(cs)
                if mode == "aens":
                    hgrads.append(gr * (coeffs[l] / L))             # mean over L of coeff*sum_frames
                else:
                    hgrads.append(gr)                               # cost = sum_l sum_f cos
                l += 1
            gx += net.backward(hgrads)
        cosm = torch.stack(cos_all)                                 # (L, N)
        if mode == "aens":
            each = (coeffs.view(-1, 1) * cosm).sum(1)               # TPAMI_attack.py:289-291
            cost = each.mean()
            prev = each.clone() if coef_CE else cosm.sum(1)         # :293-297
        elif mode == "i2v":
            cost = cosm.sum()                                       # image_attacks.py:347
        out["costs"][i] = float(cost)
        g = compose_backward(gx, mask)
        if i == first_step and trace:
            out["grad0"] = g.clone()
        if trace:
            out["grads"].append(g.clone())
        opt.step(delta, g)                                          # :351-353
        if trace:
            out["deltas"].append(delta.clone())
            out["cos"].append(cosm.clone())
    xn, _ = compose(u, delta, epsilon)                              # :360-361
    out["adv"] = unflatten_frames(xn, b, f)
    out["delta"] = delta                                            # the unconstrained Adam variable after the last step
    out["coeffs"] = coeffs
    return out


def run_ilaf(model, hook_modules, videos, ori_videos, *, steps, step_size=0.005, eps=16 / 255, modifier0=None):
    """Restatement of `ILAF.forward` (`/root/reference/image_attacks.py:534-629`) for a torch video model whose
    hooked modules are `hook_modules`: whole-tensor norms, `-(0.5*|d|/|d0| + <d0/|d0|, d/|d|>)` summed over the
    hooked layers, gradient w.r.t. the perturbation through the clamped compose, `modifier -= step*sign(grad)`.
    Returns (output in the reference's scrambled layout, costs[steps], first-step gradient, final modifier).
    `modifier0` (b,c,f,h,w), test infrastructure only: start the loop from this perturbation instead of `videos - ori_videos`
    (a teacher-forced step from another implementation's state; the initial direction still comes from `videos`)."""
    feats = []
    handles = [m.register_forward_hook(lambda mod, i, o: feats.append(o)) for m in hook_modules]
    dtype = videos.dtype
    mean = torch.tensor(MEAN, dtype=dtype).view(1, 3, 1, 1, 1)
    std = torch.tensor(STD, dtype=dtype).view(1, 3, 1, 1, 1)
    b, c, f, h, w = videos.shape
    try:
        with torch.no_grad():
            feats.clear(); model(ori_videos); ori_f = list(feats)
            feats.clear(); model(videos); adv_f = list(feats)
        d0 = [a - o for a, o in zip(adv_f, ori_f)]
        n0 = [torch.norm(d, p=2) for d in d0]
        dir0 = [d / n for d, n in zip(d0, n0)]
        ori_u = ori_videos.clone().mul_(std).add_(mean)
        modifier = videos.clone().mul_(std).add_(mean) - ori_u                 # `torch.Tensor(t)` aliases t: dtype kept (:574-575)
        if modifier0 is not None:
            modifier = modifier0.to(dtype).clone()
        costs, grad0 = [], None
        for _ in range(steps):
            modifier.requires_grad_(True)
            x = (torch.clamp(ori_u + torch.clamp(modifier, -eps, eps), 0, 1) - mean) / std
            feats.clear(); model(x)
            loss = 0
            for k, a in enumerate(feats):
                d = a - ori_f[k]
                nrm = torch.norm(d, p=2)
                loss = loss - (0.5 * nrm / n0[k] + (dir0[k] * (d / nrm)).sum())
            g = torch.autograd.grad(loss, modifier)[0]
            grad0 = g.detach().clone() if grad0 is None else grad0
            modifier = sign_step_ilaf(modifier.detach(), g, step_size)
            costs.append(float(loss.detach()))
        out = (torch.clamp(ori_u + torch.clamp(modifier, -eps, eps), 0, 1) - mean) / std
        return out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4), np.array(costs), grad0, modifier
    finally:
        for hd in handles:
            hd.remove()


def cost_strings(costs: np.ndarray) -> List[str]:
    """`str(cost.detach().cpu().numpy())` of a float32 scalar (`image_attacks.py:358`)."""
    return [str(np.float32(c)) for c in costs]


# ---------------------------------------------------------------------------
# input pipeline
# ---------------------------------------------------------------------------
def resize_center_crop_normalise(frames_u8: np.ndarray, short_side=256, crop=224) -> torch.Tensor:
    """CPU twin of `i2v_clip_resize_crop_u8_f32`: the reference's validation transform (`/root/reference/datasets.py:86-93`)
    `video_transforms.Resize(short_side, 'bilinear')` -> `CenterCrop(crop)` -> `ClipToTensor` (/255, THWC -> CTHW) ->
    `Normalize(mean, std)` on decoded uint8 frames (b, t, H, W, 3).

    The arithmetic lives in gluoncv 0.10.4 (`gluoncv/torch/data/video_transforms`: numpy frames go through `cv2.resize(img,
    (new_w, new_h), interpolation=cv2.INTER_LINEAR)`) and OpenCV (`resize.cpp`: 8-bit bilinear in fixed point: horizontal pass
    with 11-bit weights into int, vertical pass `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`), neither vendored nor
    installed: restated here from the public sources -- PARITY UNPINNED.  Written independently of `i2v_amd.clips` (its own
    coordinate / weight construction, whole-frame resize then crop) so that the two restatements check each other."""
    b, t, H, W, _ = frames_u8.shape
    # gluoncv get_resize_sizes
    if (W <= H and W == short_side) or (H <= W and H == short_side):
        rh, rw = H, W
    elif W < H:
        rh, rw = int(short_side * H / W), short_side
    else:
        rh, rw = short_side, int(short_side * W / H)

    def axis(n_out, n_in):
        idx, w0, w1 = np.zeros(n_out, np.int64), np.zeros(n_out, np.int64), np.zeros(n_out, np.int64)
        scale = float(n_in) / float(n_out)
        for d in range(n_out):
            fx = np.float32((d + 0.5) * scale - 0.5)
            sx = int(math.floor(fx))
            fx = np.float32(fx - np.float32(sx))
            if sx < 0:
                sx, fx = 0, np.float32(0)
            if sx >= n_in - 1:
                sx, fx = n_in - 1, np.float32(0)
            idx[d] = sx
            w0[d] = int(np.rint(np.float32(np.float32(1) - fx) * np.float32(2048)))
            w1[d] = int(np.rint(fx * np.float32(2048)))
        return idx, w0, w1
    xi, xa0, xa1 = axis(rw, W)
    yi, yb0, yb1 = axis(rh, H)
    src = frames_u8.astype(np.int64)
    xn = np.minimum(xi + 1, W - 1)
    S = src[:, :, :, xi, :] * xa0[None, None, None, :, None] + src[:, :, :, xn, :] * xa1[None, None, None, :, None]   # (b,t,H,rw,3)
    yn = np.minimum(yi + 1, H - 1)
    S0, S1 = S[:, :, yi], S[:, :, yn]
    d = (((yb0[None, None, :, None, None] * (S0 >> 4)) >> 16) + ((yb1[None, None, :, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    y1, x1 = int(round((rh - crop) / 2.0)), int(round((rw - crop) / 2.0))
    d = d[:, :, y1:y1 + crop, x1:x1 + crop]
    v = torch.from_numpy(d.astype(np.float32)) / 255                                   # ClipToTensor
    v = v.permute(0, 4, 1, 2, 3)
    mean = torch.tensor(MEAN).view(1, 3, 1, 1, 1)
    std = torch.tensor(STD).view(1, 3, 1, 1, 1)
    return ((v - mean) / std).contiguous()                                              # Normalize: sub then div


def ucf101_transform(frames_u8: np.ndarray, size=224, crop=224) -> torch.Tensor:
    """CPU twin of `i2v_clip_resample_crop_u8_f32`: the UCF-101 loader's validation transform
    (`/root/reference/dataset_ucf101.py:113-126`: `Scale(224)` -> `CornerCrop(224, 'c')` -> `ToTensor()` -> `Normalize`, then
    `torch.stack(clip, 0).permute(1, 0, 2, 3)`, :79) on decoded uint8 frames (b, t, H, W, 3).  The resampling arithmetic is
    Pillow's, and Pillow is installed: the frames go through `PIL.Image.resize(..., BILINEAR)` ITSELF, so this oracle is pinned by
    construction (tests/test_pil_resample.py also runs the reference's own transform classes where /root/reference exists)."""
    from PIL import Image
    b, t, H, W, _ = frames_u8.shape
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    out = torch.empty(b, 3, t, crop, crop)
    for bi in range(b):
        for ti in range(t):
            img = Image.fromarray(frames_u8[bi, ti])
            w, h = img.size
            if not ((w <= h and w == size) or (h <= w and h == size)):              # transforms_ucf101.py:278-289
                img = img.resize((size, int(size * h / w)) if w < h else (int(size * w / h), size), Image.BILINEAR)
            iw, ih = img.size
            x1, y1 = int(round((iw - crop) / 2.)), int(round((ih - crop) / 2.))      # :343-348
            img = img.crop((x1, y1, x1 + crop, y1 + crop))
            ten = torch.from_numpy(np.array(img)).permute(2, 0, 1).float().div(255)  # ToTensor, :200-213
            out[bi, :, ti] = ten.sub(mean).div(std)                                  # Normalize: t.sub_(m).div_(s)
    return out
