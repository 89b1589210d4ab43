"""White-box sign-step attacks (`/root/reference/base_attacks.py:236-340`) and ILAF's update
(`image_attacks.py:498-629`).

BIM / MI-FGSM / FGSM: the UPDATE RULE -- un-normalise, `+ step*sign(g)`, project to +-eps,
clamp to [0,1], re-normalise -- is one fused HIP kernel (`i2v_sign_step_f32`).  The attacked classifier is either a
`VideoModel(..., num_classes=K)` -- then the cross-entropy gradient is native too: 3-D backbone to its last stage,
global-pool / fc / softmax-CE head (`i2v_head_ce_f32`), input gradient -- or, as in the reference, whatever
differentiable torch module the caller passes, whose gradient then comes from PyTorch.

ILAF (a18) has two paths.  Given an `i2v_amd.video.VideoModel` (I3D / SlowFast graph IR) the WHOLE loop runs in
`libi2v_hip.so`: compose, 3-D forward to the hooked stage, ILAF loss + gradient, input gradient, masked sign step.
Given any other torch module (the reference's calling convention; e.g. a gluoncv model the IR does not cover) the
model runs in PyTorch and only the sign step is native.
"""
import numpy as np
import torch
import torch.nn as nn

from .attacks import get_engine
from .video import VideoModel

MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


_ANNOUNCED = set()


def _announce(obj):
    """Every sign-family / ILAF object says, once per (class, path), which of its two calling conventions it runs on: `native` (a
    `VideoModel`: every launch behind the C ABI) or `torch-module` (the reference's convention: the caller's torch module, its
    gradient from PyTorch autograd, only the update rule native) -- so that a torch-module run is never mistaken for the product."""
    key = (type(obj).__name__, obj.path)
    if key not in _ANNOUNCED:
        _ANNOUNCED.add(key)
        print(f"[i2v_amd] {key[0]}: path = {obj.path}" + ("" if obj.path == "native" else
              " (model and its gradient run in PyTorch; pass an i2v_amd.video.VideoModel for the native path)"), flush=True)


def norm_grads(grads, frame_level=True):
    """`/root/reference/utils.py:58-67` (asserts 32 frames like the reference)."""
    assert len(grads.shape) == 5 and grads.shape[2] == 32
    dims = [1, 3, 4] if frame_level else [1, 2, 3, 4]
    return grads / torch.mean(torch.abs(grads), dims, keepdim=True)


class _SignAttack(object):
    """`base_attacks.Attack` (`base_attacks.py:12-234`): model, device, eval/train restore."""

    def __init__(self, name, model, engine=None):
        self.attack = name
        self.model = model
        self.model_name = str(model).split("(")[0]
        self.training = model.training
        self.path = "native" if isinstance(model, VideoModel) else "torch-module"
        _announce(self)
        if isinstance(model, VideoModel):     # native classifier: graph IR + head, everything behind the C ABI (`_grad_native`)
            if model.num_classes is None:
                raise ValueError("a VideoModel used as a classifier needs num_classes (its head)")
            self._engine = engine or get_engine()
            self.device = self._engine.device
            self._net = self._net_key = None
        else:
            self.device = next(model.parameters()).device
        self._targeted = 1
        self._return_type = "float"
        self.mean, self.std = MEAN, STD
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine(str(self.device) if self.device.type == "cuda" else None)
        return self._engine

    def _unnorm(self, videos):
        mean = torch.as_tensor(self.mean, dtype=videos.dtype, device=videos.device)[:, None, None, None]
        std = torch.as_tensor(self.std, dtype=videos.dtype, device=videos.device)[:, None, None, None]
        return videos.clone().detach().mul_(std).add_(mean)

    def _grad_native(self, adv, labels):
        """`autograd.grad(targeted * CrossEntropyLoss()(model(adv), labels), adv)` (base_attacks.py:282-286) without
        autograd: frames -> 3-D backbone to its last stage -> pool / fc / softmax-CE head and its gradient -> input
        gradient, all in libi2v_hip.so.  Returns the backbone's input gradient as it is written, FRAME-major (b*f,3,h,w): the change
        to the clip layout rides along in the fused post-processing pass (`_post_native`, `i2v_grad_post_f32`)."""
        eng, m = self.engine, self.model
        b, c, f, h, w = adv.shape
        N = b * f
        key = (f, h, w)
        if self._net is None or self._net_key != key or self._net.max_frames < N:
            if self._net is not None:
                self._net.close()
            g = m.graph_for((f, h, w))
            self._net = eng.build_net(g, m.state_dict_for(g), m.classifier_hook(g), N, relu_gain=self._relu_gain(g))
            self._head = tuple(t.to(eng.device) if t is not None else None for t in m.head_weights(g))
            self._net_key = key
        net = self._net
        kw = dict(dtype=torch.float32, device=eng.device)
        x, u = torch.empty(N, 3, h, w, **kw), torch.empty(N, 3, h, w, **kw)
        eng.frames_from_video(adv.detach().to(**kw).contiguous(), x, u)
        net.forward(x)
        W, bias = self._head
        logits, loss_each = torch.empty(b, W.shape[0], **kw), torch.empty(b, **kw)
        scratch = torch.empty(eng.capi.i2v_head_scratch_bytes(W.shape[1], b), dtype=torch.uint8, device=eng.device)
        feats = 0 if len(net.hooks) == 1 else list(range(len(net.hooks)))
        net.head_ce(feats, W, bias, labels.to(eng.device).to(torch.int32).contiguous(), N, float(self._targeted), logits, loss_each, scratch)
        gx = torch.empty_like(x)
        net.backward(gx)
        self.last_logits, self._loss_each = logits, loss_each
        return gx

    @property
    def last_loss(self):
        """Mean cross-entropy of the last native gradient call (read on demand: no framework kernel inside the step)."""
        return self._loss_each.mean()

    def _post_native(self, g, state, shape, mode=None, momentum=False):
        """Native path: frame-major gradient -> [normalisation `mode`] -> [+ decay * momentum, momentum updated] -> clip layout, one fused
        pass of the library (no framework kernel between the backward pass and the sign step)."""
        mom = None
        if momentum:
            if "momentum" not in state:
                state["momentum"] = torch.zeros(tuple(shape), dtype=torch.float32, device=self.engine.device)
            mom = state["momentum"]
        return self.engine.grad_post(g, shape, mode, momentum=mom, decay=getattr(self, "decay", 1.0), frame_major=True)

    def _relu_gain(self, graph):
        """{ReLU-output tensor: backward gain} for the native classifier's planned net (SGM); None: plain gradients."""
        return None

    def _grad_step(self, adv, labels):
        """The gradient as the step loops consume it: frame-major (b*f,3,h,w) on the native path (`_pre_native` / `_post_native` turn it
        into the clip layout in their fused pass), the clip layout from autograd on the torch-module path."""
        return self._grad_native(adv, labels) if self.path == "native" else self._grad(adv, labels)

    def _grad(self, adv, labels):
        """d(targeted * CE) / d adv in the clip layout (b,3,f,h,w), whichever path."""
        if isinstance(self.model, VideoModel):
            return self.engine.grad_post(self._grad_native(adv, labels), adv.shape, None, frame_major=True)
        adv.requires_grad = True
        cost = self._targeted * nn.CrossEntropyLoss()(self.model(adv), labels)
        return torch.autograd.grad(cost, adv, retain_graph=False, create_graph=False)[0]

    def __call__(self, *input, **kwargs):
        self.model.eval()
        images = self.forward(*input, **kwargs)
        self.model.train() if self.training else self.model.eval()
        if self._return_type == "int":
            images = (images * 255).type(torch.uint8)
        return images


class FGSM(_SignAttack):
    """`base_attacks.py:236-259` (no eps projection needed: one step of size eps)."""

    def __init__(self, model, steps=None, epsilon=16 / 255, engine=None):
        super().__init__("FGSM", model, engine)
        self.epsilon = epsilon

    def forward(self, videos, labels):
        videos = videos.to(self.device).float().contiguous()
        labels = labels.to(self.device)
        grad = self._grad_step(videos.clone().detach(), labels).contiguous()
        if self.path == "native":
            grad = self._post_native(grad, {}, videos.shape)
        adv = videos.clone().detach()
        u = self._unnorm(videos)
        b, c, f, h, w = videos.shape
        self.engine.sign_step(adv, u, grad, f * h * w, self.epsilon, self.epsilon)
        return adv


class BIM(_SignAttack):
    """`base_attacks.py:261-295`."""

    def __init__(self, model, epsilon=16 / 255, steps=10, engine=None):
        super().__init__("FGSM", model, engine)          # the reference registers BIM under "FGSM" (:267)
        self.epsilon, self.steps = epsilon, steps
        self.step_size = self.epsilon / self.steps

    def _pre(self, grad, state):
        return grad

    def _pre_native(self, g, state, shape):
        """`_pre` on the native path: `g` is the frame-major input gradient; returns the clip-layout gradient the sign step takes."""
        return self._post_native(g, state, shape)

    def _uses_momentum(self):
        return bool(getattr(self, "momentum", False))

    def _gradient(self, adv, labels):
        """The gradient the step starts from; subclasses put their input transforms here (DI, SI)."""
        return self._grad_step(adv, labels)

    def _l1_momentum(self, grad, state):
        """`grad /= |grad|_1; grad += momentum * decay; momentum = grad` (base_attacks.py:394-398 and its copies)."""
        grad = grad / torch.norm(grad, p=1)
        grad = grad + state.get("momentum", torch.zeros_like(grad)) * self.decay
        state["momentum"] = grad
        return grad

    def forward(self, videos, labels):
        videos = videos.to(self.device).float().contiguous()
        labels = labels.to(self.device)
        u = self._unnorm(videos)
        adv = videos.clone().detach()
        b, c, f, h, w = videos.shape
        state = {}
        if self.path == "native" and self._uses_momentum():     # allocated (a fill kernel) before the loop, not inside a step
            state["momentum"] = torch.zeros(tuple(videos.shape), dtype=torch.float32, device=self.engine.device)
        for _ in range(self.steps):
            g = self._gradient(adv, labels)
            grad = self._pre_native(g, state, videos.shape) if self.path == "native" else self._pre(g, state).contiguous()
            adv = adv.detach()
            self.engine.sign_step(adv, u, grad, f * h * w, self.step_size, self.epsilon)   # :289-293
        return adv


class MIFGSM(BIM):
    """`base_attacks.py:297-340`: frame-level mean-abs normalisation + momentum before the step."""

    def __init__(self, model, epsilon=16 / 255, steps=10, decay=1.0, engine=None):
        super().__init__(model, epsilon, steps, engine)
        self.attack = "MIFGSM"
        self.decay = decay

    def _pre(self, grad, state):
        grad = norm_grads(grad, True)
        grad = grad + state.get("momentum", torch.zeros_like(grad)) * self.decay
        state["momentum"] = grad
        return grad

    def _pre_native(self, g, state, shape):
        assert len(shape) == 5 and shape[2] == 32                        # norm_grads' own assertion (utils.py:59)
        return self._post_native(g, state, shape, "frame", momentum=True)

    def _uses_momentum(self):
        return True


def _nearest_index(n_out: int, n_in: int):
    """Source index of every output index of `F.interpolate(mode='nearest')` (ATen `nearest_idx`: identity when the sizes agree,
    else `min(int(floorf(dst * scale)), n_in - 1)` with `scale = float32(n_in) / n_out`)."""
    if n_out == n_in:
        return np.arange(n_out, dtype=np.int64)
    scale = np.float32(n_in) / np.float32(n_out)
    return np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64), n_in - 1)


def diversity_maps(n_in: int, rnd: int, pad_lo: int, canvas: int = 250, n_out: int = 224):
    """One axis of DI-FGSM's input diversity (`base_attacks.py:364-376`) as ONE index map: nearest resize n_in -> rnd, zero pad to
    `canvas` with `pad_lo` in front, nearest resize canvas -> n_out.  Returns (map[n_out] with -1 for padding, lo[n_in], hi[n_in]):
    output positions [lo[s], hi[s]) read source position s (contiguous: both resizes are monotone)."""
    first = _nearest_index(rnd, n_in)                  # rescaled[i] = src[first[i]]
    second = _nearest_index(n_out, canvas)             # out[y] = padded[second[y]]
    inside = (second >= pad_lo) & (second < pad_lo + rnd)
    m = np.where(inside, first[np.clip(second - pad_lo, 0, rnd - 1)], -1).astype(np.int32)
    lo, hi = np.zeros(n_in, np.int32), np.zeros(n_in, np.int32)
    for src in range(n_in):
        hit = np.nonzero(m == src)[0]
        if hit.size:
            lo[src], hi[src] = hit[0], hit[-1] + 1
            assert hit.size == hi[src] - lo[src]
    return m, lo, hi


class DIFGSM(BIM):
    """Diverse Inputs (`base_attacks.py:342-409`): with probability 1/2 per step the model sees the clip resized (nearest) to a
    random rnd in [224, 250), zero-padded to 250 at a random offset and resized back to 224 -- one composed index map per axis,
    applied (`i2v_resample_nearest_f32`) and transposed for the gradient (`i2v_resample_nearest_bwd_f32`) on the device.  The
    draws consume Python's and torch's global generators exactly as the reference does (`random.random()`, three
    `torch.randint(..., size=(1, 1))`), so a seeded run sees the same transforms."""

    def __init__(self, model, epsilon=16 / 255, steps=10, decay=1.0, momentum=False, engine=None):
        super().__init__(model, epsilon, steps, engine)
        self.attack = "DIFGSM"
        self.decay, self.momentum = decay, momentum

    def _draw(self):
        import random
        if random.random() < 0.5:                                        # :360
            return None
        rnd = torch.randint(224, 250, size=(1, 1)).item()                # :363
        rem = 250 - rnd
        top = torch.randint(0, rem, size=(1, 1)).item()                  # :369
        left = torch.randint(0, rem, size=(1, 1)).item()                 # :371
        return rnd, top, left

    def _gradient(self, adv, labels):
        draw = self._draw()
        if draw is None:
            return self._grad_step(adv, labels)
        rnd, top, left = draw
        b, c, f, h, w = adv.shape
        eng, dev = self.engine, self.engine.device
        my, ylo, yhi = diversity_maps(h, rnd, top)
        mx, xlo, xhi = diversity_maps(w, rnd, left)
        t = lambda a: torch.from_numpy(a).to(dev)                          # noqa: E731
        x = eng.resample_nearest(adv.detach().to(dev).float().contiguous(), t(my), t(mx))      # (b,3,f,224,224), :364-376
        g = self._grad_step(x.to(adv.device), labels).to(dev).float().contiguous()
        return eng.resample_nearest_bwd(g, (h, w), (t(ylo), t(yhi), t(xlo), t(xhi))).to(adv.device)     # (plane-wise: either layout)

    def _pre(self, grad, state):
        return self._l1_momentum(grad, state) if self.momentum else grad

    def _pre_native(self, g, state, shape):
        return self._post_native(g, state, shape, "l1" if self.momentum else None, momentum=bool(self.momentum))


def gaussian_taps(kernlen=15, nsig=3):
    """`_initial_kernel` (`base_attacks.py:425-430`, `:626-634`): the 2-D / 3-D kernels are outer products of `norm.pdf(linspace(-nsig,
    nsig, kernlen))` divided by their sum, i.e. the outer product of this 1-D kernel normalised to sum 1."""
    x = np.linspace(-nsig, nsig, kernlen)
    k = np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)                        # scipy.stats.norm.pdf
    return (k / k.sum()).astype(np.float32)


class TIFGSM(BIM):
    """Translation-Invariant attack (`base_attacks.py:411-469`): every frame of the gradient is smoothed with a 15 x 15 Gaussian
    (depthwise `conv2d`, zero padding 7) -- two 1-D passes of `i2v_dwconv1d_f32` -- and divided by `mean(|.|, [1, 2, 3])` (sic: over
    channels, frames and ROWS, one value per sample and column, :440)."""

    def __init__(self, model, epsilon=16 / 255, steps=10, decay=1.0, momentum=False, engine=None):
        super().__init__(model, epsilon, steps, engine)
        self.attack = "MIFGSM"                                              # (sic) :414
        self.decay, self.momentum = decay, momentum
        self.taps = gaussian_taps(15, 3)

    def _smooth(self, grad):
        eng = self.engine
        g = grad.to(eng.device).float().contiguous()
        g = eng.dwconv1d(eng.dwconv1d(g, self.taps, 4), self.taps, 3)      # along W, then H
        return g / torch.mean(torch.abs(g), [1, 2, 3], True)               # :440

    def _pre(self, grad, state):
        grad = self._smooth(grad).to(grad.device)
        if self.momentum:
            grad = grad + state.get("momentum", torch.zeros_like(grad)) * self.decay
            state["momentum"] = grad
        return grad

    def _smooth_native(self, g, shape):
        """The same separable passes on the frame-major gradient (b*f,c,h,w): along W, then H."""
        eng = self.engine
        return eng.dwconv1d(eng.dwconv1d(g.contiguous(), self.taps, 3), self.taps, 2), "column"          # :440 (sic)

    def _pre_native(self, g, state, shape):
        g, mode = self._smooth_native(g, shape)
        return self._post_native(g, state, shape, mode, momentum=bool(self.momentum))


class TIFGSM3D(TIFGSM):
    """`base_attacks.py:612-675`: a 15 x 15 x 15 Gaussian over (T, H, W) (`conv3d`, groups 3, padding 7) -- three 1-D passes --, then the
    frame-level `norm_grads` (:650)."""

    def __init__(self, model, epsilon=16 / 255, steps=10, decay=1.0, momentum=False, engine=None):
        super().__init__(model, epsilon, steps, decay, momentum, engine)
        self.attack = "TIFGSM3D"

    def _smooth(self, grad):
        eng = self.engine
        g = grad.to(eng.device).float().contiguous()
        g = eng.dwconv1d(eng.dwconv1d(eng.dwconv1d(g, self.taps, 4), self.taps, 3), self.taps, 2)
        return norm_grads(g, True)

    def _smooth_native(self, g, shape):
        """Frame-major gradient viewed (b,f,c,h,w): along W, H, then T -- the torch-module path's pass order."""
        eng = self.engine
        b, c, f, h, w = shape
        assert f == 32                                                      # norm_grads' own assertion (utils.py:59)
        g5 = g.contiguous().view(b, f, c, h, w)
        g5 = eng.dwconv1d(eng.dwconv1d(eng.dwconv1d(g5, self.taps, 4), self.taps, 3), self.taps, 1)
        return g5.view(b * f, c, h, w), "frame"


class SIM(BIM):
    """Scale-Invariant method (`base_attacks.py:554-610`): the mean of the gradients at the clip scaled by 1, 1/2, ... 1/2^(m-1), each
    taken w.r.t. the SCALED clip (:566-571: no chain-rule factor)."""

    def __init__(self, model, epsilon=16 / 255, steps=10, decay=1.0, sclae_step=5, momentum=False, engine=None):
        super().__init__(model, epsilon, steps, engine)
        self.attack = "SIM"
        self.decay, self.momentum, self.sclae_step = decay, momentum, sclae_step

    def _gradient(self, adv, labels):
        mean_grad = None
        for i in range(self.sclae_step):
            g = self._grad_step((1 / 2 ** i * adv).detach(), labels)        # :575-576
            mean_grad = g if mean_grad is None else mean_grad + g
        return mean_grad / self.sclae_step             # (the scaled inputs and this mean are framework elementwise ops on either path)

    def _pre(self, grad, state):
        return self._l1_momentum(grad, state) if self.momentum else grad

    def _pre_native(self, g, state, shape):
        return self._post_native(g, state, shape, "l1" if self.momentum else None, momentum=bool(self.momentum))


class SGM(BIM):
    """Skip Gradient Method (`base_attacks.py:481-551`): the gradient passing BACK through a ReLU is multiplied by gamma ** 0.5 at every
    ReLU module whose qualified name contains 'relu' but not '0.relu' (:511-513: the stem's and those of every residual block but
    the first of its stage -- and, by the same string test, not the 10th / 20th either).  The attacked classifier is the caller's
    torch module, as in the reference; the gain is attached once, at construction, as there (:493) -- to the module's OUTPUT gradient
    (a tensor hook set from a forward hook), which for a ReLU is the same number as the reference's gain on its input gradient
    (the gate is 0 or 1) and also works on in-place ReLUs.  Update rule: `i2v_sign_step_f32`.  With a native `VideoModel` classifier
    the same name test runs over the graph's ReLU convolutions (`graphs.relu_module_names`) and the gain becomes part of the planned
    backward pass (`i2v_net_set_relu_gain`: one in-place pass over the ReLU's finished gradient)."""

    def __init__(self, model, epsilon=16 / 255, steps=10, decay=1.0, gamma=0.5, momentum=False, engine=None):
        super().__init__(model, epsilon, steps, engine)
        self.attack = "SGM"
        self.decay, self.momentum, self.gamma = decay, momentum, gamma
        gain = float(np.power(self.gamma, 0.5))
        self._gain = gain
        self.hooked = []
        if isinstance(model, VideoModel):
            return

        def scale_output_gradient(module, inputs, output):
            if torch.is_tensor(output) and output.requires_grad:
                output.register_hook(lambda g: gain * g)
        self.hooked = []
        for name, module in model.named_modules():
            if self.selects(name) and isinstance(module, nn.ReLU):
                module.register_forward_hook(scale_output_gradient)
                self.hooked.append(name)

    @staticmethod
    def selects(name):
        return "relu" in name and "0.relu" not in name                     # :512

    def _relu_gain(self, graph):
        from .graphs import relu_module_names
        names = relu_module_names(graph)
        self.hooked = sorted({n for n in names.values() if self.selects(n)})
        return {t: self._gain for t, n in names.items() if self.selects(n)}

    def _pre(self, grad, state):
        return self._l1_momentum(grad, state) if self.momentum else grad

    def _pre_native(self, g, state, shape):
        return self._post_native(g, state, shape, "l1" if self.momentum else None, momentum=bool(self.momentum))


class TAP(_SignAttack):
    """Transferable Adversarial Perturbations (`base_attacks.py:685-799`): cost = CE + 1e3 * sum|box-filtered perturbation| + 0.05 *
    sum over the hooked stages of || sgn(a) sqrt|a| - sgn(a0) sqrt|a0| ||_2 (a: the stage's activation on the adversarial clip, a0:
    on the clean one), sign step as BIM.  `params` = {'kernlen', 'temporal_kernlen', 'eta', 'conv3d'} become attributes (:699-700);
    `model_type` has to be among them (the reference reads `self.model_type` in `_find_target_layer` without ever setting it: an
    AttributeError there, and here).  `eta` is accepted and unused, as there (the weight 1e3 is a literal, :780).  The model is the
    caller's torch module (forward hooks on its own stages) or a native `VideoModel` classifier (`_forward_native`: the stages are hooks of
    the planned net, the feature distance is `i2v_tap_distance_f32`); the update is `i2v_sign_step_f32`.  As in the reference the cost is a
    (batch,)-vector, so autograd accepts one clip per call only.  `loss_info[step]` holds the three terms (the reference keys
    the dict with a loop variable its inner loop has rebound to a tensor, :793)."""

    def __init__(self, model, params, epsilon=16 / 255, steps=10, engine=None):
        super().__init__("TAP", model, engine)
        self.epsilon, self.steps = epsilon, steps
        self.step_size = self.epsilon / self.steps
        for name, value in params.items():
            setattr(self, name, value)
        if isinstance(model, VideoModel):            # native classifier: stages hooked in the planned net (`_forward_native`)
            self._find_target_stages()
            return
        k, kt = int(self.kernlen), int(self.temporal_kernlen)
        self.box2d = torch.full((3, 1, k, k), 1.0 / (k * k), dtype=torch.float32)            # :713-717, one filter per colour channel
        self.box3d = torch.full((3, 1, kt, k, k), 1.0 / (kt * k * k), dtype=torch.float32)   # :719-722
        self._features = []
        for stage in self._find_target_layer():
            stage.register_forward_hook(lambda mod, inp, out: self._features.append(out))

    def _find_target_layer(self):
        m = self.model
        if "i3d" in self.model_type:
            return [m.res_layers._modules["0"], m.res_layers._modules["1"]]
        if "slowfast" in self.model_type:
            return [m._modules[n] for n in ("slow_res2", "slow_res3", "fast_res2", "fast_res3")]
        if "tpn" in self.model_type:
            return [m.layer1, m.layer2]

    def _find_target_stages(self):
        """`_find_target_layer` on the graph IR: the stage prefixes whose last block's output is hooked."""
        if "i3d" in self.model_type:
            return ["res_layers.0", "res_layers.1"]
        if "slowfast" in self.model_type:
            return ["slow_res2", "slow_res3", "fast_res2", "fast_res3"]
        raise KeyError(f"TAP on the native classifier: no full graph for {self.model_type!r} (TPN's neck / head are not built)")

    @staticmethod
    def _stage_output(graph, prefix):
        """The tensor a forward hook on the stage MODULE sees (`base_attacks.py:700-711`): the output of the stage's last block --
        which, where gluoncv's `i3d_nl5` appends a non-local block to a Bottleneck, is that block's `nonlocal_block.out`, not the
        Bottleneck's own `.out`."""
        import re
        best = None
        for t, ts in enumerate(graph.tensors):
            m = re.fullmatch(re.escape(prefix) + r"\.(\d+)(\.nonlocal_block)?\.out", ts.name or "")
            if m:
                key = (int(m.group(1)), m.group(2) is not None)
                if best is None or key > best[0]:
                    best = (key, t)
        if best is None:
            raise KeyError(f"no stage {prefix!r} in graph {graph.arch!r}")
        return best[1]

    def _forward_native(self, videos, labels):
        """The whole TAP step behind the C ABI: planned net hooked at the stages AND at the classifier's features; per step forward,
        cross-entropy head (`i2v_head_ce_f32`), feature distance per stage (`i2v_tap_distance_f32`), one input-gradient pass over all
        hooks; the smoothness term's gradient -- 1e3 / std * box(sign(box(perts))), the box filter being symmetric under zero
        padding -- from the depthwise kernels; sign step.  One clip per call, like the reference (its cost is a (batch,)-vector)."""
        eng, m = self.engine, self.model
        dev = eng.device
        videos = videos.to(dev).float().contiguous()
        labels = labels.to(dev)
        b, c, f, h, w = videos.shape
        if b != 1:
            raise RuntimeError("grad can be implicitly created only for scalar outputs")       # what autograd says in the reference
        if self._net is None or self._net_key != (f, h, w):          # planned once per clip shape (packing, upload, autotuning)
            if self._net is not None:
                self._net.close()
            g = m.graph_for((f, h, w))
            self._tap_stages = [self._stage_output(g, p) for p in self._find_target_stages()]
            self._tap_cls = m.classifier_hook(g)
            self._net = eng.build_net(g, m.state_dict_for(g), self._tap_stages + self._tap_cls, b * f)
            self._head = tuple(t.to(dev) if t is not None else None for t in m.head_weights(g))
            self._net_key = (f, h, w)
        net, stages, cls = self._net, self._tap_stages, self._tap_cls
        W, bias = self._head
        kw = dict(dtype=torch.float32, device=dev)
        N = b * f
        x, u = torch.empty(N, 3, h, w, **kw), torch.empty(N, 3, h, w, **kw)
        eng.frames_from_video(videos, x, u)
        net.forward(x)
        ns = [net.hook_frames(i, N) for i in range(len(stages))]
        clean = [net.save_hook(i, ns[i]).contiguous() for i in range(len(stages))]
        unnorm = self._unnorm(videos)
        adv = videos.clone()
        k, kt = int(self.kernlen), int(self.temporal_kernlen)
        taps_s, taps_t = np.full(k, 1.0 / k, np.float32), np.full(kt, 1.0 / kt, np.float32)

        def box(t):
            t = eng.dwconv1d(eng.dwconv1d(t.contiguous(), taps_s, 4), taps_s, 3)
            return eng.dwconv1d(t, taps_t, 2) if self.conv3d else t
        logits, loss_each = torch.empty(b, W.shape[0], **kw), torch.empty(b, **kw)
        hscratch = torch.empty(eng.capi.i2v_head_scratch_bytes(W.shape[1], b), dtype=torch.uint8, device=dev)
        dist = [torch.empty(b, **kw) for _ in stages]
        dscratch = [torch.empty(net.ilaf_scratch_bytes(i, ns[i], ns[i] // b), dtype=torch.uint8, device=dev) for i in range(len(stages))]
        feats = list(range(len(stages), len(stages) + len(cls)))
        self.loss_info = {}
        gx = torch.empty_like(x)
        logged = torch.zeros(self.steps, 2 + len(stages), **kw)          # per step: CE, regulariser, distance per stage
        for step in range(self.steps):
            eng.frames_from_video(adv, x, u)
            net.forward(x)
            net.head_ce(feats[0] if len(feats) == 1 else feats, W, bias, labels.to(torch.int32).contiguous(), N, float(self._targeted),
                        logits, loss_each, hscratch)
            for i in range(len(stages)):
                net.tap_distance(i, clean[i], 0.05, dist[i], dscratch[i], ns[i], ns[i] // b)
            net.backward(gx)
            smooth = box(eng.tap_perts(adv, videos))                    # (sic) `_transform_perts` divides by std
            sg = eng.tap_sign_abs(smooth, logged[step, 1:2])            # sign(smooth), and the regulariser's value for the log
            grad = eng.tap_grad(gx, box(sg), 1e3)                       # clip-layout gradient + 1e3 * box(sign(box(perts))) / std
            eng.sign_step(adv, unnorm, grad, f * h * w, self.step_size, self.epsilon)
            logged[step, 0:1].copy_(loss_each)                          # the step's logged terms stay on the device (one clip per call)
            for i in range(len(stages)):
                logged[step, 2 + i:3 + i].copy_(dist[i])
        host = logged.cpu().numpy()                                      # one read-back per call
        for step in range(self.steps):
            self.loss_info[step] = {"ce loss": host[step, 0], "reg_cost": host[step, 1],
                                    "distance": np.float32(sum(np.float32(v) for v in host[step, 2:]))}
        return adv

    def _smoothness(self, perts):
        k, kt = int(self.kernlen), int(self.temporal_kernlen)
        if self.conv3d:
            out = nn.functional.conv3d(perts, self.box3d.to(perts.device), groups=3, stride=1, padding=[(kt - 1) // 2, (k - 1) // 2, (k - 1) // 2])
        else:                                                               # frame by frame (:724-731); the 2-D filter stays float32
            out = torch.stack([nn.functional.conv2d(perts[:, :, f], self.box2d.to(perts.device), groups=3, stride=1,
                                                    padding=[(k - 1) // 2, (k - 1) // 2]) for f in range(perts.shape[2])], 2)
        return torch.sum(torch.abs(out))

    def forward(self, videos, labels):
        if isinstance(self.model, VideoModel):
            return self._forward_native(videos, labels)
        videos = videos.to(self.device).float().contiguous()
        labels = labels.to(self.device)
        b, c, f, h, w = videos.shape
        self.loss_info = {}
        self._features = []
        self.model(videos)
        clean = self._features
        root = lambda a: torch.sign(a) * torch.sqrt(torch.abs(a))          # noqa: E731
        std = torch.as_tensor(self.std, dtype=videos.dtype, device=videos.device)[:, None, None, None]
        u = self._unnorm(videos)
        adv = videos.clone().detach()
        for step in range(self.steps):
            self._features = []
            adv.requires_grad = True
            cost1 = self._targeted * nn.CrossEntropyLoss()(self.model(adv), labels)
            cost2 = torch.sum(torch.stack([torch.norm(root(a).reshape(b, -1) - root(a0).reshape(b, -1), p=2, dim=1)
                                           for a, a0 in zip(self._features, clean)]), 0)
            reg = self._smoothness((adv - videos) / std)                    # (sic) divided by std, `_transform_perts` :138-143
            cost = cost1 + 1e3 * reg + 0.05 * cost2
            grad = torch.autograd.grad(cost, adv, retain_graph=False, create_graph=False)[0]
            self.loss_info[step] = {"ce loss": cost1.detach().cpu().numpy(), "reg_cost": reg.detach().cpu().numpy(),
                                    "distance": cost2.detach().cpu().numpy()}
            adv = adv.detach()
            self.engine.sign_step(adv, u, grad.contiguous(), f * h * w, self.step_size, self.epsilon)
        return adv


def run_concurrent(make_attack, items, streams=2, device=None, on_result=None):
    """Run `attack(*item)` for every item of the iterable `items` on `streams` concurrent clip streams.  Returns
    (results in item order, attack objects); with `on_result(index, item, result)` given, results are handed to it
    as they finish (serialised) instead of being kept, and items are pulled lazily.

    ILAF handles ONE clip per call (`image_fine_tune_attack.py:73-79`; its loss couples the whole batch, so clips
    cannot be batched), and one 32-frame clip leaves most of an MI355X idle between its ~3000 short launches.  Each
    worker thread owns an attack object (its own planned net and buffers) and a HIP stream, so launches of
    different clips overlap on the device: +45..55 % clips/s with 2 streams, nothing more with 4
    (tools/ilaf_streams_probe.py).  Per-clip results are bit-identical to the sequential run."""
    import threading
    it = enumerate(items)
    pull, push = threading.Lock(), threading.Lock()
    results, errors = {}, []
    n = max(1, streams)
    attacks_ = [make_attack() for _ in range(n)]
    if device is None:                   # the engine's device: a new thread starts on device 0, not on the caller's
        eng = getattr(attacks_[0], "_engine", None) or get_engine()
        device = eng.device
    device = torch.device(device)
    use_cuda = device.type == "cuda" and torch.cuda.is_available()

    def worker(atk):
        ctx = None
        try:
            if use_cuda:
                torch.cuda.set_device(device)            # per-thread state, for torch and for the HIP runtime underneath
                ctx = torch.cuda.stream(torch.cuda.Stream(device=device))
                ctx.__enter__()
            while not errors:
                with pull:
                    nxt = next(it, None)
                if nxt is None:
                    break
                i, item = nxt
                res = atk(*item)
                if ctx is not None:
                    torch.cuda.current_stream().synchronize()
                with push:
                    if on_result is not None:
                        on_result(i, item, res)
                    else:
                        results[i] = res
        except BaseException as e:       # surfaced by the caller
            errors.append(e)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
    # plan every worker's net BEFORE the first stream starts (plan-time autotuning wants a quiet device): on the first item's shape
    first = next(it, None)
    if first is not None:
        import itertools
        it = itertools.chain([first], it)
        for a in attacks_:
            if hasattr(a, "plan_for"):
                try:
                    a.plan_for(*first[1])
                except Exception:            # noqa: BLE001 -- planning here is an optimisation; the call itself will plan (or raise) in its worker
                    pass
        if use_cuda:
            torch.cuda.synchronize(device)
    threads = [threading.Thread(target=worker, args=(a,)) for a in attacks_]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return [results[i] for i in sorted(results)], attacks_


class ILAF(object):
    """`image_attacks.py:498-629`: fine-tunes an existing adversarial clip along the feature
    direction of a white-box VIDEO model.  `model` is a `VideoModel` (native path, `_native`) or a torch
    module (then the model, its hooks, the differentiable compose and the loss stay in PyTorch and the sign
    update runs in the library).
    The reference's return value applies reshape(b,f,c,h,w).permute(0,2,1,3,4) to a tensor that is
    already (b,c,f,h,w) (`:627-628`); that scramble is reproduced for drop-in fidelity."""

    def __init__(self, model, model_type, step_size=0.005, epsilon=16 / 255, steps=60, hook_layers=None, engine=None):
        self.attack = "ILAF"
        self.epsilon, self.steps, self.step_size = epsilon, steps, step_size
        self.loss_info = {}
        self.model_type, self.model = model_type, model
        self.mean, self.std = MEAN, STD
        self._engine = engine
        self.activations = {"value": []}
        self._net = self._net_key = None
        self.path = "native" if isinstance(model, VideoModel) else "torch-module"
        _announce(self)
        if isinstance(model, VideoModel):
            return                                      # native path: hooks are tensors of the graph IR
        layers = hook_layers if hook_layers is not None else self._find_target_layer()
        for layer in (layers if isinstance(layers, list) else [layers]):
            layer.register_forward_hook(lambda mod, inp, out: self.activations["value"].append(out))

    def _find_target_layer(self):
        if "i3d" in self.model_type:
            return self.model.res_layers._modules["1"]
        if "slowfast" in self.model_type:
            return [self.model._modules["slow_res2"], self.model._modules["fast_res2"]]
        if "tpn" in self.model_type:
            return self.model.layer2

    def _features(self, x):
        self.activations["value"] = []
        out = self.model(x)
        return list(self.activations["value"]), out

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    #: True: the b clips of a call are b INDEPENDENT one-clip problems (what b calls of the reference, which fine-tunes one
    #: clip per call -- image_fine_tune_attack.py:73-79 --, compute): every norm, loss and logged cost is per clip, and every
    #: clip's output is bit-identical to its own one-clip call.  False (the reference's semantics for a b > 1 call): the
    #: whole-tensor norms of image_attacks.py:563-567,595-613 run over all b clips together.
    independent_clips = False

    def _ensure_net(self, eng, b, f, h, w):
        N, key = b * f, (f, h, w)
        if self._net is None or self._net_key != key or self._net.max_frames < N:
            if self._net is not None:
                self._net.close()
            g = self.model.graph_for((f, h, w))
            self._net = eng.build_net(g, self.model.state_dict_for(g), self.model.hook_tensors(g), N)
            self._net_key = key
        return self._net

    def plan_for(self, videos, *_unused):
        """Plan (and autotune) the native net for calls of this shape NOW, in the calling thread.  Planning times every launch on the
        device, so it wants the device to itself: callers that run several attack objects on concurrent streams (`run_concurrent`,
        `bench.py --workload ilaf`) plan them one after the other before the first stream starts -- nets planned while other streams
        were already executing picked different kernels per stream from polluted timings (round 6)."""
        if isinstance(self.model, VideoModel):
            b, _, f, h, w = videos.shape
            self._ensure_net(self._engine or get_engine(), b, f, h, w)

    def forward_independent(self, videos, ori_videos, labels, video_names):
        """K one-clip ILAF problems in ONE launch list (segmented loss kernels, `i2v_ilaf_*_seg_f32`): the convolution launches
        see K times the frames -- real occupancy instead of HIP-stream concurrency.  Native models only."""
        if not isinstance(self.model, VideoModel):
            raise TypeError("forward_independent needs a native VideoModel")
        return self._native(videos, ori_videos, video_names, independent=True)

    def _native(self, videos, ori_videos, video_names, independent=None, modifier0=None, keep_gradient=False):
        """The whole of `image_attacks.py:534-629` behind the C ABI.  Activations are frame-major (b*T, C, H, W); the
        loss only needs whole-tensor norms and a dot product, so no layout change is ever materialised.
        `modifier0` (b*f, 3, h, w; frame-major like `self._modifier`) starts the loop from that perturbation instead of
        `videos - ori_videos` and `keep_gradient` leaves the last step's input gradient in `self._last_gx` -- the teacher-forced step
        of the parity tests (`tests/test_gpu_video.py`), nothing the reference's call protocol has."""
        eng = self._engine or get_engine()
        dev = eng.device
        kw = dict(dtype=torch.float32, device=dev)
        videos = videos.detach().to(**kw).contiguous()
        ori = ori_videos.detach().to(**kw).contiguous()
        b, c, f, h, w = videos.shape
        independent = self.independent_clips if independent is None else independent
        nseg = b if independent else 1                                          # loss segments: one per clip, or the whole batch
        N, eps = b * f, float(self.epsilon)
        net = self._ensure_net(eng, b, f, h, w)
        L = len(net.hooks)
        x, u_adv, u_ori = (torch.empty(N, 3, h, w, **kw) for _ in range(3))
        eng.frames_from_video(ori, x, u_ori)                                     # :572 `_transform_video_ILAF(..,'back')`
        net.forward(x)                                                          # :545-549 clean features
        nf = [net.hook_frames(i, N) for i in range(L)]
        fps = [n // nseg for n in nf]                                           # frames per segment at each hook
        ori_f = [net.save_hook(i, nf[i]) for i in range(L)]
        eng.frames_from_video(videos, x, u_adv)
        net.forward(x)                                                          # :555-559 features of the given adversarial clip
        adv_f = [net.save_hook(i, nf[i]) for i in range(L)]
        scratch = torch.empty((max(net.ilaf_scratch_bytes(i, nf[i], fps[i]) for i in range(L)) + 64) // 4, **kw)
        init_sq = torch.empty(L, nseg, dtype=torch.float64, device=dev)         # |adv0 - ori|^2 per (layer, segment), stays on the device
        for i in range(L):                                                      # :563-567 |adv0 - ori| per layer
            net.ilaf_reduce(i, ori_f[i], adv_f[i], scratch, nf[i], act=adv_f[i], frames_per_seg=fps[i])
            init_sq[i].copy_(scratch[:4 * nseg].view(torch.float64).view(nseg, 2)[:, 0])
        # A given adversarial clip whose hooked features equal its original's has no direction to follow: |adv0 - ori| = 0 makes
        # every later loss and gradient of that segment 0/0 (the reference would return a NaN clip, :566-567).  One small
        # read-back per call (60 steps) instead of a silently saved NaN file.
        dead = (init_sq <= 0).any(dim=0).nonzero().flatten().tolist()
        if dead:
            who = [video_names[k] if independent and k < len(video_names) else k for k in dead]
            raise ValueError(f"ILAF: the given adversarial clip equals its original at a hooked layer (|adv0 - ori| = 0), "
                             f"nothing to fine-tune: {who}")
        modifier = torch.sub(u_adv, u_ori)                                      # :574-575 existing perturbation
        if modifier0 is not None:
            modifier = modifier0.detach().to(**kw).reshape(N, 3, h, w).clone()
        gx = torch.empty_like(x)
        loss = torch.zeros(L, nseg, **kw)
        costs = torch.zeros(self.steps, nseg, **kw)
        slot = torch.zeros(1, dtype=torch.long, device=dev)

        def one_step():
            eng.compose(u_ori, modifier, x, b, f, eps)                          # :585-588
            net.forward(x)                                                      # :591
            for k in range(L):                                                  # :599-610
                net.ilaf_reduce(k, ori_f[k], adv_f[k], scratch, nf[k], frames_per_seg=fps[k])
                net.ilaf_grad(k, ori_f[k], adv_f[k], init_sq[k], loss[k], scratch, nf[k], frames_per_seg=fps[k])
            net.backward(gx)                                                    # :613-614 (input gradient only)
            if keep_gradient:
                self._last_gx, self._last_modifier_in = gx.clone(), modifier.clone()
            eng.sign_step_delta_gx(modifier, gx, u_ori, eps, self.step_size)    # :617
            costs.index_copy_(0, slot, loss.sum(dim=0, keepdim=True))           # :611, stays on the device
            slot.add_(1)

        # (Recording the step into a HIP graph was measured and dropped: 272.8 vs 274.8 frames/s without it --
        # the loop is device-bound even for a single clip.)
        for _ in range(self.steps):
            one_step()
        costs_h = costs.cpu().numpy()                                           # the call's only read-back
        self.last_costs = costs_h if independent else costs_h[:, 0]
        for i in range(self.steps):
            for k, name in enumerate(video_names):
                cost = costs_h[i, k if independent else 0]
                self.loss_info.setdefault(name, {})[i] = {"cost": str(np.asarray(cost, dtype=np.float32))}
        out = torch.empty(b, 3, f, h, w, **kw)
        eng.compose(u_ori, modifier, out, b, f, eps, video_layout=True)         # :625-626
        self._modifier = modifier
        if independent:         # each clip through the reference's (sic) reshape / permute of ITS one-clip call (:627-628)
            return out.reshape(b, 1, f, c, h, w).permute(0, 1, 3, 2, 4, 5).reshape(b, c, f, h, w)
        return out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)                # :627-628 (sic, see class docstring)

    def forward(self, videos, ori_videos, labels, video_names):
        if isinstance(self.model, VideoModel):
            return self._native(videos, ori_videos, video_names)
        dev = next(self.model.parameters()).device
        eng = self._engine or get_engine(str(dev) if dev.type == "cuda" else None)
        videos, ori = videos.to(dev).float().contiguous(), ori_videos.to(dev).float().contiguous()
        b, c, f, h, w = videos.shape
        mean = torch.as_tensor(self.mean, device=dev)[None, :, None, None, None]
        std = torch.as_tensor(self.std, device=dev)[None, :, None, None, None]
        with torch.no_grad():
            ori_f, _ = self._features(ori)
            adv_f, _ = self._features(videos)
        init_dirs, init_norms = [], []
        for o, a in zip(ori_f, adv_f):
            d = a - o
            init_norms.append(torch.norm(d, p=2))
            init_dirs.append(d / torch.norm(d, p=2, keepdim=True))
        ori_u = ori.clone().mul_(std).add_(mean)
        modifier = (videos.clone().mul_(std).add_(mean) - ori_u).contiguous()
        for i in range(self.steps):
            modifier.requires_grad = True
            x = (torch.clamp(ori_u + torch.clamp(modifier, -self.epsilon, self.epsilon), 0, 1) - mean) / std
            step_f, _ = self._features(x)
            losses = []
            for k, (o, a) in enumerate(zip(ori_f, step_f)):
                d = a - o
                nrm = torch.norm(d, p=2)
                dirn = d / torch.norm(d, p=2, keepdim=True)
                losses.append(-(0.5 * nrm / init_norms[k] + torch.mm(init_dirs[k].view(1, -1), dirn.view(1, -1).t())))
            cost = torch.sum(torch.stack(losses))
            grad = torch.autograd.grad(cost, modifier, retain_graph=False, create_graph=False)[0].contiguous()
            modifier = modifier.detach()
            eng.sign_step_delta(modifier, grad, self.step_size)                      # :617
            for name in video_names:
                self.loss_info.setdefault(name, {})[i] = {"cost": str(cost.detach().cpu().numpy())}
        out = (torch.clamp(ori_u + torch.clamp(modifier, -self.epsilon, self.epsilon), 0, 1) - mean) / std
        return out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)                     # :627-628 (sic)
