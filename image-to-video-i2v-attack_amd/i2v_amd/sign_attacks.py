"""White-box sign-step attacks (`/root/reference/base_attacks.py:236-340`) and ILAF's update
(`image_attacks.py:498-629`).

What is in scope here is the UPDATE RULE: un-normalise, `+ step*sign(g)`,
project to +-eps, clamp to [0,1], re-normalise -- one fused HIP kernel (`i2v_sign_step_f32`).  The
attacked VIDEO model (gluoncv I3D/SlowFast/TPN in the reference) is not part of this hot path: it
is whatever differentiable torch module the caller passes, exactly as in the reference, and its
forward/backward run in PyTorch.
"""
import torch
import torch.nn as nn

from .attacks import get_engine

MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


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

    def _grad(self, adv, labels):
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
        grad = self._grad(videos.clone().detach(), labels).contiguous()
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

    def forward(self, videos, labels):
        videos = videos.to(self.device).float().contiguous()
        labels = labels.to(self.device)
        u = self._unnorm(videos)
        adv = videos.clone().detach()
        b, c, f, h, w = videos.shape
        state = {}
        for _ in range(self.steps):
            grad = self._pre(self._grad(adv, labels), state).contiguous()
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


class ILAF(object):
    """`image_attacks.py:498-629`: fine-tunes an existing adversarial clip along the feature
    direction of a white-box VIDEO model.  The model, its hooks, the differentiable compose and the
    loss stay in PyTorch (the 3-D backbones are gluoncv's, SURVEY.md 8(f) N2, and autograd has to see
    the path from `modifier` to the features); the sign update runs in the library.
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

    def forward(self, videos, ori_videos, labels, video_names):
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
