"""The white-box VIDEO attack `attack.py --attack_type video` runs (`/root/reference/video_attacks.py:14-229`,
"Boosting the transferability of video adversarial examples via temporal translation"): every step attacks D = kernlen
cyclically frame-shifted copies of the clip, mixes their input gradients over the shifts with a temporal kernel, optionally
accumulates momentum, and takes the BIM sign / clip / project step.

Built on the same pieces as the BIM family: the classifier gradient comes from `_SignAttack._grad` (natively through
`VideoModel(..., num_classes=K)`, or from the caller's torch module as in the reference), the gradient mix is one fused
kernel (`i2v_tt_grad_mix_f32`) and the update is `i2v_sign_step_f32`.  The frame shifts are `torch.roll` (data movement).
"""
import math
import random

import numpy as np
import torch

from .sign_attacks import _SignAttack, norm_grads


class TemporalTranslation(_SignAttack):
    """`params` as built by `attack.py:79`: kernlen, momentum, weight, move_type ('adj' | 'large' | 'random'), kernel_mode
    ('gaussian' | 'linear' | 'random')."""

    def __init__(self, model, params, epsilon=16 / 255, steps=10, delay=1.0, engine=None):
        super().__init__("TemporalTranslation", model, engine)
        self.epsilon, self.steps, self.delay = epsilon, steps, delay
        self.step_size = self.epsilon / self.steps
        for name, value in params.items():
            setattr(self, name, value)
        self.frames = 32                                                  # video_attacks.py:35
        max_move = int((self.kernlen - 1) / 2)
        self.cycle_move_list = list(range(-max_move, max_move + 1))       # :46-48
        self.kernel = self._temporal_kernel(self.kernel_mode, self.kernlen).astype(np.float32)

    @staticmethod
    def _temporal_kernel(mode, kernlen):
        """:49-80 -- normalised weights over the shifts (float64 until the final cast, as numpy computes them there)."""
        if mode == "gaussian":
            assert kernlen % 2 == 1
            k = (kernlen - 1) / 2
            sigma = k / 3
            k = int(k)
            kern = np.array([1 / (sigma * np.sqrt(2 * np.pi)) * math.exp(-(x ** 2) / (2 * sigma ** 2)) for x in range(-k, k + 1)])
        elif mode == "linear":
            k = int((kernlen - 1) / 2)
            half = [1 - i / (k + 1) for i in range(k + 1)]
            kern = np.array(half[::-1][:-1] + half)
        elif mode == "random":                                            # the reference's name for the uniform kernel (:42-43)
            kern = np.ones(kernlen)
        else:
            raise UnboundLocalError("local variable 'kernel' referenced before assignment")     # as the reference fails (:38-45)
        return kern / kern.sum()

    def _shift(self, cycle_move):
        """Frames to roll a clip forward by for one entry of the move list (:96-141)."""
        direction = -1 if cycle_move < 0 else 1
        m = abs(cycle_move)
        if self.move_type == "adj":
            m = m % self.frames
        elif self.move_type == "large":
            m = m % self.frames if m == 0 else (m + (int(self.frames / 2) - 1)) % self.frames
        elif self.move_type == "random":
            m = cycle_move % self.frames if cycle_move == 0 else random.randint(0, 100) % self.frames
        else:
            raise UnboundLocalError("local variable 'new_videos' referenced before assignment")  # :196-203
        return direction * m

    def forward(self, videos, labels):
        videos = videos.to(self.device).float().contiguous()
        labels = labels.to(self.device)
        b, c, f, h, w = videos.shape
        u = self._unnorm(videos)
        adv = videos.clone().detach()
        momentum = torch.zeros_like(videos)
        length = len(self.cycle_move_list)
        batch_times = length if self.model_name == "TPNet" else 5         # :206-210
        chunk = math.ceil(length / batch_times)
        for _ in range(self.steps):
            # a clip rolled forward by m frames: new[(i + m) % T] = old[i]  (:96-109)
            variants = torch.cat([torch.roll(adv, self._shift(m), dims=2) for m in self.cycle_move_list], dim=0)
            grads = []
            for i in range(batch_times):
                part = variants[i * chunk:min((i + 1) * chunk, length)]
                if part.shape[0] == 0:                                    # kernlen not a multiple of ceil(kernlen / 5): the reference
                    continue                                              # would hand its model an empty batch here
                used_labels = torch.cat([labels] * part.shape[0], dim=0)  # :151-152 (mean cross-entropy over the chunk)
                grads.append(self._grad(part.clone().detach(), used_labels))
            grads = torch.cat(grads, dim=0).unsqueeze(1).contiguous()     # (D, 1, C, T, H, W), :215-216
            grad = self.engine.tt_grad_mix(grads.to(self.engine.device), self.kernel, self.cycle_move_list, float(self.weight)).to(adv.device)
            if self.momentum:                                             # :220-225
                grad = norm_grads(grad)
                grad = grad + momentum * self.delay
                momentum = grad
            adv = adv.detach()
            self.engine.sign_step(adv, u, grad.contiguous(), f * h * w, self.step_size, self.epsilon)   # :227-231
        return adv
