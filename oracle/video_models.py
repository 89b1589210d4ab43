"""TEST INFRASTRUCTURE -- torch restatement of the video backbones ILAF hooks.

The reference obtains I3D / SlowFast from gluoncv 0.10.4 (`/root/reference/image_fine_tune_attack.py:63`,
configs `utils.py:9-14`); gluoncv is not vendored and not installed, so `i2v_amd.graphs.i3d_resnet` /
`slowfast_res2` restate the public architectures and this file is their plain-PyTorch counterpart
(`nn.Conv3d`, `nn.BatchNorm3d`, `nn.MaxPool3d`, autograd).  PARITY UNPINNED against gluoncv's own module
layout; what IS pinned here is that the HIP engine computes exactly these modules (tests/test_video_*).

The modules expose the attribute names the reference's `ILAF._find_target_layer` looks up
(`image_attacks.py:513-519`): `res_layers._modules['1']` for I3D, `_modules['slow_res2'/'fast_res2']` for
SlowFast, so the UNMODIFIED reference class runs on them (oracle/make_golden.py).  `state_dict()` keys are the
graph IR's weight keys.
"""
import torch
import torch.nn as nn


class NonLocal3d(nn.Module):
    """gluoncv / mmaction `NonLocalModule` as the reference's `i3d_nl5_*` models use it (restated from the public code, gluoncv
    is not installed): embedded gaussian, `sub_sample=True` (a 1x2x2 max-pool in front of phi and g), `use_bn=True`.
    State-dict keys: theta.{weight,bias}, phi.1.*, g.1.*, W.0.{weight,bias}, W.1.{BatchNorm}."""

    def __init__(self, in_channels):
        super().__init__()
        e = in_channels // 2
        self.theta = nn.Conv3d(in_channels, e, 1)
        self.max_pool = nn.MaxPool3d((1, 2, 2))
        self.phi = nn.Sequential(self.max_pool, nn.Conv3d(in_channels, e, 1))
        self.g = nn.Sequential(self.max_pool, nn.Conv3d(in_channels, e, 1))
        self.W = nn.Sequential(nn.Conv3d(e, in_channels, 1), nn.BatchNorm3d(in_channels))
        self.softmax = nn.Softmax(dim=2)

    def forward(self, x):
        theta, phi, g = self.theta(x), self.phi(x), self.g(x)
        shape = theta.shape
        theta, phi, g = theta.reshape(shape[0], shape[1], -1), phi.reshape(shape[0], shape[1], -1), g.reshape(shape[0], shape[1], -1)
        p = self.softmax(torch.matmul(theta.transpose(1, 2), phi))          # (b, THW, T'H'W')
        y = torch.matmul(g, p.transpose(1, 2)).reshape(shape)
        return self.W(y) + x


class Bottleneck3d(nn.Module):
    def __init__(self, inplanes, planes, stride=1, head_t=1, downsample=None, nonlocal_block=False):
        super().__init__()
        self.nonlocal_block = NonLocal3d(planes * 4) if nonlocal_block else None
        self.conv1 = nn.Conv3d(inplanes, planes, (head_t, 1, 1), padding=(head_t // 2, 0, 0), bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, (1, 3, 3), stride=(1, stride, stride), padding=(0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        out = self.relu(out + idt)
        return out if self.nonlocal_block is None else self.nonlocal_block(out)


def _stage(inplanes, planes, blocks, stride, head_t_of, nl_of=lambda b: False):
    layers = []
    for b in range(blocks):
        s = stride if b == 0 else 1
        ds = None
        if s != 1 or inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv3d(inplanes, planes * 4, 1, stride=(1, s, s), bias=False), nn.BatchNorm3d(planes * 4))
        layers.append(Bottleneck3d(inplanes, planes, s, head_t_of(b), ds, nl_of(b)))
        inplanes = planes * 4
    return nn.Sequential(*layers), inplanes


class I3DResNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), width=64, inflate=((1, 1, 1), (1, 0, 1, 0), (1, 0, 1, 0, 1, 0), (0, 1, 0)), nonlocal_freq=None):
        super().__init__()
        self.conv1 = nn.Conv3d(3, width, (5, 7, 7), stride=(2, 2, 2), padding=(2, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d((1, 3, 3), stride=(2, 2, 2), padding=(0, 1, 1))
        self.pool2 = nn.MaxPool3d((2, 1, 1), stride=(2, 1, 1))
        stages, inplanes = [], width
        for li, nb in enumerate(layers):
            infl = inflate[li]
            nl = nonlocal_freq[li] if nonlocal_freq is not None and li < len(nonlocal_freq) else ()
            st, inplanes = _stage(inplanes, width * 2 ** li, nb, 1 if li == 0 else 2,
                                  lambda b, infl=infl: 3 if infl[b % len(infl)] else 1,
                                  lambda b, nl=nl: bool(b < len(nl) and nl[b]))
            stages.append(st)
        self.res_layers = nn.Sequential(*stages)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        for i, st in enumerate(self.res_layers):
            x = st(x)
            if i == 0:
                x = self.pool2(x)
        return x


class SlowFastRes2(nn.Module):
    """SlowFast stems + lateral connection + the two res2 stages (the part ILAF's hooks depend on)."""

    def __init__(self, width=64, slow_stride=8, fast_stride=1, beta_inv=8, fusion_ratio=2, fusion_kernel=5, blocks=3):
        super().__init__()
        fw = width // beta_inv
        self.slow_stride, self.fast_stride = slow_stride, fast_stride
        self.fast_conv1 = nn.Conv3d(3, fw, (5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False)
        self.fast_bn1 = nn.BatchNorm3d(fw)
        self.fast_maxpool = nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.slow_conv1 = nn.Conv3d(3, width, (1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), bias=False)
        self.slow_bn1 = nn.BatchNorm3d(width)
        self.slow_maxpool = nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.relu = nn.ReLU(inplace=True)
        alpha = slow_stride // fast_stride
        self.lateral_p1 = nn.Sequential(
            nn.Conv3d(fw, fw * fusion_ratio, (fusion_kernel, 1, 1), stride=(alpha, 1, 1), padding=(fusion_kernel // 2, 0, 0), bias=False),
            nn.BatchNorm3d(fw * fusion_ratio), nn.ReLU(inplace=True))
        self.fast_res2, _ = _stage(fw, fw, blocks, 1, lambda b: 3)
        self.slow_res2, _ = _stage(width + fw * fusion_ratio, width, blocks, 1, lambda b: 1)

    def forward(self, x):
        fast_in, slow_in = x[:, :, ::self.fast_stride], x[:, :, ::self.slow_stride]
        f = self.fast_maxpool(self.relu(self.fast_bn1(self.fast_conv1(fast_in))))
        lat = self.lateral_p1(f)
        fr = self.fast_res2(f)
        s = self.slow_maxpool(self.relu(self.slow_bn1(self.slow_conv1(slow_in))))
        sr = self.slow_res2(torch.cat([s, lat], dim=1))
        return sr, fr


class SlowFastNet(nn.Module):
    """The whole SlowFast backbone (both pathways through res5, lateral connections in front of every slow stage):
    counterpart of `graphs.slowfast_resnet`; forward returns (slow_res5, fast_res5), which a classifier head pools and
    concatenates in that order."""

    def __init__(self, layers=(3, 4, 6, 3), width=64, slow_stride=8, fast_stride=1, beta_inv=8, fusion_ratio=2, fusion_kernel=5):
        super().__init__()
        fw = width // beta_inv
        self.slow_stride, self.fast_stride = slow_stride, fast_stride
        self.fast_conv1 = nn.Conv3d(3, fw, (5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False)
        self.fast_bn1 = nn.BatchNorm3d(fw)
        self.fast_maxpool = nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.slow_conv1 = nn.Conv3d(3, width, (1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), bias=False)
        self.slow_bn1 = nn.BatchNorm3d(width)
        self.slow_maxpool = nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.relu = nn.ReLU(inplace=True)
        alpha = slow_stride // fast_stride

        def lateral(c):
            return nn.Sequential(
                nn.Conv3d(c, c * fusion_ratio, (fusion_kernel, 1, 1), stride=(alpha, 1, 1), padding=(fusion_kernel // 2, 0, 0), bias=False),
                nn.BatchNorm3d(c * fusion_ratio), nn.ReLU(inplace=True))
        self.lateral_p1 = lateral(fw)
        slow_t = (1, 1, 3, 3)
        fin, sin = fw, width + fw * fusion_ratio
        for li, nb in enumerate(layers):
            stride = 1 if li == 0 else 2
            fs, fout = _stage(fin, fw * 2 ** li, nb, stride, lambda b: 3)
            ss, sout = _stage(sin, width * 2 ** li, nb, stride, lambda b, t=slow_t[li]: t)
            setattr(self, f"fast_res{li + 2}", fs)
            setattr(self, f"slow_res{li + 2}", ss)
            if li < len(layers) - 1:
                setattr(self, f"lateral_res{li + 2}", lateral(fout))
                sin = sout + fout * fusion_ratio
            fin = fout
        self.nstages = len(layers)

    def forward(self, x):
        fast_in, slow_in = x[:, :, ::self.fast_stride], x[:, :, ::self.slow_stride]
        f = self.fast_maxpool(self.relu(self.fast_bn1(self.fast_conv1(fast_in))))
        s = self.slow_maxpool(self.relu(self.slow_bn1(self.slow_conv1(slow_in))))
        s = torch.cat([s, self.lateral_p1(f)], dim=1)
        for li in range(self.nstages):
            f = getattr(self, f"fast_res{li + 2}")(f)
            s = getattr(self, f"slow_res{li + 2}")(s)
            if li < self.nstages - 1:
                s = torch.cat([s, getattr(self, f"lateral_res{li + 2}")(f)], dim=1)
        return s, f


class TPNBackbone(nn.Module):
    """SlowOnly-style backbone of TPN up to `layer2` (+ `layer3` so that something runs behind the hook): stem 1x7x7,
    no temporal kernels in layer1 / layer2.  `model.layer2` is what the reference hooks (`image_attacks.py:517-518`)."""

    def __init__(self, layers=(3, 4, 6, 3), width=64):
        super().__init__()
        self.conv1 = nn.Conv3d(3, width, (1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.layer1, c = _stage(width, width, layers[0], 1, lambda b: 1)
        self.layer2, c = _stage(c, width * 2, layers[1], 2, lambda b: 1)
        self.layer3, c = _stage(c, width * 4, layers[2], 2, lambda b: 3)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer3(self.layer2(self.layer1(x)))


def make(model_type: str, tiny: bool, full: bool = False) -> nn.Module:
    """Counterpart of `i2v_amd.graphs.build_video` / `build_video_tiny`."""
    if full and "slowfast" in model_type:
        if tiny:
            return SlowFastNet((2, 2, 1, 1), 16, slow_stride=4, fast_stride=1, beta_inv=4)
        return SlowFastNet((3, 4, 23, 3) if "101" in model_type else (3, 4, 6, 3), slow_stride=8, fast_stride=2, fusion_kernel=7)
    if "tpn" in model_type:
        if tiny:
            return TPNBackbone((2, 2, 1, 1), 8)
        return TPNBackbone((3, 4, 23, 3) if "101" in model_type else (3, 4, 6, 3))
    if "i3d" in model_type:          # the reference's I3D models are the non-local i3d_nl5 ones (utils.py:9-10); 'i3d_plain_*' are not
        from i2v_amd.graphs import NL5_FREQ, TINY_NL_FREQ
        plain = "plain" in model_type
        if tiny:
            return I3DResNet((2, 2, 1, 1), 8, inflate=((1, 1), (1, 0), (1,), (0,)), nonlocal_freq=None if plain else TINY_NL_FREQ)
        return I3DResNet((3, 4, 23, 3) if "101" in model_type else (3, 4, 6, 3), nonlocal_freq=None if plain else NL5_FREQ)
    if tiny:
        return SlowFastRes2(16, slow_stride=4, fast_stride=1, beta_inv=4, blocks=2)
    return SlowFastRes2(slow_stride=8, fast_stride=2, fusion_kernel=7)          # the 8x8 configuration the reference names (utils.py:11-12)


def load_weights(model: nn.Module, sd: dict) -> nn.Module:
    """Load a graph-IR state_dict (only the stages the graph reaches carry keys; the rest keep their init)."""
    own = model.state_dict()
    for k, v in sd.items():
        assert k in own and tuple(own[k].shape) == tuple(v.shape), (k, tuple(v.shape), tuple(own[k].shape) if k in own else None)
        own[k] = v.clone()
    model.load_state_dict(own)
    return model.eval()


def hook_modules(model: nn.Module, model_type: str):
    """The modules the reference hooks (`image_attacks.py:513-519`), fast before slow to match `graphs.video_hooks`."""
    if "i3d" in model_type:
        return [model.res_layers._modules["1"]]
    if "tpn" in model_type:
        return [model.layer2]
    return [model._modules["fast_res2"], model._modules["slow_res2"]]


def to_frames(t: torch.Tensor) -> torch.Tensor:
    """(b, C, T, H, W) -> the engine's frame-major (b*T, C, H, W)."""
    b, c, tt, h, w = t.shape
    return t.permute(0, 2, 1, 3, 4).reshape(b * tt, c, h, w).contiguous()


class StageClassifier(nn.Module):
    """A backbone of this file + global average pool + Linear, shaped like the gluoncv classifiers the reference's white-box attacks
    take (`attack.py:63-96`): the backbone's stages stay reachable under their gluoncv names ON THE CLASSIFIER itself
    (`model.res_layers`, `model.slow_res2`, `model.layer1` ...), which is where `base_attacks.TAP._find_target_layer` (:737-743) looks
    them up, and every ReLU keeps a qualified name ending in `.relu`, which is what `base_attacks.SGM` (:511-513) selects by."""

    def __init__(self, backbone: nn.Module, feat_channels: int, num_classes: int, seed: int = 0):
        super().__init__()
        for name, child in backbone.named_children():
            setattr(self, name, child)
        self._backbone_forward = backbone.forward
        self.fc = nn.Linear(feat_channels, num_classes)
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            self.fc.weight.copy_(torch.randn(self.fc.weight.shape, generator=gen) * 0.05)
            self.fc.bias.copy_(torch.randn(self.fc.bias.shape, generator=gen) * 0.05)

    def forward(self, x):
        f = self._backbone_forward(x)
        feats = f if isinstance(f, tuple) else (f,)
        return self.fc(torch.cat([t.mean(dim=(2, 3, 4)) for t in feats], dim=1))


def tiny_stage_classifier(model_type="i3d_plain_resnet50", thw=(8, 32, 32), num_classes=5, wseed=4):
    """The float32 classifier the SGM / TAP fixtures of oracle/make_golden.py and their tests attack (tiny widths, every stage)."""
    from i2v_amd import graphs, weights
    g = graphs.build_video_tiny(model_type, thw, full=True)
    back = load_weights(make(model_type, True, full=True), weights.synthetic_state_dict(g, wseed)).float()
    with torch.no_grad():
        c = back(torch.zeros(1, 3, *thw))
    c = sum(t.shape[1] for t in (c if isinstance(c, tuple) else (c,)))
    return StageClassifier(back, c, num_classes, wseed).eval()
