"""TEST INFRASTRUCTURE -- not part of the product path.

Torch-only `nn.Module` restatements of the torchvision 0.10.1 backbones the reference
constructs (`/root/reference/image_attacks.py:88-101`, pinned by `I2V_attack-env.yml:127`).
torchvision is not installed here and is third-party to the reference, so the public
architectures are restated with torchvision's attribute names (`conv1`, `bn1`, `layer1..4`,
`features[i]`, `Fire.expand3x3_activation`) -- exactly what the reference's
`_find_target_layer` (`image_attacks.py:260-271`) dereferences -- and torchvision's
`state_dict` keys.  They are handed to the *imported reference classes* through the
`torchvision.models` shim (`oracle/ref_shim.py`) and are written independently of the
product's graph IR (`i2v_amd/graphs.py`), so an IR mistake shows up as a parity failure.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # v1.5: stride on 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, layers, width=64, num_classes=10):
        super().__init__()
        self.inplanes = width
        self.conv1 = nn.Conv2d(3, width, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(width, layers[0], 1)
        self.layer2 = self._make_layer(width * 2, layers[1], 2)
        self.layer3 = self._make_layer(width * 4, layers[2], 2)
        self.layer4 = self._make_layer(width * 8, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(width * 8 * 4, num_classes)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class VGG(nn.Module):
    def __init__(self, cfg, num_classes=10):
        super().__init__()
        layers, c = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((2, 2))
        self.classifier = nn.Sequential(nn.Linear(c * 4, 16), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(16, num_classes))

    def forward(self, x):
        x = self.avgpool(self.features(x))
        return self.classifier(torch.flatten(x, 1))


class AlexNet(nn.Module):
    def __init__(self, c=(64, 192, 384, 256, 256), num_classes=10):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, c[0], 11, 4, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(c[0], c[1], 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(c[1], c[2], 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(c[2], c[3], 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(c[3], c[4], 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2))
        self.avgpool = nn.AdaptiveAvgPool2d((2, 2))
        self.classifier = nn.Sequential(nn.Dropout(), nn.Linear(c[4] * 4, 16), nn.ReLU(True),
                                        nn.Linear(16, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


class Fire(nn.Module):
    def __init__(self, inplanes, squeeze, e1, e3):
        super().__init__()
        self.squeeze = nn.Conv2d(inplanes, squeeze, 1)
        self.squeeze_activation = nn.ReLU(inplace=True)
        self.expand1x1 = nn.Conv2d(squeeze, e1, 1)
        self.expand1x1_activation = nn.ReLU(inplace=True)
        self.expand3x3 = nn.Conv2d(squeeze, e3, 3, padding=1)
        self.expand3x3_activation = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.squeeze_activation(self.squeeze(x))
        return torch.cat([self.expand1x1_activation(self.expand1x1(x)),
                          self.expand3x3_activation(self.expand3x3(x))], 1)


class SqueezeNet11(nn.Module):
    def __init__(self, div=1, num_classes=10):
        super().__init__()

        def ch(v):
            return max(4, v // div)
        self.features = nn.Sequential(
            nn.Conv2d(3, ch(64), 3, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(ch(64), ch(16), ch(64), ch(64)), Fire(2 * ch(64), ch(16), ch(64), ch(64)),
            nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(2 * ch(64), ch(32), ch(128), ch(128)), Fire(2 * ch(128), ch(32), ch(128), ch(128)),
            nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(2 * ch(128), ch(48), ch(192), ch(192)), Fire(2 * ch(192), ch(48), ch(192), ch(192)),
            Fire(2 * ch(192), ch(64), ch(256), ch(256)), Fire(2 * ch(256), ch(64), ch(256), ch(256)))
        self.classifier = nn.Sequential(nn.Dropout(), nn.Conv2d(2 * ch(256), num_classes, 1),
                                        nn.ReLU(inplace=True), nn.AdaptiveAvgPool2d((1, 1)))

    def forward(self, x):
        return torch.flatten(self.classifier(self.features(x)), 1)


class _DenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1)
        return self.conv2(self.relu2(self.norm2(self.conv1(self.relu1(self.norm1(x))))))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, nlayers, cin, bn_size, growth):
        super().__init__()
        for i in range(nlayers):
            self.add_module(f"denselayer{i + 1}", _DenseLayer(cin + i * growth, growth, bn_size))

    def forward(self, x):
        feats = [x]
        for _, layer in self.items():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class DenseNet(nn.Module):
    """torchvision DenseNet-BC: features.{conv0,norm0,relu0,pool0,denseblock{i},transition{i},norm5}."""

    def __init__(self, growth=32, block_config=(6, 12, 24, 16), init_features=64, bn_size=4, num_classes=10):
        super().__init__()
        from collections import OrderedDict
        self.features = nn.Sequential(OrderedDict([
            ("conv0", nn.Conv2d(3, init_features, 7, 2, 3, bias=False)), ("norm0", nn.BatchNorm2d(init_features)),
            ("relu0", nn.ReLU(inplace=True)), ("pool0", nn.MaxPool2d(3, 2, 1))]))
        c = init_features
        for i, n in enumerate(block_config):
            self.features.add_module(f"denseblock{i + 1}", _DenseBlock(n, c, bn_size, growth))
            c += n * growth
            if i != len(block_config) - 1:
                self.features.add_module(f"transition{i + 1}", nn.Sequential(OrderedDict([
                    ("norm", nn.BatchNorm2d(c)), ("relu", nn.ReLU(inplace=True)),
                    ("conv", nn.Conv2d(c, c // 2, 1, bias=False)), ("pool", nn.AvgPool2d(2, 2))])))
                c //= 2
        self.features.add_module("norm5", nn.BatchNorm2d(c))
        self.classifier = nn.Linear(c, num_classes)

    def forward(self, x):
        f = torch.relu(self.features(x))
        return self.classifier(torch.flatten(nn.functional.adaptive_avg_pool2d(f, (1, 1)), 1))


def make(model_name: str, tiny: bool) -> nn.Module:
    """Same vocabulary as the reference's `get_model` (`image_attacks.py:84-108`); the
    tiny variants mirror `i2v_amd.graphs.build_tiny` parameter for parameter."""
    if model_name in ("resnet", "resnet101"):
        return ResNet((2, 1, 2, 1), 8) if tiny else ResNet((3, 4, 23, 3), 64, 1000)
    if model_name == "resnet50":
        return ResNet((2, 1, 2, 1), 8) if tiny else ResNet((3, 4, 6, 3), 64, 1000)
    if model_name in ("vgg", "vgg16"):
        if tiny:
            return VGG((8, 8, "M", 16, 16, "M", 16, 16, 16, "M", 32, 32, 32, "M", 32, 32, 32, "M"))
        return VGG((64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M",
                    512, 512, 512, "M"), 1000)
    if model_name == "alexnet":
        return AlexNet((8, 24, 48, 32, 32)) if tiny else AlexNet(num_classes=1000)
    if model_name in ("squeezenet", "squeezenet1_1"):
        return SqueezeNet11(4) if tiny else SqueezeNet11(1, 1000)
    if model_name == "densenet121":
        return DenseNet(8, (2, 3, 2, 2), 16, 2) if tiny else DenseNet(32, (6, 12, 24, 16), 64, 4, 1000)
    if model_name == "densenet161":
        return DenseNet(8, (2, 3, 2, 2), 16, 2) if tiny else DenseNet(48, (6, 12, 36, 24), 96, 4, 1000)
    raise KeyError(model_name)


def load_backbone_weights(model: nn.Module, sd: dict) -> nn.Module:
    """Load the (truncated) backbone `state_dict` produced by `i2v_amd.weights`; parameters
    behind the deepest hook keep their torch default init (they cannot affect the loss)."""
    own = model.state_dict()
    for k, v in sd.items():
        assert k in own and own[k].shape == v.shape, (k, tuple(v.shape))
        own[k].copy_(v)
    return model
