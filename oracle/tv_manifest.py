"""TEST INFRASTRUCTURE -- key / shape manifest of the torchvision 0.10.1 `state_dict`s the reference loads with
`models.<arch>(pretrained=True)` (`/root/reference/image_attacks.py:88-101`, `TPAMI_attack.py:104-117`; version pinned by
`I2V_attack-env.yml:127`).

torchvision is not installed here, so the manifest is RESTATED from the public model definitions (torchvision/models/
{resnet,vgg,alexnet,squeezenet,densenet}.py at 0.10.1) with explicit loops written independently of the product's graph
IR (`i2v_amd/graphs.py`): `tests/test_weights_manifest.py` checks that every parameter the IR asks for exists here with the
same shape, and that `weights.load_state_dict` digests a checkpoint with EXACTLY these keys (incl. `layer4.*`, `fc.*`,
`classifier.*`, `num_batches_tracked`).  Unpinned against a real torchvision install (none available offline).

    python -m oracle.tv_manifest        # rewrites tests/golden/torchvision_0_10_1_state_dict_keys.json
"""
import json
import os


def _bn(out, p, c):
    for s in ("weight", "bias", "running_mean", "running_var"):
        out[f"{p}.{s}"] = [c]
    out[f"{p}.num_batches_tracked"] = []


def resnet(layers):
    out = {"conv1.weight": [64, 3, 7, 7]}
    _bn(out, "bn1", 64)
    inplanes = 64
    for li, n in enumerate(layers):
        planes = 64 * 2 ** li
        for b in range(n):
            p = f"layer{li + 1}.{b}"
            stride = 2 if (b == 0 and li > 0) else 1
            out[f"{p}.conv1.weight"] = [planes, inplanes, 1, 1]
            _bn(out, f"{p}.bn1", planes)
            out[f"{p}.conv2.weight"] = [planes, planes, 3, 3]
            _bn(out, f"{p}.bn2", planes)
            out[f"{p}.conv3.weight"] = [planes * 4, planes, 1, 1]
            _bn(out, f"{p}.bn3", planes * 4)
            if stride != 1 or inplanes != planes * 4:
                out[f"{p}.downsample.0.weight"] = [planes * 4, inplanes, 1, 1]
                _bn(out, f"{p}.downsample.1", planes * 4)
            inplanes = planes * 4
    out["fc.weight"] = [1000, 2048]
    out["fc.bias"] = [1000]
    return out


def vgg16():
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    out, c, idx = {}, 3, 0
    for v in cfg:
        if v == "M":
            idx += 1
        else:
            out[f"features.{idx}.weight"] = [v, c, 3, 3]
            out[f"features.{idx}.bias"] = [v]
            c, idx = v, idx + 2
    for i, (o, n) in zip((0, 3, 6), ((4096, 512 * 7 * 7), (4096, 4096), (1000, 4096))):
        out[f"classifier.{i}.weight"] = [o, n]
        out[f"classifier.{i}.bias"] = [o]
    return out


def alexnet():
    out = {}
    for i, shp in zip((0, 3, 6, 8, 10), ([64, 3, 11, 11], [192, 64, 5, 5], [384, 192, 3, 3], [256, 384, 3, 3], [256, 256, 3, 3])):
        out[f"features.{i}.weight"] = shp
        out[f"features.{i}.bias"] = [shp[0]]
    for i, (o, n) in zip((1, 4, 6), ((4096, 256 * 6 * 6), (4096, 4096), (1000, 4096))):
        out[f"classifier.{i}.weight"] = [o, n]
        out[f"classifier.{i}.bias"] = [o]
    return out


def squeezenet1_1():
    out = {"features.0.weight": [64, 3, 3, 3], "features.0.bias": [64]}
    fires = {3: (64, 16, 64, 64), 4: (128, 16, 64, 64), 6: (128, 32, 128, 128), 7: (256, 32, 128, 128),
             9: (256, 48, 192, 192), 10: (384, 48, 192, 192), 11: (384, 64, 256, 256), 12: (512, 64, 256, 256)}
    for i, (cin, sq, e1, e3) in fires.items():
        p = f"features.{i}"
        out[f"{p}.squeeze.weight"], out[f"{p}.squeeze.bias"] = [sq, cin, 1, 1], [sq]
        out[f"{p}.expand1x1.weight"], out[f"{p}.expand1x1.bias"] = [e1, sq, 1, 1], [e1]
        out[f"{p}.expand3x3.weight"], out[f"{p}.expand3x3.bias"] = [e3, sq, 3, 3], [e3]
    out["classifier.1.weight"], out["classifier.1.bias"] = [1000, 512, 1, 1], [1000]
    return out


def densenet(growth, blocks, init, bn_size=4):
    out = {"features.conv0.weight": [init, 3, 7, 7]}
    _bn(out, "features.norm0", init)
    c = init
    for bi, n in enumerate(blocks):
        for li in range(n):
            p = f"features.denseblock{bi + 1}.denselayer{li + 1}"
            _bn(out, f"{p}.norm1", c)
            out[f"{p}.conv1.weight"] = [bn_size * growth, c, 1, 1]
            _bn(out, f"{p}.norm2", bn_size * growth)
            out[f"{p}.conv2.weight"] = [growth, bn_size * growth, 3, 3]
            c += growth
        if bi != len(blocks) - 1:
            p = f"features.transition{bi + 1}"
            _bn(out, f"{p}.norm", c)
            out[f"{p}.conv.weight"] = [c // 2, c, 1, 1]
            c //= 2
    _bn(out, "features.norm5", c)
    out["classifier.weight"], out["classifier.bias"] = [1000, c], [1000]
    return out


def manifest():
    return {"resnet50": resnet((3, 4, 6, 3)), "resnet101": resnet((3, 4, 23, 3)), "vgg16": vgg16(), "alexnet": alexnet(),
            "squeezenet1_1": squeezenet1_1(), "densenet121": densenet(32, (6, 12, 24, 16), 64),
            "densenet161": densenet(48, (6, 12, 36, 24), 96)}


PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                    "torchvision_0_10_1_state_dict_keys.json")

if __name__ == "__main__":
    m = manifest()
    with open(PATH, "w") as fh:
        json.dump(m, fh, separators=(",", ":"))
    for k, v in m.items():
        n = sum(int(__import__("math").prod(s)) for s in v.values())
        print(k, len(v), "keys", round(n / 1e6, 2), "M values")
