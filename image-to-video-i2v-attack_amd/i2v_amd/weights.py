"""Backbone weights in torchvision `state_dict` layout.

The reference downloads ImageNet checkpoints with `models.<arch>(pretrained=True)`
(`/root/reference/image_attacks.py:88-101`).  There is no network here, so the default is a
seeded synthetic initialiser: Kaiming-normal(fan_out) convolutions and
randomised BatchNorm statistics, so that BN folding is actually exercised.  A real checkpoint
(`torch.save(model.state_dict())` of the torchvision model) is picked up from
`$I2V_WEIGHTS_DIR/<arch>.pth` when present; without one the synthetic initialiser must be asked
for explicitly (`load_state_dict`).
"""
import os
import zlib
from typing import Dict

import torch

from .graphs import Graph

BN_EPS = 1e-5   # torchvision BatchNorm2d default, used by every ResNet BN


def _gen(seed: int, key: str) -> torch.Generator:
    """One generator per parameter, keyed by its state_dict name: the values do not depend on
    the order in which the graph lists its nodes."""
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63))
    return g


def synthetic_state_dict(graph: Graph, seed: int = 0) -> Dict[str, torch.Tensor]:
    sd = {}
    for nd in graph.nodes:
        if nd.op != "conv":
            continue
        fan_out = nd.cout * nd.kt * nd.kh * nd.kw
        std = (2.0 / fan_out) ** 0.5
        # Non-local blocks: gluoncv zero-initialises the BatchNorm behind W (the block starts as the identity) and a trained
        # block stays a correction to its input, with attention logits of moderate size.  Kaiming-sized theta / phi and a unit
        # BatchNorm make the un-normalised softmax one-hot and the whole network chaotic instead (two float32 evaluations of a
        # 4-step ILAF run then differ by percents): smaller embeddings, and a small BatchNorm gain below.
        nl = ".nonlocal_block." in nd.weight
        if nl and (".theta." in nd.weight or ".phi." in nd.weight):
            std *= 0.1
        shape = (nd.cout, nd.cin, nd.kt, nd.kh, nd.kw) if graph.video else (nd.cout, nd.cin, nd.kh, nd.kw)
        sd[nd.weight] = torch.randn(*shape, generator=_gen(seed, nd.weight)) * std
        if nd.bias:
            sd[nd.bias] = torch.randn(nd.cout, generator=_gen(seed, nd.bias)) * 0.05
        if nd.bn:
            g = _gen(seed, nd.bn)
            sd[nd.bn + ".weight"] = (torch.rand(nd.cout, generator=g) + 0.5) * (0.1 if nl else 1.0)
            sd[nd.bn + ".bias"] = torch.randn(nd.cout, generator=g) * 0.1
            sd[nd.bn + ".running_mean"] = torch.randn(nd.cout, generator=g) * 0.1
            sd[nd.bn + ".running_var"] = torch.rand(nd.cout, generator=g) + 0.5
        if nd.pre_bn:
            g = _gen(seed, nd.pre_bn)
            sd[nd.pre_bn + ".weight"] = torch.rand(nd.cin, generator=g) + 0.5
            sd[nd.pre_bn + ".bias"] = torch.randn(nd.cin, generator=g) * 0.1
            sd[nd.pre_bn + ".running_mean"] = torch.randn(nd.cin, generator=g) * 0.1
            sd[nd.pre_bn + ".running_var"] = torch.rand(nd.cin, generator=g) + 0.5
    return sd


class MissingWeights(FileNotFoundError):
    pass


#: classifier-head parameters (torchvision and gluoncv both name the last Linear `fc`); not part of any graph's
#: `param_shapes()`, kept next to the backbone's tensors whenever the checkpoint has them
HEAD_KEYS = ("fc.weight", "fc.bias")


#: arch -> where its weights came from in this process ("<path>" or "synthetic(seed=N)"); printed once per arch
SOURCES: Dict[str, str] = {}


def synthetic_allowed() -> bool:
    return os.environ.get("I2V_SYNTHETIC_WEIGHTS", "") not in ("", "0")


def load_state_dict(graph: Graph, seed=None, keep_head: bool = False) -> Dict[str, torch.Tensor]:
    """`$I2V_WEIGHTS_DIR/<arch>.pth` (a torchvision-layout `state_dict`; extra keys such as `layer4.*`, `fc.*`,
    `num_batches_tracked` are ignored; with `keep_head` the classifier's `fc.weight` / `fc.bias` are returned as well) when it exists.  Otherwise the seeded synthetic initialiser -- but only when
    the caller asked for it: an explicit integer `seed` (tests, bench) or `I2V_SYNTHETIC_WEIGHTS=1`.  The reference
    attacks ImageNet-pretrained backbones (`pretrained=True`, image_attacks.py:88-101); silently perturbing clips
    against random weights would produce valid-looking but meaningless `*-adv.npy` files, so that case raises."""
    root = os.environ.get("I2V_WEIGHTS_DIR", "")
    path = os.path.join(root, graph.arch + ".pth") if root else ""
    if path and os.path.exists(path):
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
        shapes = graph.param_shapes()
        missing = [k for k in shapes if k not in sd]
        if missing:
            raise KeyError(f"{path}: missing keys {missing[:4]}...")
        for k, shp in shapes.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{path}: {k} has shape {tuple(sd[k].shape)}, expected {shp}")
        _note(graph.arch, path)
        out = {k: sd[k].float().contiguous() for k in shapes}
        for k in HEAD_KEYS if keep_head else ():      # the classifier head travels with its backbone (video.VideoModel.head_weights)
            if k in sd:
                out[k] = sd[k].float().contiguous()
        return out
    explicit = seed is not None
    if seed is None:
        if not synthetic_allowed():
            raise MissingWeights(
                f"no pretrained weights for {graph.arch!r}: put the torchvision state_dict at "
                f"$I2V_WEIGHTS_DIR/{graph.arch}.pth (I2V_WEIGHTS_DIR={root!r}), or opt in to seeded SYNTHETIC weights "
                "with I2V_SYNTHETIC_WEIGHTS=1 / --synthetic_weights / an explicit weight_seed")
        seed = 0
    _note(graph.arch, f"synthetic(seed={seed})", warn=not explicit)
    return synthetic_state_dict(graph, seed)


def _note(arch: str, source: str, warn: bool = True):
    if SOURCES.get(arch) != source:
        SOURCES[arch] = source
        if source.startswith("synthetic") and warn and os.environ.get("I2V_QUIET_WEIGHTS", "") in ("", "0"):
            import sys
            print(f"[i2v_amd] WARNING: backbone {arch} runs on {source} weights, not an ImageNet checkpoint", file=sys.stderr)
        elif not source.startswith("synthetic"):
            print(f"[i2v_amd] backbone {arch}: weights from {source}")


def convert_gluoncv_state_dict(graph: Graph, sd: Dict[str, torch.Tensor], rules=None) -> Dict[str, torch.Tensor]:
    """Map a gluoncv-torch video checkpoint (the reference builds its white-box models with
    `gluoncv.torch.model_zoo.get_model(cfg)`, `image_fine_tune_attack.py:58-66`) onto the key layout of the video graph IR
    (`graphs.i3d_resnet` / `graphs.slowfast_res2`): unwraps `{'state_dict': ...}`, strips a DataParallel `module.` prefix,
    applies `rules` -- `(regex, replacement)` pairs, first match wins -- and then REQUIRES every parameter the graph reads
    to be present with the right shape.  Dropped: `num_batches_tracked`, and parameters of stages the graph does not
    contain at all (later stages when the graph stops at a hook).  The classifier head (`fc.*`) is kept.  NOT dropped:
    a checkpoint parameter that lives INSIDE a stage the graph builds but that the graph does not read (e.g. the
    `res_layers.1.*` non-local block of an `i3d_nl5` checkpoint offered to a plain I3D graph) -- the checkpoint then
    describes another network than the graph and loading it "successfully" would attack the wrong model: KeyError.

    The graph's own names follow the attributes the reference dereferences on those models (`res_layers`, `slow_res2`,
    `fast_res2`, `image_attacks.py:513-519`) and the Bottleneck layout of gluoncv's action-recognition ResNets
    (`conv1/bn1/conv2/bn2/conv3/bn3/downsample.{0,1}`), so the default rule set is the identity.  gluoncv is not
    installed and cannot be fetched here: the mapping is therefore UNPINNED against a real checkpoint -- which is why a
    key that does not land, or lands with another shape, is an error and never a silent re-initialisation."""
    import re
    if isinstance(sd, dict) and isinstance(sd.get("state_dict"), dict):
        sd = sd["state_dict"]
    rules = [(re.compile(a), b) for a, b in (rules or [])]
    renamed = {}
    for k, v in sd.items():
        k = k[len("module."):] if k.startswith("module.") else k
        for rx, rep in rules:
            if rx.search(k):
                k = rx.sub(rep, k)
                break
        renamed[k] = v
    out, missing, wrong = {}, [], []
    shapes = graph.param_shapes()
    for k, shp in shapes.items():
        if k not in renamed:
            missing.append(k)
        elif tuple(renamed[k].shape) != tuple(shp):
            wrong.append((k, tuple(renamed[k].shape), tuple(shp)))
        else:
            out[k] = renamed[k].float().contiguous()
    if missing or wrong:
        raise KeyError(f"checkpoint does not cover the {graph.arch} graph: {len(missing)} missing (e.g. {missing[:3]}), "
                       f"{len(wrong)} with another shape (e.g. {wrong[:2]}); pass `rules=[(regex, replacement), ...]` to rename")
    stages = {_stage(k) for k in shapes}
    unread = sorted(k for k in renamed if k not in shapes and k not in HEAD_KEYS and not k.endswith("num_batches_tracked")
                    and _stage(k) in stages)
    if unread:
        raise KeyError(f"checkpoint has {len(unread)} parameters inside stages the {graph.arch} graph builds but does not read "
                       f"(e.g. {unread[:3]}): it describes a different network (non-local blocks?) than this graph")
    for k in HEAD_KEYS:
        if k in renamed:
            out[k] = renamed[k].float().contiguous()
    return out


def _stage(key: str) -> str:
    """The stage a parameter belongs to: its first path component, plus the second when that is an index
    (`res_layers.1.0.conv1.weight` -> `res_layers.1`, `slow_res2.0.conv1.weight` -> `slow_res2`, `first_stage.conv1.weight`
    -> `first_stage`)."""
    parts = key.split(".")
    return ".".join(parts[:2]) if len(parts) > 2 and parts[1].isdigit() else parts[0]


def fold_affine(nd, sd):
    """Per-output-channel (scale, shift) such that the node computes
    `y = conv(x, W) * scale + shift` -- BatchNorm in eval mode
    (`image_attacks.py:253-256` freezes every BN) or the conv bias."""
    if nd.bn:
        g = sd[nd.bn + ".weight"].double()
        b = sd[nd.bn + ".bias"].double()
        m = sd[nd.bn + ".running_mean"].double()
        v = sd[nd.bn + ".running_var"].double()
        s = g / torch.sqrt(v + BN_EPS)
        t = b - m * s
        if nd.bias:                 # conv WITH bias followed by BatchNorm (the non-local block's W): BN(conv + bias)
            t = t + sd[nd.bias].double() * s
        return s.float(), t.float()
    s = torch.ones(nd.cout)
    t = sd[nd.bias].float() if nd.bias else torch.zeros(nd.cout)
    return s, t


def fold_pre_affine(nd, sd):
    """(scale, shift) of the eval-mode BatchNorm in FRONT of a pre-activation conv: the conv reads
    relu(x*scale + shift) (torchvision `_DenseLayer`: norm1 -> relu1 -> conv1)."""
    g = sd[nd.pre_bn + ".weight"].double()
    b = sd[nd.pre_bn + ".bias"].double()
    m = sd[nd.pre_bn + ".running_mean"].double()
    v = sd[nd.pre_bn + ".running_var"].double()
    s = g / torch.sqrt(v + BN_EPS)
    return s.float(), (b - m * s).float()
