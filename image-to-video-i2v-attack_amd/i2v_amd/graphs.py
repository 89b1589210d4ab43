"""Architecture IR for the image backbones the attack classes hook.

The reference obtains its backbones from torchvision 0.10.1
(`/root/reference/image_attacks.py:84-108`, `TPAMI_attack.py:100-124`); torchvision
is third-party and not vendored, so the graphs are restated here from the public
architecture definitions as a flat list of nodes.  The same IR is consumed by

  * the HIP engine (`engine.py` -> `i2v_net_add_*` of the C-ABI), and
  * the CPU oracle (`oracle/restate.py`) -- test infrastructure only.

Only what lies at or before a hook layer is ever executed; everything behind the
deepest hook (layer4 / avgpool / fc / classifier tails) cannot influence the loss
(`image_attacks.py:336-347`) and is never emitted.

Tensor ids index `Graph.tensors`.  A tensor is a channel-slice *view* of a buffer
(`buf`, `c_off`, buffer width `Graph.buffers[buf]`) so that SqueezeNet Fire
concatenation needs no copy.

Weight keys follow the torchvision `state_dict` layout, so a real ImageNet
checkpoint loads unchanged when one is supplied.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


@dataclass
class TensorSpec:
    C: int
    H: int
    W: int
    buf: int            # buffer id
    c_off: int          # channel offset inside the buffer
    post_relu: bool     # values are outputs of a ReLU (gradient mask = value > 0)
    name: str = ""
    T: int = 1          # frames per clip (video backbones; 1 for image backbones)


@dataclass
class ConvNode:
    src: int
    dst: int
    cin: int
    cout: int
    kh: int
    kw: int
    stride: int
    pad: int
    weight: str                 # state_dict key of the conv weight
    bias: Optional[str]         # state_dict key of the conv bias (VGG/AlexNet/SqueezeNet)
    bn: Optional[str]           # state_dict prefix of the BatchNorm2d that follows (ResNet)
    relu: bool                  # ReLU applied after (bias/bn [+ residual])
    residual: Optional[int] = None   # tensor added before the ReLU (Bottleneck identity)
    pre_bn: Optional[str] = None     # DenseNet pre-activation: relu(BatchNorm(x)) applied to the conv INPUT
    op: str = "conv"
    # video backbones: temporal extent of the kernel (output frame t reads input frames t*stride_t - pad_t + q*dil_t)
    kt: int = 1
    stride_t: int = 1
    pad_t: int = 0
    dil_t: int = 1


@dataclass
class PoolNode:
    src: int
    dst: int
    k: int
    stride: int
    pad: int
    ceil_mode: bool
    op: str = "maxpool"
    kt: int = 1
    stride_t: int = 1
    pad_t: int = 0


@dataclass
class AttnNode:
    """Core of a non-local block: dst[c][i] = sum_j g[c][j] softmax_j(scale * <theta[:, i], phi[:, j]>) per clip (`i2v_net_add_attention`)."""
    src: int                    # theta
    phi: int
    g: int
    dst: int
    scale: float = 1.0
    op: str = "attention"

    @property
    def extra_srcs(self):
        return (self.phi, self.g)


@dataclass
class Graph:
    arch: str
    in_hw: Tuple[int, int]
    tensors: List[TensorSpec] = field(default_factory=list)
    buffers: List[int] = field(default_factory=list)          # channel width per buffer
    nodes: list = field(default_factory=list)
    hooks: Dict[int, int] = field(default_factory=dict)       # depth (1..4) -> tensor id
    # depth -> tensor id of the WHOLE hooked module's output where that differs from `hooks[depth]`: the list-depth
    # lookup of the adaptive attack hooks the Fire module itself (TPAMI_attack.py:195-197), i.e. cat(expand1x1,
    # expand3x3), while the scalar-depth lookup takes `.expand3x3_activation` only (:199, image_attacks.py:269-271)
    hooks_module: Dict[int, int] = field(default_factory=dict)
    input: int = 0
    video: bool = False                                       # 3-D backbone: conv weights are (cout,cin,kt,kh,kw)
    classifier_feats: List[int] = field(default_factory=list)  # tensors a classifier head pools and concatenates (video graphs built to their last stage)

    # -- construction helpers -------------------------------------------------
    def new_buffer(self, C: int) -> int:
        self.buffers.append(C)
        return len(self.buffers) - 1

    def new_tensor(self, C, H, W, post_relu, name="", buf=None, c_off=0, T=1) -> int:
        if buf is None:
            buf = self.new_buffer(C)
        self.tensors.append(TensorSpec(C, H, W, buf, c_off, post_relu, name, T))
        return len(self.tensors) - 1

    def conv(self, src, cout, k, stride, pad, weight, bias=None, bn=None, relu=True,
             residual=None, name="", dst_buf=None, dst_c_off=0, pre_bn=None) -> int:
        s = self.tensors[src]
        Ho = (s.H + 2 * pad - k) // stride + 1
        Wo = (s.W + 2 * pad - k) // stride + 1
        dst = self.new_tensor(cout, Ho, Wo, relu, name, dst_buf, dst_c_off)
        assert pre_bn is None or (k == 1 and stride == 1 and pad == 0), "pre-activation only on 1x1 convs"
        self.nodes.append(ConvNode(src, dst, s.C, cout, k, k, stride, pad, weight, bias, bn,
                                   relu, residual, pre_bn))
        return dst

    def maxpool(self, src, k, stride, pad=0, ceil_mode=False, name="", dst_buf=None, dst_c_off=0, op="maxpool") -> int:
        s = self.tensors[src]

        def out(n):
            if ceil_mode:
                o = -(-(n + 2 * pad - k) // stride) + 1
                if (o - 1) * stride >= n + pad:     # last window must start inside the input
                    o -= 1
            else:
                o = (n + 2 * pad - k) // stride + 1
            return o
        dst = self.new_tensor(s.C, out(s.H), out(s.W), False, name, dst_buf, dst_c_off)
        self.nodes.append(PoolNode(src, dst, k, stride, pad, ceil_mode, op))
        return dst

    def avgpool(self, src, k, stride, name="", dst_buf=None, dst_c_off=0) -> int:
        return self.maxpool(src, k, stride, 0, False, name, dst_buf, dst_c_off, op="avgpool")

    def conv3d(self, src, cout, k, stride, pad, weight, bn=None, relu=True, residual=None, name="",
               dst_buf=None, dst_c_off=0, dil_t=1, bias=None) -> int:
        """k / stride / pad are (temporal, spatial) pairs, as in `nn.Conv3d((kt,k,k), (st,s,s), (pt,p,p))`."""
        (kt, ks), (st, ss), (pt, ps) = k, stride, pad
        s = self.tensors[src]
        To = (s.T + 2 * pt - dil_t * (kt - 1) - 1) // st + 1
        Ho = (s.H + 2 * ps - ks) // ss + 1
        Wo = (s.W + 2 * ps - ks) // ss + 1
        dst = self.new_tensor(cout, Ho, Wo, relu, name, dst_buf, dst_c_off, T=To)
        self.nodes.append(ConvNode(src, dst, s.C, cout, ks, ks, ss, ps, weight, bias, bn, relu, residual, None,
                                   "conv", kt, st, pt, dil_t))
        return dst

    def attention(self, theta, phi, gg, name="", scale=1.0) -> int:
        t = self.tensors[theta]
        assert t.C == self.tensors[phi].C == self.tensors[gg].C
        dst = self.new_tensor(t.C, t.H, t.W, False, name, T=t.T)
        self.nodes.append(AttnNode(theta, phi, gg, dst, scale))
        return dst

    def nonlocal_block(self, x, prefix) -> int:
        """gluoncv / mmaction `NonLocalModule` (embedded gaussian, sub-sampled, with BatchNorm): theta = conv1x1x1(x); phi, g =
        conv1x1x1(maxpool 1x2x2 (x)); y = g softmax(theta^T phi)^T; z = BN(W y) + x.  Node order matters to the planner: a
        convolution (theta) must be the LAST contributor to x's gradient, i.e. the first consumer listed."""
        c = self.tensors[x].C
        e = c // 2
        theta = self.conv3d(x, e, (1, 1), (1, 1), (0, 0), f"{prefix}.theta.weight", relu=False, name=f"{prefix}.theta", bias=f"{prefix}.theta.bias")
        xp = self.maxpool3d(x, (1, 2), (1, 2), (0, 0), name=f"{prefix}.max_pool")
        phi = self.conv3d(xp, e, (1, 1), (1, 1), (0, 0), f"{prefix}.phi.1.weight", relu=False, name=f"{prefix}.phi", bias=f"{prefix}.phi.1.bias")
        gg = self.conv3d(xp, e, (1, 1), (1, 1), (0, 0), f"{prefix}.g.1.weight", relu=False, name=f"{prefix}.g", bias=f"{prefix}.g.1.bias")
        y = self.attention(theta, phi, gg, name=f"{prefix}.y")
        return self.conv3d(y, c, (1, 1), (1, 1), (0, 0), f"{prefix}.W.0.weight", bn=f"{prefix}.W.1", relu=False, residual=x,
                           name=f"{prefix}.out", bias=f"{prefix}.W.0.bias")

    def maxpool3d(self, src, k, stride, pad=(0, 0), name="") -> int:
        (kt, ks), (st, ss), (pt, ps) = k, stride, pad
        s = self.tensors[src]
        dst = self.new_tensor(s.C, (s.H + 2 * ps - ks) // ss + 1, (s.W + 2 * ps - ks) // ss + 1, False, name,
                              T=(s.T + 2 * pt - kt) // st + 1)
        self.nodes.append(PoolNode(src, dst, ks, ss, ps, False, "maxpool", kt, st, pt))
        return dst

    # -- analysis -------------------------------------------------------------
    def truncated(self, hook_tensors: List[int]) -> "Graph":
        """Copy of the graph holding only the nodes some hook tensor depends on."""
        needed_bufs = set()
        keep = [False] * len(self.nodes)
        needed = set(hook_tensors)
        for i in range(len(self.nodes) - 1, -1, -1):
            nd = self.nodes[i]
            d = self.tensors[nd.dst]
            hit = nd.dst in needed or any(
                self.tensors[t].buf == d.buf and _overlap(self.tensors[t], d) for t in needed)
            if hit:
                keep[i] = True
                needed.add(nd.src)
                needed.update(getattr(nd, "extra_srcs", ()))
                if getattr(nd, "residual", None) is not None:
                    needed.add(nd.residual)
        g = Graph(self.arch, self.in_hw, self.tensors, self.buffers,
                  [n for n, k in zip(self.nodes, keep) if k], dict(self.hooks), dict(self.hooks_module), self.input,
                  self.video, list(self.classifier_feats))
        return g

    def hook_for(self, depth: int, whole_module: bool = False) -> int:
        """Hooked tensor of `depth`; `whole_module` selects the list-depth lookup of `AENS_I2V_MF`
        (`/root/reference/TPAMI_attack.py:176-200`), which differs from the scalar one for SqueezeNet only."""
        if depth not in self.hooks:
            raise KeyError(depth)
        return self.hooks_module.get(depth, self.hooks[depth]) if whole_module else self.hooks[depth]

    def macs_per_frame(self) -> int:
        """Multiply-adds of one forward pass per input frame (image backbones) / per input CLIP (video)."""
        tot = 0
        for nd in self.nodes:
            if nd.op == "conv":
                d = self.tensors[nd.dst]
                tot += d.T * d.H * d.W * nd.cout * nd.cin * nd.kt * nd.kh * nd.kw
            elif nd.op == "attention":          # theta^T phi and g P^T
                t, k = self.tensors[nd.src], self.tensors[nd.phi]
                tot += 2 * t.C * (t.T * t.H * t.W) * (k.T * k.H * k.W)
        return tot

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """state_dict key -> shape for every parameter the nodes reference."""
        out = {}
        for nd in self.nodes:
            if nd.op != "conv":
                continue
            out[nd.weight] = (nd.cout, nd.cin, nd.kt, nd.kh, nd.kw) if self.video else (nd.cout, nd.cin, nd.kh, nd.kw)
            if nd.bias:
                out[nd.bias] = (nd.cout,)
            if nd.bn:
                for s in ("weight", "bias", "running_mean", "running_var"):
                    out[f"{nd.bn}.{s}"] = (nd.cout,)
            if nd.pre_bn:
                for s in ("weight", "bias", "running_mean", "running_var"):
                    out[f"{nd.pre_bn}.{s}"] = (nd.cin,)
        return out


def _overlap(a: TensorSpec, b: TensorSpec) -> bool:
    return a.c_off < b.c_off + b.C and b.c_off < a.c_off + a.C


# ---------------------------------------------------------------------------
# ResNet v1.5 Bottleneck nets (torchvision `resnet50` / `resnet101`)
# ---------------------------------------------------------------------------
def resnet(layers=(3, 4, 23, 3), width=64, in_hw=(224, 224), arch="resnet101") -> Graph:
    """Bottleneck ResNet.  Hook d = output of `layer{d}[-1]` (post-ReLU), as in
    `/root/reference/image_attacks.py:261-262`."""
    g = Graph(arch, in_hw)
    x = g.new_tensor(3, in_hw[0], in_hw[1], False, "input")
    g.input = x
    x = g.conv(x, width, 7, 2, 3, "conv1.weight", bn="bn1", relu=True, name="stem")
    x = g.maxpool(x, 3, 2, 1, name="maxpool")
    inplanes = width
    for li, nblocks in enumerate(layers):
        planes = width * (2 ** li)
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 0) else 1
            p = f"layer{li + 1}.{b}"
            a = g.conv(x, planes, 1, 1, 0, f"{p}.conv1.weight", bn=f"{p}.bn1", relu=True,
                       name=f"{p}.conv1")
            a = g.conv(a, planes, 3, stride, 1, f"{p}.conv2.weight", bn=f"{p}.bn2", relu=True,
                       name=f"{p}.conv2")
            # the projection shortcut is emitted AFTER conv1/conv2: in the reversed (gradient)
            # order its input-gradient is then a pending addend of conv1's, which finalises x
            if stride != 1 or inplanes != planes * 4:
                idt = g.conv(x, planes * 4, 1, stride, 0, f"{p}.downsample.0.weight",
                             bn=f"{p}.downsample.1", relu=False, name=f"{p}.downsample")
            else:
                idt = x
            x = g.conv(a, planes * 4, 1, 1, 0, f"{p}.conv3.weight", bn=f"{p}.bn3", relu=True,
                       residual=idt, name=f"{p}.out")
            inplanes = planes * 4
        g.hooks[li + 1] = x
    return g


# ---------------------------------------------------------------------------
# VGG-16 `features`
# ---------------------------------------------------------------------------
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


def vgg(cfg=VGG16_CFG, in_hw=(224, 224), arch="vgg16", hook_index=None) -> Graph:
    """Hook d = ReLU module `features[{1:1,2:11,3:20,4:29}[d]]`
    (`/root/reference/image_attacks.py:266-268`)."""
    g = Graph(arch, in_hw)
    x = g.new_tensor(3, in_hw[0], in_hw[1], False, "input")
    g.input = x
    idx = 0
    relu_at = {}
    for v in cfg:
        if v == "M":
            x = g.maxpool(x, 2, 2, 0, name=f"features.{idx}")
            idx += 1
        else:
            x = g.conv(x, v, 3, 1, 1, f"features.{idx}.weight", bias=f"features.{idx}.bias",
                       relu=True, name=f"features.{idx + 1}")
            relu_at[idx + 1] = x
            idx += 2
    hook_index = hook_index or {1: 1, 2: 11, 3: 20, 4: 29}
    for d, i in hook_index.items():
        if i in relu_at:
            g.hooks[d] = relu_at[i]
    return g


# ---------------------------------------------------------------------------
# AlexNet `features`
# ---------------------------------------------------------------------------
def alexnet(width_div=1, in_hw=(224, 224), arch="alexnet") -> Graph:
    """Hook d = ReLU module `features[{1:1,2:4,3:7,4:11}[d]]`
    (`/root/reference/image_attacks.py:263-265`)."""
    c = [max(4, v // width_div) for v in (64, 192, 384, 256, 256)]
    g = Graph(arch, in_hw)
    x = g.new_tensor(3, in_hw[0], in_hw[1], False, "input")
    g.input = x
    x = g.conv(x, c[0], 11, 4, 2, "features.0.weight", bias="features.0.bias", name="features.1")
    g.hooks[1] = x
    x = g.maxpool(x, 3, 2, 0, name="features.2")
    x = g.conv(x, c[1], 5, 1, 2, "features.3.weight", bias="features.3.bias", name="features.4")
    g.hooks[2] = x
    x = g.maxpool(x, 3, 2, 0, name="features.5")
    x = g.conv(x, c[2], 3, 1, 1, "features.6.weight", bias="features.6.bias", name="features.7")
    g.hooks[3] = x
    x = g.conv(x, c[3], 3, 1, 1, "features.8.weight", bias="features.8.bias", name="features.9")
    x = g.conv(x, c[4], 3, 1, 1, "features.10.weight", bias="features.10.bias", name="features.11")
    g.hooks[4] = x
    return g


# ---------------------------------------------------------------------------
# SqueezeNet 1.1 `features`
# ---------------------------------------------------------------------------
def squeezenet(width_div=1, in_hw=(224, 224), arch="squeezenet1_1") -> Graph:
    """Hook d = `features[{1:3,2:6,3:9,4:12}[d]].expand3x3_activation`, i.e. only the
    3x3 branch of the Fire module (`/root/reference/image_attacks.py:269-271`)."""
    def ch(v):
        return max(4, v // width_div)
    g = Graph(arch, in_hw)
    x = g.new_tensor(3, in_hw[0], in_hw[1], False, "input")
    g.input = x
    x = g.conv(x, ch(64), 3, 2, 0, "features.0.weight", bias="features.0.bias", name="features.1")
    x = g.maxpool(x, 3, 2, 0, ceil_mode=True, name="features.2")
    fires = {3: (16, 64, 64), 4: (16, 64, 64), 6: (32, 128, 128), 7: (32, 128, 128),
             9: (48, 192, 192), 10: (48, 192, 192), 11: (64, 256, 256), 12: (64, 256, 256)}
    hook_of = {3: 1, 6: 2, 9: 3, 12: 4}
    for idx in range(3, 13):
        if idx in (5, 8):
            x = g.maxpool(x, 3, 2, 0, ceil_mode=True, name=f"features.{idx}")
            continue
        sq, e1, e3 = (ch(v) for v in fires[idx])
        p = f"features.{idx}"
        s = g.conv(x, sq, 1, 1, 0, f"{p}.squeeze.weight", bias=f"{p}.squeeze.bias",
                   name=f"{p}.squeeze_activation")
        st = g.tensors[s]
        cat_buf = g.new_buffer(e1 + e3)
        g.conv(s, e1, 1, 1, 0, f"{p}.expand1x1.weight", bias=f"{p}.expand1x1.bias",
               name=f"{p}.expand1x1_activation", dst_buf=cat_buf, dst_c_off=0)
        t3 = g.conv(s, e3, 3, 1, 1, f"{p}.expand3x3.weight", bias=f"{p}.expand3x3.bias",
                    name=f"{p}.expand3x3_activation", dst_buf=cat_buf, dst_c_off=e1)
        x = g.new_tensor(e1 + e3, st.H, st.W, True, f"{p}.cat", buf=cat_buf, c_off=0)
        if idx in hook_of:
            g.hooks[hook_of[idx]] = t3
            g.hooks_module[hook_of[idx]] = x
    return g


# ---------------------------------------------------------------------------
# DenseNet-BC (torchvision `densenet121` / `densenet161`) -- extension, see `build`
# ---------------------------------------------------------------------------
def densenet(growth=32, block_config=(6, 12, 24, 16), init_features=64, bn_size=4, in_hw=(224, 224),
             arch="densenet121") -> Graph:
    """Hook d = output of `features.denseblock{d}` (the raw concatenation, not a ReLU output) -- the
    only hint the reference gives for DenseNet (`image_attacks.py:98-99`).  A dense block is ONE buffer:
    layer l reads its first C_l channels through BN+ReLU (pre-activation, folded into the 1x1 conv's
    operand read) and appends `growth` channels."""
    g = Graph(arch, in_hw)
    x = g.new_tensor(3, in_hw[0], in_hw[1], False, "input")
    g.input = x
    x = g.conv(x, init_features, 7, 2, 3, "features.conv0.weight", bn="features.norm0", relu=True, name="features.relu0")
    C = init_features
    pending_pool = ("max", x)
    for bi, nlayers in enumerate(block_config):
        total = C + nlayers * growth
        cb = g.new_buffer(total)
        kind, src = pending_pool
        if kind == "max":
            v = g.maxpool(src, 3, 2, 1, name="features.pool0", dst_buf=cb, dst_c_off=0)
        else:
            v = g.avgpool(src, 2, 2, name=f"features.transition{bi}.pool", dst_buf=cb, dst_c_off=0)
        H, W = g.tensors[v].H, g.tensors[v].W
        for li in range(nlayers):
            p = f"features.denseblock{bi + 1}.denselayer{li + 1}"
            view = g.new_tensor(C, H, W, False, f"{p}.in", buf=cb, c_off=0)
            a = g.conv(view, bn_size * growth, 1, 1, 0, f"{p}.conv1.weight", bn=f"{p}.norm2", relu=True,
                       name=f"{p}.relu2", pre_bn=f"{p}.norm1")
            g.conv(a, growth, 3, 1, 1, f"{p}.conv2.weight", relu=False, name=f"{p}.conv2", dst_buf=cb, dst_c_off=C)
            C += growth
        full = g.new_tensor(C, H, W, False, f"features.denseblock{bi + 1}", buf=cb, c_off=0)
        g.hooks[bi + 1] = full
        if bi != len(block_config) - 1:
            p = f"features.transition{bi + 1}"
            t = g.conv(full, C // 2, 1, 1, 0, f"{p}.conv.weight", relu=False, name=f"{p}.conv", pre_bn=f"{p}.norm")
            C //= 2
            pending_pool = ("avg", t)
    return g


# ---------------------------------------------------------------------------
# Video backbones: the white-box models ILAF fine-tunes against (`image_attacks.py:513-519`).
# The reference takes them from gluoncv 0.10.4 (`image_fine_tune_attack.py:63`, configs `utils.py:9-14`), which is
# not vendored and not installed; the graphs below restate the PUBLIC architectures (I3D: Carreira & Zisserman /
# Wang et al. "3x1x1" inflation; SlowFast: Feichtenhofer et al.) up to the hooked stage.  Parity with gluoncv's
# exact module layout and checkpoints is therefore UNPINNED; what is pinned is the arithmetic of every node
# against `torch.nn.functional.conv3d / max_pool3d` (oracle/video_models.py).  The non-local blocks of the `i3d_nl5_*`
# configs ARE built (`NL5_FREQ` below, the attention core behind `i2v_net_add_attention`): `'i3d_resnet50'` is that
# network, `'i3d_plain_resnet50'` the one without them.  Not built: TPN's pyramid neck and heads (its ResNet backbone is).
# ---------------------------------------------------------------------------
NL5_FREQ = ((0, 0, 0), (0, 1, 0, 1), (0, 1, 0, 1, 0, 1), (0, 0, 0))     # gluoncv i3d_nl5: non-local blocks behind blocks 1, 3 of res3 and 1, 3, 5 of res4


def i3d_resnet(layers=(3, 4, 6, 3), width=64, in_thw=(32, 224, 224), arch="i3d_resnet50",
               inflate=((1, 1, 1), (1, 0, 1, 0), (1, 0, 1, 0, 1, 0), (0, 1, 0)), nonlocal_freq=None) -> Graph:
    """Inflated Bottleneck ResNet: stem 5x7x7 / (2,2,2), max-pool 1x3x3 / (2,2,2), pool2 2x1x1 after the first
    stage, `inflate[stage][block] == 1` turns that block's first 1x1x1 convolution into 3x1x1.  `hooks[d]` is the
    output of stage d; the reference hooks `res_layers._modules['1']`, i.e. d = 2 (`image_attacks.py:514`).
    `nonlocal_freq[stage][block] == 1` appends a non-local block to that Bottleneck (`Graph.nonlocal_block`): the reference's I3D
    configurations are gluoncv's `i3d_nl5_resnet50/101_v1` (`utils.py:9-10`), NL5_FREQ; for the 101-layer net the pattern of
    res4 repeats over its 23 blocks' first six only (gluoncv lists five blocks in all)."""
    T, H, W = in_thw
    g = Graph(arch, (H, W), video=True)
    x = g.new_tensor(3, H, W, False, "input", T=T)
    g.input = x
    x = g.conv3d(x, width, (5, 7), (2, 2), (2, 3), "conv1.weight", bn="bn1", name="stem")
    x = g.maxpool3d(x, (1, 3), (2, 2), (0, 1), name="maxpool")
    inplanes = width
    for li, nblocks in enumerate(layers):
        planes = width * (2 ** li)
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 0) else 1
            p = f"res_layers.{li}.{b}"
            infl = inflate[li][b % len(inflate[li])] if li < len(inflate) else 0
            a = g.conv3d(x, planes, (3, 1) if infl else (1, 1), (1, 1), (1, 0) if infl else (0, 0),
                         f"{p}.conv1.weight", bn=f"{p}.bn1", name=f"{p}.conv1")
            a = g.conv3d(a, planes, (1, 3), (1, stride), (0, 1), f"{p}.conv2.weight", bn=f"{p}.bn2", name=f"{p}.conv2")
            if stride != 1 or inplanes != planes * 4:
                idt = g.conv3d(x, planes * 4, (1, 1), (1, stride), (0, 0), f"{p}.downsample.0.weight",
                               bn=f"{p}.downsample.1", relu=False, name=f"{p}.downsample")
            else:
                idt = x
            x = g.conv3d(a, planes * 4, (1, 1), (1, 1), (0, 0), f"{p}.conv3.weight", bn=f"{p}.bn3", residual=idt,
                         name=f"{p}.out")
            if nonlocal_freq is not None and li < len(nonlocal_freq) and b < len(nonlocal_freq[li]) and nonlocal_freq[li][b]:
                x = g.nonlocal_block(x, f"{p}.nonlocal_block")
            inplanes = planes * 4
        g.hooks[li + 1] = x
        if li == 0:
            x = g.maxpool3d(x, (2, 1), (2, 1), (0, 0), name="pool2")
    return g


def slowfast_res2(width=64, in_thw=(32, 224, 224), arch="slowfast_resnet50", slow_stride=8, fast_stride=1,
                  beta_inv=8, fusion_ratio=2, fusion_kernel=5, blocks=3) -> Graph:
    """SlowFast up to its `res2` stages, the two tensors the reference hooks (`image_attacks.py:515-516`):
    hooks[1] = slow_res2, hooks[2] = fast_res2, in the order their forward hooks fire (the fast pathway runs first).
    The pathway inputs `x[:, :, ::stride]` are expressed as temporal stride / dilation of the stem convolutions,
    so both stems read the same frame tensor."""
    T, H, W = in_thw
    alpha = slow_stride // fast_stride
    fw = width // beta_inv
    g = Graph(arch, (H, W), video=True)
    x = g.new_tensor(3, H, W, False, "input", T=T)
    g.input = x
    # fast pathway: conv 5x7x7 over every `fast_stride`-th frame
    f = g.conv3d(x, fw, (5, 7), (fast_stride, 2), (2 * fast_stride, 3), "fast_conv1.weight", bn="fast_bn1",
                 name="fast_stem", dil_t=fast_stride)
    f = g.maxpool3d(f, (1, 3), (1, 2), (0, 1), name="fast_maxpool")
    # slow pathway: conv 1x7x7 over every `slow_stride`-th frame, concatenated with the lateral connection
    ft = g.tensors[f]
    cat = g.new_buffer(width + fw * fusion_ratio)
    s = g.conv3d(x, width, (1, 7), (slow_stride, 2), (0, 3), "slow_conv1.weight", bn="slow_bn1", name="slow_stem")
    st = g.tensors[s]
    sp_t = g.new_tensor(width, ft.H, ft.W, False, "slow_maxpool", buf=cat, c_off=0, T=st.T)
    g.nodes.append(PoolNode(s, sp_t, 3, 2, 1, False, "maxpool", 1, 1, 0))
    lat_t = g.new_tensor(fw * fusion_ratio, ft.H, ft.W, True, "lateral_p1", buf=cat, c_off=width, T=st.T)
    g.nodes.append(ConvNode(f, lat_t, fw, fw * fusion_ratio, 1, 1, 1, 0, "lateral_p1.0.weight", None, "lateral_p1.1",
                            True, None, None, "conv", fusion_kernel, alpha, fusion_kernel // 2, 1))
    assert (ft.T + 2 * (fusion_kernel // 2) - fusion_kernel) // alpha + 1 == st.T, "pathway frame counts disagree"
    xs = g.new_tensor(width + fw * fusion_ratio, ft.H, ft.W, False, "slow_cat", buf=cat, c_off=0, T=st.T)

    def stage(x, inplanes, planes, prefix, head_t):
        for b in range(blocks):
            p = f"{prefix}.{b}"
            a = g.conv3d(x, planes, (head_t, 1), (1, 1), (head_t // 2, 0), f"{p}.conv1.weight", bn=f"{p}.bn1", name=f"{p}.conv1")
            a = g.conv3d(a, planes, (1, 3), (1, 1), (0, 1), f"{p}.conv2.weight", bn=f"{p}.bn2", name=f"{p}.conv2")
            if inplanes != planes * 4:
                idt = g.conv3d(x, planes * 4, (1, 1), (1, 1), (0, 0), f"{p}.downsample.0.weight",
                               bn=f"{p}.downsample.1", relu=False, name=f"{p}.downsample")
            else:
                idt = x
            x = g.conv3d(a, planes * 4, (1, 1), (1, 1), (0, 0), f"{p}.conv3.weight", bn=f"{p}.bn3", residual=idt,
                         name=f"{p}.out")
            inplanes = planes * 4
        return x
    fr = stage(f, fw, fw, "fast_res2", 3)
    sr = stage(xs, width + fw * fusion_ratio, width, "slow_res2", 1)
    g.hooks[1] = fr
    g.hooks[2] = sr
    return g


def slowfast_resnet(layers=(3, 4, 6, 3), width=64, in_thw=(32, 224, 224), arch="slowfast_resnet50_full", slow_stride=8,
                    fast_stride=1, beta_inv=8, fusion_ratio=2, fusion_kernel=5) -> Graph:
    """The WHOLE SlowFast backbone (Feichtenhofer et al.; gluoncv `slowfast_4x16_resnet50/101_kinetics400`): both pathways
    through res2..res5 with a time-strided lateral convolution (5x1x1 / (alpha,1,1), BN, ReLU) from the fast into the slow
    pathway in front of every slow stage.  Slow pathway: temporal kernels only in res4 / res5; fast pathway: 3x1x1 in every
    block.  hooks[1] / hooks[2] = fast_res2 / slow_res2 as in `slowfast_res2` (same weight keys, same order), and
    `classifier_feats` = [slow_res5, fast_res5]: the head pools each over (T, H, W), concatenates them in that order and
    applies `fc` (the classifier of the white-box sign-step attacks, `attack.py:63-96`).  Parity with gluoncv's module
    layout unpinned, like the other video graphs."""
    T, H, W = in_thw
    alpha = slow_stride // fast_stride
    fw = width // beta_inv
    g = Graph(arch, (H, W), video=True)
    x = g.new_tensor(3, H, W, False, "input", T=T)
    g.input = x
    f = g.conv3d(x, fw, (5, 7), (fast_stride, 2), (2 * fast_stride, 3), "fast_conv1.weight", bn="fast_bn1",
                 name="fast_stem", dil_t=fast_stride)
    f = g.maxpool3d(f, (1, 3), (1, 2), (0, 1), name="fast_maxpool")
    ft = g.tensors[f]
    s = g.conv3d(x, width, (1, 7), (slow_stride, 2), (0, 3), "slow_conv1.weight", bn="slow_bn1", name="slow_stem")
    st = g.tensors[s]

    def lateral(src, cat, c_off, key):
        """fast feature -> 5x1x1 / alpha convolution + BN + ReLU into channels [c_off, ...) of the slow pathway's next input"""
        sp = g.tensors[src]
        t = g.new_tensor(sp.C * fusion_ratio, sp.H, sp.W, True, key, buf=cat, c_off=c_off,
                         T=(sp.T + 2 * (fusion_kernel // 2) - fusion_kernel) // alpha + 1)
        g.nodes.append(ConvNode(src, t, sp.C, sp.C * fusion_ratio, 1, 1, 1, 0, f"{key}.0.weight", None, f"{key}.1",
                                True, None, None, "conv", fusion_kernel, alpha, fusion_kernel // 2, 1))
        return t

    cat = g.new_buffer(width + fw * fusion_ratio)
    sp_t = g.new_tensor(width, ft.H, ft.W, False, "slow_maxpool", buf=cat, c_off=0, T=st.T)
    g.nodes.append(PoolNode(s, sp_t, 3, 2, 1, False, "maxpool", 1, 1, 0))
    lat = lateral(f, cat, width, "lateral_p1")
    assert g.tensors[lat].T == st.T, "pathway frame counts disagree"
    xs = g.new_tensor(width + fw * fusion_ratio, ft.H, ft.W, False, "slow_cat1", buf=cat, c_off=0, T=st.T)

    def stage(x, planes, nblocks, stride, prefix, head_t, out_buf=None):
        inplanes = g.tensors[x].C
        for b in range(nblocks):
            sb = stride if b == 0 else 1
            p = f"{prefix}.{b}"
            a = g.conv3d(x, planes, (head_t, 1), (1, 1), (head_t // 2, 0), f"{p}.conv1.weight", bn=f"{p}.bn1", name=f"{p}.conv1")
            a = g.conv3d(a, planes, (1, 3), (1, sb), (0, 1), f"{p}.conv2.weight", bn=f"{p}.bn2", name=f"{p}.conv2")
            if sb != 1 or inplanes != planes * 4:
                idt = g.conv3d(x, planes * 4, (1, 1), (1, sb), (0, 0), f"{p}.downsample.0.weight",
                               bn=f"{p}.downsample.1", relu=False, name=f"{p}.downsample")
            else:
                idt = x
            last = b == nblocks - 1 and out_buf is not None
            x = g.conv3d(a, planes * 4, (1, 1), (1, 1), (0, 0), f"{p}.conv3.weight", bn=f"{p}.bn3", residual=idt,
                         name=f"{p}.out", dst_buf=out_buf if last else None, dst_c_off=0)
            inplanes = planes * 4
        return x

    slow_t = (1, 1, 3, 3)                       # temporal kernel of conv1 in the slow pathway's stages
    fr, sr = f, xs
    for li, nb in enumerate(layers):
        stride = 1 if li == 0 else 2
        fr = stage(fr, fw * 2 ** li, nb, stride, f"fast_res{li + 2}", 3)
        nxt = None
        if li < len(layers) - 1:                # the slow stage's output lands next to the lateral it is concatenated with
            nxt = g.new_buffer(width * 4 * 2 ** li + g.tensors[fr].C * fusion_ratio)
        sr = stage(sr, width * 2 ** li, nb, stride, f"slow_res{li + 2}", slow_t[li], nxt)
        if li == 0:
            g.hooks[1], g.hooks[2] = fr, sr
        if nxt is not None:
            so = g.tensors[sr]
            lt = lateral(fr, nxt, so.C, f"lateral_res{li + 2}")
            assert (g.tensors[lt].T, g.tensors[lt].H) == (so.T, so.H), "pathway shapes disagree"
            sr = g.new_tensor(so.C + g.tensors[lt].C, so.H, so.W, False, f"slow_cat{li + 2}", buf=nxt, c_off=0, T=so.T)
    g.hooks[3], g.hooks[4] = sr, fr
    g.classifier_feats = [sr, fr]
    return g


def tpn_resnet(layers=(3, 4, 6, 3), width=64, in_thw=(32, 224, 224), arch="tpn_resnet50") -> Graph:
    """Backbone of TPN (Yang et al., "Temporal Pyramid Network") up to `layer2`, the module the reference hooks for
    'tpn' models (`image_attacks.py:517-518`).  TPN's backbone is a SlowOnly-style 3-D ResNet: stem 1x7x7 / (1,2,2),
    max-pool 1x3x3 / (1,2,2), and NO temporal convolution in `layer1` / `layer2` (the 3x1x1 inflation starts at
    `layer3`) -- up to the hook it is a per-frame 2-D ResNet, i.e. frame-major image launches.  Keys: torchvision-style
    `conv1 / bn1 / layer{i}.{b}.conv{1,2,3} / bn{1,2,3} / downsample.{0,1}` (parity with gluoncv's module layout
    unpinned, like the other video graphs)."""
    T, H, W = in_thw
    g = Graph(arch, (H, W), video=True)
    x = g.new_tensor(3, H, W, False, "input", T=T)
    g.input = x
    x = g.conv3d(x, width, (1, 7), (1, 2), (0, 3), "conv1.weight", bn="bn1", name="stem")
    x = g.maxpool3d(x, (1, 3), (1, 2), (0, 1), name="maxpool")
    inplanes = width
    for li, nblocks in enumerate(layers[:2]):
        planes = width * (2 ** li)
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 0) else 1
            p = f"layer{li + 1}.{b}"
            a = g.conv3d(x, planes, (1, 1), (1, 1), (0, 0), f"{p}.conv1.weight", bn=f"{p}.bn1", name=f"{p}.conv1")
            a = g.conv3d(a, planes, (1, 3), (1, stride), (0, 1), f"{p}.conv2.weight", bn=f"{p}.bn2", name=f"{p}.conv2")
            if stride != 1 or inplanes != planes * 4:
                idt = g.conv3d(x, planes * 4, (1, 1), (1, stride), (0, 0), f"{p}.downsample.0.weight",
                               bn=f"{p}.downsample.1", relu=False, name=f"{p}.downsample")
            else:
                idt = x
            x = g.conv3d(a, planes * 4, (1, 1), (1, 1), (0, 0), f"{p}.conv3.weight", bn=f"{p}.bn3", residual=idt,
                         name=f"{p}.out")
            inplanes = planes * 4
        g.hooks[li + 1] = x
    return g


# The reference's configs name `slowfast_8x8_resnet50/101_kinetics400` (`utils.py:11-12`): gluoncv's 8x8 variant samples the slow
# pathway every 8th and the fast pathway every 2nd frame (alpha = 4) and fuses with a 7x1x1 lateral kernel (the 4x16 variant: 16 / 2,
# alpha = 8, kernel 5).  On the 32-frame clips of this path that is 4 slow and 16 fast frames.
SLOWFAST_8X8 = dict(slow_stride=8, fast_stride=2, fusion_kernel=7)


def build_video(model_type: str, in_thw=(32, 224, 224), full: bool = False) -> Graph:
    """`model_type` follows `image_fine_tune_attack.py:53` / `utils.py:9-14`.  `full`: the graph to its last stage (what a
    classifier head needs); the I3D graphs always are, SlowFast then gets res3..res5 and its lateral connections."""
    if full and model_type in ("slowfast_resnet50", "slowfast_resnet101"):
        return slowfast_resnet((3, 4, 23, 3) if "101" in model_type else (3, 4, 6, 3), 64, in_thw, model_type + "_full", **SLOWFAST_8X8)
    # the reference's 'i3d_resnet50' / 'i3d_resnet101' ARE gluoncv's i3d_nl5_resnet50/101_v1 (utils.py:9-10): five non-local blocks
    if model_type in ("i3d_resnet50", "i3d_nl5_resnet50"):
        return i3d_resnet((3, 4, 6, 3), 64, in_thw, "i3d_nl5_resnet50", nonlocal_freq=NL5_FREQ)
    if model_type in ("i3d_resnet101", "i3d_nl5_resnet101"):
        return i3d_resnet((3, 4, 23, 3), 64, in_thw, "i3d_nl5_resnet101", nonlocal_freq=NL5_FREQ)
    if model_type == "i3d_plain_resnet50":          # the inflated ResNet without non-local blocks (not a reference configuration)
        return i3d_resnet((3, 4, 6, 3), 64, in_thw, "i3d_plain_resnet50")
    if model_type == "i3d_plain_resnet101":
        return i3d_resnet((3, 4, 23, 3), 64, in_thw, "i3d_plain_resnet101")
    if model_type == "slowfast_resnet50":
        return slowfast_res2(64, in_thw, "slowfast_resnet50", **SLOWFAST_8X8)
    if model_type == "slowfast_resnet101":          # res2 is identical for the 50- and 101-layer variants
        return slowfast_res2(64, in_thw, "slowfast_resnet101", **SLOWFAST_8X8)
    if model_type == "tpn_resnet50":
        return tpn_resnet((3, 4, 6, 3), 64, in_thw, "tpn_resnet50")
    if model_type == "tpn_resnet101":          # layer1 / layer2 are the same for the 50- and 101-layer backbones
        return tpn_resnet((3, 4, 23, 3), 64, in_thw, "tpn_resnet101")
    raise KeyError(f"video backbone {model_type!r} is not built")


TINY_NL_FREQ = ((0, 0), (1, 1), (1,), (0,))      # tiny I3D: a non-local block behind both blocks of the hooked stage and one in the next


def build_video_tiny(model_type: str, in_thw=(8, 32, 32), full: bool = False) -> Graph:
    if full and "slowfast" in model_type:
        return slowfast_resnet((2, 2, 1, 1), 16, in_thw, "slowfast_tiny_full", slow_stride=4, fast_stride=1, beta_inv=4)
    if "tpn" in model_type:
        return tpn_resnet((2, 2, 1, 1), 8, in_thw, "tpn_tiny")
    if "i3d" in model_type:
        plain = "plain" in model_type
        return i3d_resnet((2, 2, 1, 1), 8, in_thw, "i3d_plain_tiny" if plain else "i3d_tiny", inflate=((1, 1), (1, 0), (1,), (0,)),
                          nonlocal_freq=None if plain else TINY_NL_FREQ)
    return slowfast_res2(16, in_thw, "slowfast_tiny", slow_stride=4, fast_stride=1, beta_inv=4, blocks=2)


def relu_module_names(g: Graph) -> dict:
    """{output tensor of a ReLU convolution: qualified name of the gluoncv `nn.ReLU` module that applies it} for the video
    backbones -- what `base_attacks.SGM` (:511-513) selects its hooks by.  A residual block owns ONE ReLU module, `<block>.relu`,
    applied after each of its three convolutions; the stems' are top-level modules named `*relu` (`relu`; SlowFast: `fast_relu` /
    `slow_relu`); the ReLUs inside SlowFast's lateral `nn.Sequential`s are numbered children (`lateral_p1.2`), no `relu` in their
    names, and are left out."""
    out = {}
    for nd in g.nodes:
        if getattr(nd, "op", "") != "conv" or not nd.relu:
            continue
        parts = nd.weight.split(".")
        if len(parts) == 2 and parts[0].endswith("conv1"):                       # conv1 / fast_conv1 / slow_conv1
            out[nd.dst] = parts[0][:-len("conv1")] + "relu"
        elif len(parts) >= 3 and parts[-1] == "weight" and parts[-2] in ("conv1", "conv2", "conv3"):
            out[nd.dst] = ".".join(parts[:-2] + ["relu"])
    return out


def video_hooks(g: Graph, model_type: str) -> List[int]:
    """Hooked tensors (`image_attacks.py:513-519`).  ILAF's loss is a plain sum over the hooked layers, each paired
    with its own clean feature, so the order is immaterial."""
    if "i3d" in model_type or "tpn" in model_type:          # res_layers['1'] / layer2: the second stage
        return [g.hooks[2]]
    return [g.hooks[1], g.hooks[2]]


# ---------------------------------------------------------------------------
# name -> graph, following the reference's `get_model` vocabulary
# ---------------------------------------------------------------------------
def build(model_name: str, in_hw=(224, 224)) -> Graph:
    """`model_name` uses the reference's names (`image_attacks.py:85-87`):
    'resnet' is ResNet-101 there (`:94-95`); 'resnet50' is added because
    BASELINE.json quotes its metric on ResNet-50."""
    if model_name == "resnet":
        return resnet((3, 4, 23, 3), 64, in_hw, "resnet101")
    if model_name == "resnet50":
        return resnet((3, 4, 6, 3), 64, in_hw, "resnet50")
    if model_name == "vgg":
        return vgg(VGG16_CFG, in_hw, "vgg16")
    if model_name == "alexnet":
        return alexnet(1, in_hw)
    if model_name == "squeezenet":
        return squeezenet(1, in_hw)
    if model_name == "densenet121":     # extension (BASELINE.json configs[2] names it); see `densenet`
        return densenet(32, (6, 12, 24, 16), 64, 4, in_hw, "densenet121")
    if model_name == "densenet161":
        return densenet(48, (6, 12, 36, 24), 96, 4, in_hw, "densenet161")
    if model_name == "densenet":
        # The reference constructs densenet161 (`image_attacks.py:96-97`) but no attack class
        # has a densenet branch in `_find_target_layer` (`:260-271`): the hook lookup returns
        # None and `None.register_forward_hook` raises AttributeError.  Mirror that error.
        raise AttributeError("'NoneType' object has no attribute 'register_forward_hook'")
    # the reference leaves `model` unbound for unknown names (`image_attacks.py:103`)
    raise UnboundLocalError("local variable 'model' referenced before assignment")


#: tiny variants used by parity tests / golden fixtures (same topology, fewer channels/blocks)
def build_tiny(model_name: str, in_hw=(64, 64)) -> Graph:
    if model_name in ("resnet", "resnet50"):
        return resnet((2, 1, 2, 1), 8, in_hw, "resnet_tiny")
    if model_name == "vgg":
        cfg = (8, 8, "M", 16, 16, "M", 16, 16, 16, "M", 32, 32, 32, "M", 32, 32, 32, "M")
        return vgg(cfg, in_hw, "vgg_tiny")
    if model_name == "alexnet":
        return alexnet(8, in_hw, "alexnet_tiny")
    if model_name == "squeezenet":
        return squeezenet(4, in_hw, "squeezenet_tiny")
    if model_name in ("densenet121", "densenet161"):
        return densenet(8, (2, 3, 2, 2), 16, 2, in_hw, "densenet_tiny")
    return build(model_name, in_hw)
