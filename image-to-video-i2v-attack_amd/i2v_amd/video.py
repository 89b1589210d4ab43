"""White-box VIDEO backbones for ILAF (`/root/reference/image_attacks.py:498-629`).

The reference builds them with `gluoncv.torch.model_zoo.get_model(cfg)` (`image_fine_tune_attack.py:58-66`) and
hands the torch module to `ILAF(model, model_type)`.  gluoncv is neither vendored nor installed, so the native path
takes a `VideoModel` instead: the graph IR of the backbone up to the hooked stage (`graphs.build_video`) plus its
weights.  `ILAF` still accepts an arbitrary torch module (then the model runs in PyTorch and only the update rule is
native); handing it a `VideoModel` makes the whole loop -- compose, 3-D forward to the hooks, ILAF loss, input
gradient, sign step -- run in `libi2v_hip.so`.
"""
from typing import Optional

from . import graphs as _graphs
from . import weights as _weights


class VideoModel(object):
    """`model_type` as in `image_fine_tune_attack.py:53` ('i3d_resnet50', 'i3d_resnet101', 'slowfast_resnet50',
    'slowfast_resnet101', 'tpn_resnet50', 'tpn_resnet101').  Weights: `state_dict` (graph key layout; `weights.convert_gluoncv_state_dict` maps a gluoncv
    checkpoint onto it), else `$I2V_WEIGHTS_DIR/<arch>.pth`, else -- only with an explicit `weight_seed` or
    `I2V_SYNTHETIC_WEIGHTS=1` -- the seeded synthetic initialiser (`weights.load_state_dict`)."""

    def __init__(self, model_type: str, in_thw=(32, 224, 224), weight_seed: Optional[int] = None, tiny: bool = False,
                 state_dict: Optional[dict] = None, num_classes: Optional[int] = None):
        """`num_classes`: also carry the classifier head (global average pool -> `fc`), which makes the model usable as the
        white-box CLASSIFIER of the BIM family (`attack.py:63-96`, `base_attacks.py:261-340`) with the whole cross-entropy
        gradient computed natively.  Only graphs that reach their last stage have one (the I3D ResNets; SlowFast, whose graph
        is then built through res5 with all lateral connections)."""
        self.num_classes = num_classes
        if num_classes is not None and not ("i3d" in model_type or "slowfast" in model_type):
            raise KeyError(f"{model_type!r}: only the I3D and SlowFast graphs are built to their last stage; no native classifier head")
        self.model_type = model_type
        self.in_thw = tuple(in_thw)
        self.tiny = tiny
        self.weight_seed = weight_seed
        self._sd = state_dict
        self._loaded = None                  # (arch, state_dict) read from $I2V_WEIGHTS_DIR or synthesised: backbone AND head
        self.training = False
        self.graph_for(self.in_thw)          # unknown names fail at construction, like get_model(cfg)

    def graph_for(self, thw):
        build = _graphs.build_video_tiny if self.tiny else _graphs.build_video
        return build(self.model_type, tuple(thw), full=self.num_classes is not None)

    def state_dict_for(self, graph):
        if self._sd is not None:
            return self._sd
        if self._loaded is None or self._loaded[0] != graph.arch:
            self._loaded = (graph.arch, _weights.load_state_dict(graph, self.weight_seed, keep_head=self.num_classes is not None))
        return self._loaded[1]

    def hook_tensors(self, graph):
        return _graphs.video_hooks(graph, self.model_type)

    # ---- classifier head (num_classes given) ----
    def classifier_hook(self, graph):
        """The tensors the head pools and concatenates: the output of the last stage (I3D), or of both pathways' last
        stages, slow first (SlowFast)."""
        return list(graph.classifier_feats) or [graph.hooks[max(graph.hooks)]]

    def head_weights(self, graph):
        """(fc.weight (K, C), fc.bias (K,)) -- from the state_dict (gluoncv names its head `fc` as torchvision does) or,
        under the same opt-in rules as the backbone, seeded synthetic values."""
        import torch
        C_ = sum(graph.tensors[t].C for t in self.classifier_hook(graph))
        sd = self.state_dict_for(graph)      # the SAME source as the backbone: a checkpoint's `fc.*` travels with it
        if "fc.weight" in sd:
            w, b = sd["fc.weight"].float(), sd.get("fc.bias")
            if tuple(w.shape) != (self.num_classes, C_):
                raise ValueError(f"fc.weight has shape {tuple(w.shape)}, expected {(self.num_classes, C_)}")
            return w.contiguous(), (b.float().contiguous() if b is not None else None)
        source = "state_dict" if self._sd is not None else _weights.SOURCES.get(graph.arch, "")
        if not source.startswith("synthetic"):
            # a real backbone with a random classifier would attack a meaningless model and still write valid-looking files
            raise KeyError(f"{source or 'the state_dict'} has no fc.weight for the classifier head of {graph.arch!r}; "
                           "a pretrained backbone is never paired with a synthetic head")
        gen = torch.Generator().manual_seed(7919 * (self.weight_seed or 0) + 13)
        return (torch.randn(self.num_classes, C_, generator=gen) * (1.0 / C_) ** 0.5).contiguous(), \
            (torch.randn(self.num_classes, generator=gen) * 0.01).contiguous()

    # the reference calls these on the torch module (`image_fine_tune_attack.py:67`, base_attacks.py:227-229)
    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self


class NativeClassifier(object):
    """Forward-only classifier for the evaluator (`/root/reference/reference.py:108-129`: `model(clips)` -> logits, top-1): a
    `VideoModel` with its head, every launch behind the C ABI (3-D backbone to the last stage, global average pool, `fc`).
    Quacks like the torch module the evaluator expects (`.to()`, `.eval()`, call)."""

    def __init__(self, model: VideoModel, engine=None):
        if model.num_classes is None:
            raise ValueError("a VideoModel used as a classifier needs num_classes (its head)")
        self.model, self._engine = model, engine
        self._net = self._key = self._head = None

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def _forward(self, clips):
        """Backbone forward of a (b,3,f,h,w) batch to the classifier's features; returns (engine, net, frames, clips)."""
        import torch
        from .attacks import get_engine
        eng = self._engine or get_engine()
        kw = dict(dtype=torch.float32, device=eng.device)
        clips = clips.detach().to(**kw).contiguous()
        b, c, f, h, w = clips.shape
        N, key = b * f, (f, h, w)
        if self._net is None or self._key != key or self._net.max_frames < N:
            if self._net is not None:
                self._net.close()
            g = self.model.graph_for((f, h, w))
            self._net = eng.build_net(g, self.model.state_dict_for(g), self.model.classifier_hook(g), N)
            self._head = tuple(t.to(eng.device) if t is not None else None for t in self.model.head_weights(g))
            self._key = key
        net = self._net
        x, u = torch.empty(N, 3, h, w, **kw), torch.empty(N, 3, h, w, **kw)
        eng.frames_from_video(clips, x, u)
        net.forward(x)
        return eng, net, N, b

    def _head_apply(self, eng, net, N, b, W, bias):
        import torch
        kw = dict(dtype=torch.float32, device=eng.device)
        logits = torch.empty(b, W.shape[0], **kw)
        scratch = torch.empty(eng.capi.i2v_head_scratch_bytes(W.shape[1], b), dtype=torch.uint8, device=eng.device)
        net.head_logits(list(range(len(net.hooks))), W, bias, N, logits, scratch)
        return logits

    def __call__(self, clips):
        eng, net, N, b = self._forward(clips)
        W, bias = self._head
        return self._head_apply(eng, net, N, b, W, bias)

    def pooled_features(self, clips):
        """The pooled pre-`fc` features (clips, C) the head multiplies (`i2v_head_pool_f32`; SlowFast: slow then fast): the same
        launches as a call, with the identity in place of `fc` (x + 0 sums are exact).  What an evaluator head is fitted on
        (`tools/fooling_parity.py`: a least-squares head on the clean clips of the sample list)."""
        import torch
        eng, net, N, b = self._forward(clips)
        C_ = self._head[0].shape[1]
        if getattr(self, "_eye", None) is None or self._eye.shape[0] != C_:
            self._eye = torch.eye(C_, dtype=torch.float32, device=eng.device)
        return self._head_apply(eng, net, N, b, self._eye, None)
