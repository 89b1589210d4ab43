"""TEST INFRASTRUCTURE -- import the reference's attack classes on CPU (this container only).

`/root/reference` never travels to the GPU box; everything that uses this module skips when
the directory is absent.  The reference modules are imported UNMODIFIED from where they lie;
only their missing third-party imports are stubbed:

  cv2                                   (image_cam_utils.py:1)        -> empty module
  timm.models.create_model              (TPAMI_attack.py:13)          -> stub
  gluoncv.torch.engine.config           (utils.py:2, via base_attacks)-> stub
  torchvision.models.{resnet101,...}    (image_attacks.py:88-101)     -> oracle/tv_models.py nets
  Tensor.cuda / Module.cuda             (image_attacks.py:45,103,...) -> identity
  numpy.math                            (video_attacks.py:69; removed in NumPy 2) -> the math module

Nothing here is reference code.
"""
import contextlib
import io
import os
import sys
import types

import torch

REFERENCE_DIR = os.environ.get("I2V_REFERENCE_DIR", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "image_attacks.py"))


class ModelFactory:
    """What `torchvision.models.<arch>(pretrained=True)` returns under the shim."""

    def __init__(self):
        self.tiny = True
        self.seed = 0
        self.in_hw = (64, 64)
        self.dtype = torch.float32

    def make(self, ref_name):
        from i2v_amd import graphs, weights
        from . import tv_models
        g = graphs.build_tiny(ref_name, self.in_hw) if self.tiny else graphs.build(ref_name, self.in_hw)
        sd = weights.synthetic_state_dict(g, self.seed)
        m = tv_models.make(ref_name, self.tiny)
        tv_models.load_backbone_weights(m, sd)
        return m.to(self.dtype)


FACTORY = ModelFactory()
_installed = False


def install():
    """Idempotently install the stubs and put the reference directory first on sys.path."""
    global _installed
    if _installed:
        return
    if not available():
        raise FileNotFoundError(REFERENCE_DIR)
    sys.dont_write_bytecode = True          # the reference directory is read-only
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_models.create_model = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("timm stub"))
    timm.models = timm_models
    sys.modules.setdefault("timm", timm)
    sys.modules.setdefault("timm.models", timm_models)
    names = ["gluoncv", "gluoncv.torch", "gluoncv.torch.engine", "gluoncv.torch.engine.config"]
    mods = [types.ModuleType(n) for n in names]
    mods[3].get_cfg_defaults = lambda: None
    for n, m in zip(names, mods):
        sys.modules.setdefault(n, m)
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    # reference vocabulary: image_attacks.py:88-101
    tvm.resnet101 = lambda pretrained=False, **k: FACTORY.make("resnet")
    tvm.vgg16 = lambda pretrained=False, **k: FACTORY.make("vgg")
    tvm.alexnet = lambda pretrained=False, **k: FACTORY.make("alexnet")
    tvm.squeezenet1_1 = lambda pretrained=False, **k: FACTORY.make("squeezenet")
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    import math
    import numpy
    if not hasattr(numpy, "math"):          # `np.math.exp` (video_attacks.py:69): the alias NumPy 2 removed
        numpy.math = math
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _installed = True


@contextlib.contextmanager
def quiet():
    """The reference prints the hooked module and the cost every step
    (`image_attacks.py:285,349`)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with contextlib.redirect_stdout(io.StringIO()):
            yield


_REF_MODULES = {}
_COLLIDING = ("image_attacks", "TPAMI_attack", "base_attacks", "utils", "image_main", "run_image_guided", "video_attacks")


def import_reference(module: str):
    """Load `/root/reference/<module>.py` under the private name `_ref_<module>`.  The product ships
    drop-in modules with the SAME names (`image_attacks`, `TPAMI_attack`, `base_attacks`); whatever
    of those is already imported is put back afterwards, so both can live in one test process."""
    if module in _REF_MODULES:
        return _REF_MODULES[module]
    install()
    import importlib.util
    saved = {k: sys.modules.pop(k) for k in _COLLIDING if k in sys.modules}
    path0 = list(sys.path)
    sys.path.insert(0, REFERENCE_DIR)
    try:
        with quiet():
            spec = importlib.util.spec_from_file_location("_ref_" + module, os.path.join(REFERENCE_DIR, module + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules["_ref_" + module] = mod
            spec.loader.exec_module(mod)
    finally:
        sys.path[:] = path0
        for k in _COLLIDING:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    _REF_MODULES[module] = mod
    return mod


class AdamTap:
    """Records, around every `torch.optim.Adam.step`, the gradient the reference handed to it and delta / exp_avg /
    exp_avg_sq after it -- the per-step checkpoints teacher-forcing tests start from (the reference builds its
    optimiser inside `forward`, `image_attacks.py:306`)."""

    def __init__(self):
        self.deltas, self.grads, self.ms, self.vs, self.grad0 = [], [], [], [], None

    def __enter__(self):
        self._orig = torch.optim.Adam.step
        tap = self

        def step(opt, *a, **k):
            p = opt.param_groups[0]["params"][0]
            if tap.grad0 is None:
                tap.grad0 = p.grad.detach().clone()
            tap.grads.append(p.grad.detach().clone())
            r = tap._orig(opt, *a, **k)
            tap.deltas.append(p.detach().clone())
            st = opt.state[p]
            tap.ms.append(st["exp_avg"].detach().clone())
            tap.vs.append(st["exp_avg_sq"].detach().clone())
            return r
        torch.optim.Adam.step = step
        return self

    def __exit__(self, *exc):
        torch.optim.Adam.step = self._orig
