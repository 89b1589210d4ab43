"""Host-side owner of the native handle: turns a graph IR + state_dict into a planned backbone
inside `libi2v_hip.so` and exposes forward-to-hooks / backward-to-input on torch tensors.

PyTorch is used for device memory and streams only (`tensor.data_ptr()`,
`torch.cuda.current_stream().cuda_stream`); all arithmetic happens behind the C ABI.
"""
import ctypes as C
import sys
import threading
import time
from typing import List, Sequence

import torch

from . import lib as _lib
from .graphs import Graph
from .weights import fold_affine, fold_pre_affine


_DEVICE_TYPES = {}          # device type -> number of engines ALIVE on it in this process ('cuda' for libi2v_hip.so)


def _ptr(t: torch.Tensor, eng: "Engine" = None):
    """Device pointer of a tensor handed to the library.  With `eng` the tensor must live on that engine's device type (every
    call that knows its engine passes it); without, on the device type of some engine that is still alive (an entry goes
    when its last engine closes -- a host-simulation engine that once existed does not let CPU tensors through for good)."""
    assert t.is_contiguous() and t.dtype == torch.float32, (t.dtype, t.is_contiguous())
    ok = t.device.type == eng.device.type if eng is not None else _DEVICE_TYPES.get(t.device.type, 0) > 0
    if not ok:      # a host pointer handed to a HIP kernel reads garbage or faults the GPU
        where = eng.device if eng is not None else sorted(k for k, v in _DEVICE_TYPES.items() if v > 0)
        raise _lib.I2VError(f"tensor on {t.device} passed to an engine on {where}: move it to the engine's device")
    return C.c_void_p(t.data_ptr())


def _hptr(t: torch.Tensor):
    """HOST array handed to the description calls (`i2v_net_add_conv*` take host weights / scale / shift)."""
    assert t.is_contiguous() and t.dtype == torch.float32 and t.device.type == "cpu", (t.dtype, t.device)
    return C.c_void_p(t.data_ptr())


class Engine:
    """One per (process, device) -- mirrors the reference's single `.cuda()` device
    (`image_attacks.py:103`).  `capi` is injected only by the planner unit tests."""

    def __init__(self, device="cuda:0", capi=None):
        self.device = torch.device(device)
        self.capi = capi if capi is not None else _lib.load()
        if capi is None and self.device.type != "cuda":
            raise _lib.I2VError("the I2V engine needs a ROCm device; there is no CPU path")
        self.h = C.c_void_p()
        # net creation / planning / destruction touch the handle's net table.  Re-entrant: Net.__del__ -> close() takes it, and the
        # cyclic GC may finalise an unreachable Net on the very thread that is inside build_net
        self.plan_lock = threading.RLock()
        self.plan_ms, self.plans = 0.0, 0      # host wall time spent building nets (never inside a timed region of bench.py)
        idx = self.device.index or 0
        _lib.check(self.capi, self.capi.i2v_create(idx, C.byref(self.h)))
        _DEVICE_TYPES[self.device.type] = _DEVICE_TYPES.get(self.device.type, 0) + 1

    def stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def close(self):
        if self.h:
            self.capi.i2v_destroy(self.h)
            self.h = C.c_void_p()
            _DEVICE_TYPES[self.device.type] = max(0, _DEVICE_TYPES.get(self.device.type, 0) - 1)

    def __del__(self):
        try:
            if not sys.is_finalizing():        # at interpreter exit the HIP runtime may already be tearing down: let the OS reclaim
                self.close()
        except Exception:
            pass

    def build_net(self, graph: Graph, state_dict, hook_tensors: Sequence[int], max_frames: int, relu_gain=None) -> "Net":
        """Thread-safe (concurrent clip streams plan their own nets); execution of DIFFERENT nets on different
        streams needs no lock -- a planned net owns its arena, the library keeps no other mutable state."""
        with self.plan_lock:
            t0 = time.perf_counter()
            net = Net(self, graph, state_dict, list(hook_tensors), max_frames, relu_gain)
            self.plan_ms += 1e3 * (time.perf_counter() - t0)      # pack + upload + plan + autotune (i2v_net_plan syncs)
            self.plans += 1
            return net

    # ---- measurement ----
    KINDS = ("conv_igemm_fwd", "conv_igemm_imggrad", "pool_fwd", "pool_bwd", "addmask", "conv_igemm_dgrad")

    def timing_enable(self, on=True):
        """True / 1: one HIP event pair per backbone launch (per-launch times, the low-intensity split, dump lines);
        'segments' / 2: one pair per run of consecutive launches of one kind (a forward list is 3-4 segments: ~2 % less
        overhead on the headline attack, same per-kind totals); False: off."""
        mode = 2 if on in (2, "segments") else (1 if on else 0)
        _lib.check(self.capi, self.capi.i2v_timing_enable(self.h, mode))

    def timing_collect(self):
        """Per kernel kind: device ms, algorithmic flops, launches, algorithmic bytes, and the same over the launches
        whose flops/byte is below the machine balance (`lowi_*`: the HBM-bound ones)."""
        n, f = len(self.KINDS), 8
        out = (C.c_double * (n * f))()
        _lib.check(self.capi, self.capi.i2v_timing_collect_ex(self.h, out, n, f))
        keys = ("ms", "flops", "launches", "bytes", "lowi_ms", "lowi_bytes", "lowi_launches", "lowi_flops")
        return {k: {key: out[i * f + j] for j, key in enumerate(keys)} for i, k in enumerate(self.KINDS)}

    # ---- loop kernels (thin, typed wrappers) ----
    def clip_from_u8(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """(b,t,h,w,3) uint8 decoded frames on the device -> normalised (b,3,t,h,w) float32 clip."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.shape[-1] == 3
        b, t, h, w, _ = frames_u8.shape
        out = torch.empty(b, 3, t, h, w, dtype=torch.float32, device=frames_u8.device)
        _lib.check(self.capi, self.capi.i2v_clip_from_u8_f32(C.c_void_p(frames_u8.data_ptr()), _ptr(out, self), b, t, h, w, self.stream()))
        return out

    def clip_resize_crop(self, frames_u8: torch.Tensor, short_side=256, crop=224) -> torch.Tensor:
        """(b,t,H,W,3) uint8 decoded frames on the device -> the reference's validation transform (datasets.py:86-93:
        resize the short side to `short_side` with cv2-style 8-bit bilinear, centre-crop `crop`, /255, normalise) ->
        (b,3,t,crop,crop) float32, in one kernel."""
        from . import clips as _clips
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.shape[-1] == 3
        b, t, H, W, _ = frames_u8.shape
        rh, rw = _clips.resize_sizes(H, W, short_side)
        cy, cx = _clips.center_crop_origin(rh, rw, crop, crop)
        key = (H, W, rh, rw)
        tabs = self.__dict__.setdefault("_resize_tabs", {})
        if key not in tabs:
            tabs[key] = tuple(torch.from_numpy(_clips.resize_table(n_out, n_in)).to(frames_u8.device) for n_out, n_in in ((rw, W), (rh, H)))
        xt, yt = tabs[key]
        out = torch.empty(b, 3, t, crop, crop, dtype=torch.float32, device=frames_u8.device)
        _lib.check(self.capi, self.capi.i2v_clip_resize_crop_u8_f32(
            C.c_void_p(frames_u8.data_ptr()), _ptr(out, self), C.c_void_p(xt.data_ptr()), C.c_void_p(yt.data_ptr()), b, t, H, W, rh, rw,
            cy, cx, crop, crop, self.stream()))
        return out

    def clip_resample_crop(self, frames_u8: torch.Tensor, size=224, crop=224) -> torch.Tensor:
        """(b,t,H,W,3) uint8 decoded frames on the device -> the UCF-101 loader's validation transform (dataset_ucf101.py:113-126:
        PIL BILINEAR `Scale(size)`, `CornerCrop(crop, 'c')`, ToTensor, Normalize) -> (b,3,t,crop,crop) float32, in one kernel."""
        from . import clips as _clips
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.shape[-1] == 3
        b, t, H, W, _ = frames_u8.shape
        rh, rw = _clips.scale_sizes(H, W, size)
        cy, cx = _clips.corner_crop_center_origin(rh, rw, crop)
        key = ("pil", H, W, rh, rw)
        tabs = self.__dict__.setdefault("_resize_tabs", {})
        if key not in tabs:
            (xb, xk, kx), (yb, yk, ky) = _clips.pil_resample_table(W, rw), _clips.pil_resample_table(H, rh)
            tabs[key] = tuple(torch.from_numpy(a).to(frames_u8.device) for a in (xb, xk, yb, yk)) + (kx, ky)
        xb, xk, yb, yk, kx, ky = tabs[key]
        out = torch.empty(b, 3, t, crop, crop, dtype=torch.float32, device=frames_u8.device)
        _lib.check(self.capi, self.capi.i2v_clip_resample_crop_u8_f32(
            C.c_void_p(frames_u8.data_ptr()), _ptr(out, self), C.c_void_p(xb.data_ptr()), C.c_void_p(xk.data_ptr()), kx, C.c_void_p(yb.data_ptr()),
            C.c_void_p(yk.data_ptr()), ky, b, t, H, W, rh, rw, cy, cx, crop, crop, self.stream()))
        return out

    def frames_from_video(self, video, x, u):
        b, c, f, h, w = video.shape
        _lib.check(self.capi, self.capi.i2v_frames_from_video_f32(_ptr(video, self), _ptr(x, self), _ptr(u, self), b, f, h, w, self.stream()))

    def compose(self, u, delta, out, b, f, eps, video_layout=False):
        h, w = u.shape[-2:]
        _lib.check(self.capi, self.capi.i2v_compose_f32(_ptr(u, self), _ptr(delta, self), _ptr(out, self), b, f, h, w, eps,
                                                        1 if video_layout else 0, self.stream()))

    def adam_step(self, delta, m, v, gx, u, eps, lr, step_t, beta1=0.9, beta2=0.999, adam_eps=1e-8):
        n, _, h, w = delta.shape
        _lib.check(self.capi, self.capi.i2v_adam_step_f32(_ptr(delta, self), _ptr(m, self), _ptr(v, self), _ptr(gx, self), _ptr(u, self), n, h * w,
                                                          eps, lr, beta1, beta2, adam_eps, step_t, self.stream()))

    def sign_step(self, adv, u, grad, chan_stride, step, eps):
        _lib.check(self.capi, self.capi.i2v_sign_step_f32(_ptr(adv, self), _ptr(u, self), _ptr(grad, self), adv.numel(), chan_stride,
                                                          step, eps, self.stream()))

    def sign_step_delta_gx(self, delta, gx, u, eps, step):
        _lib.check(self.capi, self.capi.i2v_sign_step_delta_gx_f32(_ptr(delta, self), _ptr(gx, self), _ptr(u, self), delta.numel(), eps, step,
                                                                   self.stream()))

    def sign_step_delta(self, delta, grad, step):
        _lib.check(self.capi, self.capi.i2v_sign_step_delta_f32(_ptr(delta, self), _ptr(grad, self), delta.numel(), step, self.stream()))

    def tt_grad_mix(self, grads, kernel, moves, weight):
        """`TemporalTranslation._grad_augmentation` (video_attacks.py:160-175): grads (D, N, C, T, H, W) on the device, kernel /
        moves host sequences of length D -> (N, C, T, H, W)."""
        import numpy as np
        D, N, C_, T, H, W = grads.shape
        k = np.ascontiguousarray(np.asarray(kernel, dtype=np.float32).reshape(D))
        mv = np.ascontiguousarray(np.asarray(moves, dtype=np.int32).reshape(D))
        out = torch.empty(N, C_, T, H, W, dtype=torch.float32, device=grads.device)
        _lib.check(self.capi, self.capi.i2v_tt_grad_mix_f32(_ptr(grads.contiguous(), self), _ptr(out, self), C.c_void_p(k.ctypes.data), C.c_void_p(mv.ctypes.data),
                                                            D, N * C_, T, H * W, float(weight), self.stream()))
        return out

    # ---- base_attacks.py transforms (DI / TI) ----
    def resample_nearest(self, src: torch.Tensor, map_y: torch.Tensor, map_x: torch.Tensor) -> torch.Tensor:
        """dst[..., y, x] = src[..., map_y[y], map_x[x]] (0 where a map is negative); maps: int32 tensors on the device."""
        Hs, Ws = src.shape[-2:]
        out = torch.empty(*src.shape[:-2], map_y.numel(), map_x.numel(), dtype=torch.float32, device=src.device)
        planes = src.numel() // (Hs * Ws)
        _lib.check(self.capi, self.capi.i2v_resample_nearest_f32(_ptr(src, self), _ptr(out, self), planes, Hs, Ws, map_y.numel(), map_x.numel(),
                                                                 C.c_void_p(map_y.data_ptr()), C.c_void_p(map_x.data_ptr()), self.stream()))
        return out

    def resample_nearest_bwd(self, g: torch.Tensor, src_hw, ranges) -> torch.Tensor:
        """Transpose of `resample_nearest`: `ranges` = (ylo, yhi, xlo, xhi) int32 device tensors over the SOURCE rows / columns."""
        Hd, Wd = g.shape[-2:]
        Hs, Ws = src_hw
        out = torch.empty(*g.shape[:-2], Hs, Ws, dtype=torch.float32, device=g.device)
        planes = g.numel() // (Hd * Wd)
        _lib.check(self.capi, self.capi.i2v_resample_nearest_bwd_f32(_ptr(g, self), _ptr(out, self), planes, Hd, Wd, Hs, Ws,
                                                                     *(C.c_void_p(r.data_ptr()) for r in ranges), self.stream()))
        return out

    def dwconv1d(self, src: torch.Tensor, taps, axis: int) -> torch.Tensor:
        """Depthwise 1-D convolution (zero padding, odd len(taps) <= 64) along `axis` of a dense tensor, out of place."""
        import numpy as np
        axis = axis % src.dim()
        k = np.ascontiguousarray(np.asarray(taps, dtype=np.float32))
        outer = int(np.prod(src.shape[:axis], dtype=np.int64)) if axis > 0 else 1
        inner = int(np.prod(src.shape[axis + 1:], dtype=np.int64)) if axis + 1 < src.dim() else 1
        out = torch.empty_like(src)
        _lib.check(self.capi, self.capi.i2v_dwconv1d_f32(_ptr(src, self), _ptr(out, self), outer, src.shape[axis], inner,
                                                         C.c_void_p(k.ctypes.data), len(k), self.stream()))
        return out

    GRAD_POST_MODES = {None: 0, "frame": 1, "clip": 2, "column": 3, "l1": 4}

    def grad_post(self, g: torch.Tensor, shape, mode=None, momentum=None, decay=1.0, frame_major=False) -> torch.Tensor:
        """The step between the input gradient and the sign step, fused (`i2v_grad_post_f32`): mean-abs (`'frame'` / `'clip'` /
        `'column'`) or L1 (`'l1'`) normalisation, `+ decay * momentum` with the momentum updated in place, and -- `frame_major` -- the
        change from the backbone's (b*f,c,h,w) gradient to the clip layout.  Returns the (b,c,f,h,w) gradient."""
        b, c, f, h, w = (int(v) for v in shape)
        m = self.GRAD_POST_MODES[mode]
        g = g.contiguous()
        assert g.numel() == b * c * f * h * w, (tuple(g.shape), (b, c, f, h, w))
        assert momentum is None or (momentum.is_contiguous() and tuple(momentum.shape) == (b, c, f, h, w))
        out = torch.empty(b, c, f, h, w, dtype=torch.float32, device=g.device)
        nbytes = int(self.capi.i2v_grad_post_scratch_bytes(b, c, f, h, w, m))
        scratch = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=g.device)
        _lib.check(self.capi, self.capi.i2v_grad_post_f32(_ptr(g, self), _ptr(momentum, self) if momentum is not None else None, _ptr(out, self),
                                                          b, c, f, h, w, 1 if frame_major else 0, m, float(decay), C.c_void_p(scratch.data_ptr()), self.stream()))
        return out

    def tap_perts(self, adv: torch.Tensor, videos: torch.Tensor) -> torch.Tensor:
        """TAP: `(adv - videos) / std` (`base_attacks.py:138-143`, sic), clip layout."""
        b, c, f, h, w = adv.shape
        out = torch.empty_like(adv)
        _lib.check(self.capi, self.capi.i2v_tap_perts_f32(_ptr(adv, self), _ptr(videos, self), _ptr(out, self), b, c, f, h, w, self.stream()))
        return out

    def tap_sign_abs(self, smooth: torch.Tensor, reg_out: torch.Tensor) -> torch.Tensor:
        """TAP: sign(smooth) and, into the one-element `reg_out`, sum|smooth| (`base_attacks.py:731`)."""
        sg = torch.empty_like(smooth)
        scratch = torch.empty(int(self.capi.i2v_tap_scratch_bytes(smooth.numel())), dtype=torch.uint8, device=smooth.device)
        _lib.check(self.capi, self.capi.i2v_tap_sign_abs_f32(_ptr(smooth, self), _ptr(sg, self), _ptr(reg_out, self), smooth.numel(),
                                                            C.c_void_p(scratch.data_ptr()), self.stream()))
        return sg

    def tap_grad(self, gx: torch.Tensor, boxsign: torch.Tensor, weight: float) -> torch.Tensor:
        """TAP: clip-layout gradient = frame-major backbone gradient + weight * boxsign / std."""
        b, c, f, h, w = boxsign.shape
        out = torch.empty_like(boxsign)
        _lib.check(self.capi, self.capi.i2v_tap_grad_f32(_ptr(gx, self), _ptr(boxsign, self), _ptr(out, self), b, c, f, h, w, float(weight), self.stream()))
        return out

    def aens_coeffs(self, prev, coeffs, momentum):
        _lib.check(self.capi, self.capi.i2v_aens_coeffs_f32(_ptr(prev, self), _ptr(coeffs, self), momentum, coeffs.numel(), self.stream()))

    def aens_reduce(self, cos, coeffs, feat_sum, weighted):
        L, n = cos.shape
        _lib.check(self.capi, self.capi.i2v_aens_reduce_f32(_ptr(cos, self), _ptr(coeffs, self), L, n, _ptr(feat_sum, self), _ptr(weighted, self), self.stream()))


class HookInfo:
    __slots__ = ("act", "act_stride", "grad", "grad_stride", "D", "post_relu", "shape", "T")


class Net:
    """A planned backbone truncated at its deepest hook."""

    def __init__(self, eng: Engine, graph: Graph, sd, hook_tensors: List[int], max_frames: int, relu_gain=None):
        """`relu_gain`: {ReLU-output tensor of `graph`: backward gain} (`i2v_net_set_relu_gain`; the Skip Gradient Method)."""
        self.eng, capi, h = eng, eng.capi, eng.h
        self.graph = g = graph.truncated(hook_tensors)
        self.max_frames = max_frames
        nid = C.c_int()
        _lib.check(capi, capi.i2v_net_create(h, C.byref(nid)))
        self.id = nid.value
        # buffers / tensors actually referenced
        used_t = {g.input}
        for nd in g.nodes:
            used_t.update([nd.src, nd.dst])
            used_t.update(getattr(nd, "extra_srcs", ()))
            if getattr(nd, "residual", None) is not None:
                used_t.add(nd.residual)
        used_t.update(hook_tensors)
        self.buf_id, self.ten_id = {}, {}
        for t in sorted(used_t):
            ts = g.tensors[t]
            if ts.buf not in self.buf_id:
                out = C.c_int()
                if g.video:
                    _lib.check(capi, capi.i2v_net_add_buffer3d(h, self.id, g.buffers[ts.buf], ts.T, ts.H, ts.W, C.byref(out)))
                else:
                    _lib.check(capi, capi.i2v_net_add_buffer(h, self.id, g.buffers[ts.buf], ts.H, ts.W, C.byref(out)))
                self.buf_id[ts.buf] = out.value
            out = C.c_int()
            _lib.check(capi, capi.i2v_net_add_tensor(h, self.id, self.buf_id[ts.buf], ts.c_off, ts.C,
                                                     1 if ts.post_relu else 0, C.byref(out)))
            self.ten_id[t] = out.value
        _lib.check(capi, capi.i2v_net_set_input(h, self.id, self.ten_id[g.input]))
        for nd in g.nodes:
            if nd.op == "conv":
                w = sd[nd.weight].float().cpu().contiguous()
                scale, shift = fold_affine(nd, sd)
                scale, shift = scale.contiguous(), shift.contiguous()
                if g.video:
                    d = _lib.Conv3dDesc(self.ten_id[nd.src], self.ten_id[nd.dst], nd.cin, nd.cout, nd.kt, nd.kh, nd.kw,
                                        nd.stride_t, nd.stride, nd.pad_t, nd.pad, nd.dil_t, 1 if nd.relu else 0,
                                        -1 if nd.residual is None else self.ten_id[nd.residual])
                    _lib.check(capi, capi.i2v_net_add_conv3d(h, self.id, C.byref(d), _hptr(w), _hptr(scale), _hptr(shift)))
                    continue
                d = _lib.ConvDesc(self.ten_id[nd.src], self.ten_id[nd.dst], nd.cin, nd.cout, nd.kh, nd.kw,
                                  nd.stride, nd.pad, 1 if nd.relu else 0,
                                  -1 if nd.residual is None else self.ten_id[nd.residual])
                if nd.pre_bn:
                    ps, pt = fold_pre_affine(nd, sd)
                    ps, pt = ps.contiguous(), pt.contiguous()
                    _lib.check(capi, capi.i2v_net_add_conv_preact(h, self.id, C.byref(d), _hptr(w), _hptr(scale), _hptr(shift),
                                                                  _hptr(ps), _hptr(pt)))
                else:
                    _lib.check(capi, capi.i2v_net_add_conv(h, self.id, C.byref(d), _hptr(w), _hptr(scale), _hptr(shift)))
            elif nd.op == "attention":
                d = _lib.AttnDesc(self.ten_id[nd.src], self.ten_id[nd.phi], self.ten_id[nd.g], self.ten_id[nd.dst], float(nd.scale))
                _lib.check(capi, capi.i2v_net_add_attention(h, self.id, C.byref(d)))
            elif g.video:
                d = _lib.Pool3dDesc(self.ten_id[nd.src], self.ten_id[nd.dst], nd.kt, nd.k, nd.stride_t, nd.stride, nd.pad_t, nd.pad)
                _lib.check(capi, capi.i2v_net_add_maxpool3d(h, self.id, C.byref(d)))
            else:
                d = _lib.PoolDesc(self.ten_id[nd.src], self.ten_id[nd.dst], nd.k, nd.stride, nd.pad)
                add = capi.i2v_net_add_avgpool if nd.op == "avgpool" else capi.i2v_net_add_maxpool
                _lib.check(capi, add(h, self.id, C.byref(d)))
        for t, gain in (relu_gain or {}).items():
            if t in self.ten_id:                  # (a ReLU behind the deepest hook is not part of the truncated graph)
                _lib.check(capi, capi.i2v_net_set_relu_gain(h, self.id, self.ten_id[t], float(gain)))
        hooks = (C.c_int * len(hook_tensors))(*[self.ten_id[t] for t in hook_tensors])
        _lib.check(capi, capi.i2v_net_plan(h, self.id, hooks, len(hook_tensors), max_frames))
        self.hook_tensors = hook_tensors
        self.hooks = [self._hook_info(i) for i in range(len(hook_tensors))]

    def _hook_info(self, i) -> HookInfo:
        act, grad = C.c_void_p(), C.c_void_p()
        a_s, g_s, D, pr = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        _lib.check(self.eng.capi, self.eng.capi.i2v_net_hook_info(
            self.eng.h, self.id, i, C.byref(act), C.byref(a_s), C.byref(grad), C.byref(g_s), C.byref(D), C.byref(pr)))
        hi = HookInfo()
        hi.act, hi.act_stride, hi.grad, hi.grad_stride = act.value, a_s.value, grad.value, g_s.value
        hi.D, hi.post_relu = D.value, pr.value
        ts = self.graph.tensors[self.hook_tensors[i]]
        hi.shape = (ts.C, ts.H, ts.W)
        hi.T = ts.T                     # frames per clip at the hook (video backbones)
        return hi

    def close(self):
        """Release the arena and packed weights now (they also go when the engine is destroyed)."""
        if self.id is not None and self.eng.h:
            with self.eng.plan_lock:
                self.eng.capi.i2v_net_destroy(self.eng.h, self.id)
        self.id = None

    def __del__(self):            # an attack object going out of scope gives its arenas back (tens of GB at full size)
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def workspace_bytes(self) -> int:
        return self.eng.capi.i2v_net_workspace_bytes(self.eng.h, self.id)

    def fusion_info(self):
        """(eligible forward pairs, eligible backward pairs, fused forward pairs, fused backward pairs) of the planned launch lists
        (`i2v_net_fusion_info`): 3x3 convolution + the pointwise convolution over its output as one launch."""
        out = (C.c_int32 * 4)()
        _lib.check(self.eng.capi, self.eng.capi.i2v_net_fusion_info(self.eng.h, self.id, out))
        return tuple(int(v) for v in out)

    def forward(self, x: torch.Tensor):
        _lib.check(self.eng.capi, self.eng.capi.i2v_net_forward(self.eng.h, self.id, _ptr(x, self.eng), x.shape[0], self.eng.stream()))

    def backward(self, gx: torch.Tensor, accumulate=False):
        _lib.check(self.eng.capi, self.eng.capi.i2v_net_backward(self.eng.h, self.id, _ptr(gx, self.eng), 1 if accumulate else 0,
                                                                 self.eng.stream()))

    def read_tensor(self, tid: int, frames: int, grad=False) -> torch.Tensor:
        ts = self.graph.tensors[tid]
        out = torch.empty(frames, ts.C, ts.H, ts.W, dtype=torch.float32, device=self.eng.device)
        _lib.check(self.eng.capi, self.eng.capi.i2v_net_read_tensor(self.eng.h, self.id, self.ten_id[tid], 1 if grad else 0,
                                                                    _ptr(out, self.eng), frames, self.eng.stream()))
        return out

    def save_hook(self, i: int, frames: int) -> torch.Tensor:
        """Detached copy of hook i's feature (the clean `init_feature_maps`, image_attacks.py:319-323)."""
        return self.read_tensor(self.hook_tensors[i], frames)

    def cossim(self, i: int, init: torch.Tensor, cos_out: torch.Tensor, scratch: torch.Tensor, frames: int,
               coef_dev=None, coef_index=0, coef_host=1.0):
        """cos of hook i against its clean feature, gradient written into the hook's gradient view."""
        hi = self.hooks[i]
        capi = self.eng.capi
        _lib.check(capi, capi.i2v_cossim_fwd_bwd_f32(
            C.c_void_p(hi.act), hi.act_stride, _ptr(init, self.eng), hi.D, hi.D, frames,
            C.c_void_p(coef_dev.data_ptr()) if coef_dev is not None else C.c_void_p(0), coef_index, coef_host,
            hi.post_relu, 0, C.c_void_p(cos_out.data_ptr()), C.c_void_p(hi.grad), hi.grad_stride,
            C.c_void_p(scratch.data_ptr()), self.eng.stream()))

    def stdloss(self, i: int, std_out: torch.Tensor, scratch: torch.Tensor, frames: int, exchange=None):
        """Unbiased std of hook i over ALL frames and its gradient.  `exchange(sums, count)` -- given the
        (2,) float64 device tensor of local (sum, sum of squares) and the local element count -- may
        all-reduce the sums in place and return the global count (clip-sharded runs)."""
        hi = self.hooks[i]
        capi = self.eng.capi
        _lib.check(capi, capi.i2v_std_reduce_f32(C.c_void_p(hi.act), hi.act_stride, hi.D, frames,
                                                 C.c_void_p(scratch.data_ptr()), self.eng.stream()))
        total = frames * hi.D
        if exchange is not None:
            total = exchange(scratch[:16].view(torch.float64), total)
        _lib.check(capi, capi.i2v_std_grad_f32(
            C.c_void_p(hi.act), hi.act_stride, hi.D, frames, total, hi.post_relu, 0, C.c_void_p(std_out.data_ptr()),
            C.c_void_p(hi.grad), hi.grad_stride, C.c_void_p(scratch.data_ptr()), self.eng.stream()))

    def hook_frames(self, i: int, in_frames: int) -> int:
        """Frames hook i holds when `in_frames` input frames were run (video backbones change the clip length)."""
        return in_frames // self.graph.tensors[self.graph.input].T * self.hooks[i].T

    def ilaf_scratch_bytes(self, i: int, frames: int, frames_per_seg: int = 0) -> int:
        return self.eng.capi.i2v_ilaf_scratch_bytes(self.hooks[i].D, frames, frames_per_seg)

    def ilaf_reduce(self, i: int, ori: torch.Tensor, adv0: torch.Tensor, scratch: torch.Tensor, frames: int, act=None,
                    frames_per_seg: int = 0):
        """(sum d*d, sum d*d0) of hook i against the dense clean / initial-adversarial feature copies, left as two
        doubles at the start of `scratch`.  `act` overrides the activation (used once to measure |d0|).  `frames_per_seg`:
        independent segments of that many frames (one clip each), two doubles per segment."""
        hi = self.hooks[i]
        capi = self.eng.capi
        a, a_s = (C.c_void_p(hi.act), hi.act_stride) if act is None else (_ptr(act, self.eng), hi.D)
        if frames_per_seg:
            _lib.check(capi, capi.i2v_ilaf_reduce_seg_f32(a, a_s, _ptr(ori, self.eng), _ptr(adv0, self.eng), hi.D, frames, frames_per_seg,
                                                          C.c_void_p(scratch.data_ptr()), self.eng.stream()))
            return
        _lib.check(capi, capi.i2v_ilaf_reduce_f32(a, a_s, _ptr(ori, self.eng), _ptr(adv0, self.eng), hi.D, frames,
                                                  C.c_void_p(scratch.data_ptr()), self.eng.stream()))

    def ilaf_grad(self, i: int, ori, adv0, init_norm, loss_out: torch.Tensor, scratch: torch.Tensor, frames: int,
                  frames_per_seg: int = 0):
        """`init_norm`: |d0| as a host float -- or, with `frames_per_seg`, a DEVICE tensor of float64 holding |d0|^2 per segment;
        `loss_out` then has one element per segment."""
        hi = self.hooks[i]
        capi = self.eng.capi
        if frames_per_seg:
            assert init_norm.dtype == torch.float64 and init_norm.numel() == frames // frames_per_seg
            _lib.check(capi, capi.i2v_ilaf_grad_seg_f32(C.c_void_p(hi.act), hi.act_stride, _ptr(ori, self.eng), _ptr(adv0, self.eng), hi.D, frames,
                                                        frames_per_seg, C.c_void_p(init_norm.data_ptr()), hi.post_relu, 0,
                                                        C.c_void_p(loss_out.data_ptr()), C.c_void_p(hi.grad), hi.grad_stride,
                                                        C.c_void_p(scratch.data_ptr()), self.eng.stream()))
            return
        _lib.check(capi, capi.i2v_ilaf_grad_f32(C.c_void_p(hi.act), hi.act_stride, _ptr(ori, self.eng), _ptr(adv0, self.eng), hi.D, frames,
                                                init_norm, hi.post_relu, 0, C.c_void_p(loss_out.data_ptr()),
                                                C.c_void_p(hi.grad), hi.grad_stride, C.c_void_p(scratch.data_ptr()),
                                                self.eng.stream()))

    def tap_distance(self, i: int, clean: torch.Tensor, coef: float, dist_out: torch.Tensor, scratch: torch.Tensor, frames: int,
                     frames_per_seg: int):
        """TAP's feature-distance term over hook i (`i2v_tap_distance_f32`): per clip || r(a) - r(clean) ||_2 into `dist_out`, and
        coef * its gradient into the hook's gradient view."""
        hi = self.hooks[i]
        _lib.check(self.eng.capi, self.eng.capi.i2v_tap_distance_f32(
            C.c_void_p(hi.act), hi.act_stride, _ptr(clean, self.eng), hi.D, frames, frames_per_seg, float(coef), hi.post_relu, 0,
            C.c_void_p(dist_out.data_ptr()), C.c_void_p(hi.grad), hi.grad_stride, C.c_void_p(scratch.data_ptr()), self.eng.stream()))

    def head_ce(self, i, W: torch.Tensor, bias, labels: torch.Tensor, in_frames: int, scale: float, logits: torch.Tensor,
                loss_each: torch.Tensor, scratch: torch.Tensor):
        """Classifier head over hook i -- or over a LIST of hooks whose pooled features are concatenated in that order
        (SlowFast) -- : global average pool over the clip -> Linear -> cross-entropy, mean over clips, and
        `scale * d loss / d feature` into every hook's gradient view (base_attacks.py:282-284)."""
        capi = self.eng.capi
        clips = in_frames // self.graph.tensors[self.graph.input].T
        bias_p = _ptr(bias, self.eng) if bias is not None else C.c_void_p(0)
        assert labels.dtype == torch.int32 and labels.numel() == clips
        if isinstance(i, int):
            hi = self.hooks[i]
            C_, H_, W_ = hi.shape
            assert W.shape[1] == C_
            _lib.check(capi, capi.i2v_head_ce_f32(
                C.c_void_p(hi.act), hi.act_stride, C_, H_ * W_, hi.T, clips, _ptr(W, self.eng), bias_p,
                W.shape[0], C.c_void_p(labels.data_ptr()), scale, hi.post_relu, 0, _ptr(logits, self.eng), _ptr(loss_each, self.eng), C.c_void_p(hi.grad),
                hi.grad_stride, C.c_void_p(scratch.data_ptr()), self.eng.stream()))
            return
        his = [self.hooks[k] for k in i]
        Ctot = sum(hi.shape[0] for hi in his)
        assert W.shape[1] == Ctot and scratch.numel() >= capi.i2v_head_scratch_bytes(Ctot, clips)
        sp, st = C.c_void_p(scratch.data_ptr()), self.eng.stream()
        off = 0
        for hi in his:
            _lib.check(capi, capi.i2v_head_pool_f32(C.c_void_p(hi.act), hi.act_stride, hi.shape[0], hi.shape[1] * hi.shape[2], hi.T, clips,
                                                    Ctot, off, sp, st))
            off += hi.shape[0]
        _lib.check(capi, capi.i2v_head_logits_ce_f32(Ctot, clips, _ptr(W, self.eng), bias_p, W.shape[0], C.c_void_p(labels.data_ptr()), scale,
                                                     _ptr(logits, self.eng), _ptr(loss_each, self.eng), sp, st))
        off = 0
        for hi in his:
            _lib.check(capi, capi.i2v_head_grad_f32(C.c_void_p(hi.act), hi.act_stride, hi.shape[0], hi.shape[1] * hi.shape[2], hi.T, clips,
                                                    Ctot, off, hi.post_relu, 0, C.c_void_p(hi.grad), hi.grad_stride, sp, st))
            off += hi.shape[0]

    def head_logits(self, i, W: torch.Tensor, bias, in_frames: int, logits: torch.Tensor, scratch: torch.Tensor):
        """Forward-only classifier head (the evaluator, `reference.py:108-129`): global average pool of hook i -- or of a LIST of
        hooks, concatenated in that order -- then `fc`; no loss gradient is written.  `logits`: (clips, K) on the device."""
        capi = self.eng.capi
        his = [self.hooks[k] for k in (i if isinstance(i, (list, tuple)) else [i])]
        clips = in_frames // self.graph.tensors[self.graph.input].T
        Ctot = sum(hi.shape[0] for hi in his)
        assert W.shape[1] == Ctot and scratch.numel() >= capi.i2v_head_scratch_bytes(Ctot, clips)
        sp, st = C.c_void_p(scratch.data_ptr()), self.eng.stream()
        off = 0
        for hi in his:
            _lib.check(capi, capi.i2v_head_pool_f32(C.c_void_p(hi.act), hi.act_stride, hi.shape[0], hi.shape[1] * hi.shape[2], hi.T, clips,
                                                    Ctot, off, sp, st))
            off += hi.shape[0]
        labels = torch.zeros(clips, dtype=torch.int32, device=logits.device)
        loss_each = torch.empty(clips, dtype=torch.float32, device=logits.device)
        bias_p = _ptr(bias, self.eng) if bias is not None else C.c_void_p(0)
        _lib.check(capi, capi.i2v_head_logits_ce_f32(Ctot, clips, _ptr(W, self.eng), bias_p, W.shape[0], C.c_void_p(labels.data_ptr()), 1.0,
                                                     _ptr(logits, self.eng), _ptr(loss_each, self.eng), sp, st))

    def scratch_bytes(self, frames: int) -> int:
        return max(self.eng.capi.i2v_cossim_scratch_bytes(hi.D, frames) for hi in self.hooks)
