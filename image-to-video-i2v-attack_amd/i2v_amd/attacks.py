"""Attack classes with the reference's names, constructor signatures and call protocol
(SURVEY.md section 8(b)); the per-frame arithmetic runs in `libi2v_hip.so`.

    ImageGuidedFMDirection_Adam(model_name_lists, depth, step_size, epsilon=16/255, steps=10)   # I2V
    ImageGuidedStd_Adam(model_name_lists, depth, step_size, epsilon=16/255, steps=10)           # DR
    ImageGuidedFML2_Adam_MultiModels(model_name_lists, depths, epsilon=16/255, steps=60)        # ENS-I2V
    AENS_I2V_MF(model_name_lists, depths, step_size, momentum=0, coef_CE=False, ...)            # adaptive ENS
    adv = attack(videos, labels, video_names)          # videos (b,3,f,h,w) ImageNet-normalised

Differences from the reference that do not change results: nothing behind the deepest hook is
executed, no weight gradients are formed, the per-step cost is kept on the device and read back
once per call (the reference syncs twice per step, image_attacks.py:349,358), and the returned clip
is a contiguous (b,3,f,h,w) tensor (the reference returns a permuted view of the same values).
"""
import copy
import os
import threading
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import graphs as _graphs
from . import weights as _weights
from .engine import Engine

_ENGINES: Dict[str, Engine] = {}


def default_device() -> str:
    return f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"


def get_engine(device: Optional[str] = None) -> Engine:
    """Process-wide engine per device (the reference's single `.cuda()` device)."""
    device = device or default_device()
    if device not in _ENGINES:
        _ENGINES[device] = Engine(device)
    return _ENGINES[device]


def shutdown():
    """Orderly end of a CLI process: wait for the device, then release every engine (planned nets, arenas, event pools)
    while the HIP runtime is certainly still alive -- instead of leaving it to interpreter finalisation, whose order
    against the runtime's own teardown is not defined."""
    for dev, eng in list(_ENGINES.items()):
        try:
            if eng.device.type == "cuda":
                torch.cuda.synchronize(eng.device)
            eng.close()
        finally:
            _ENGINES.pop(dev, None)


class Attack(object):
    """Base class, `/root/reference/image_attacks.py:12-82`: attack name, ImageNet mean/std,
    `__call__ -> forward`."""

    def __init__(self, name, model=None):
        self.attack = name
        self.model = model
        self.model_name = str(model).split("(")[0]
        self.mean = [0.485, 0.456, 0.406]
        self.std = [0.229, 0.224, 0.225]

    def forward(self, *input):
        raise NotImplementedError

    def __call__(self, *input, **kwargs):
        return self.forward(*input, **kwargs)


class _ImageGuided(Attack):
    """Shared loop of the image-model-guided attacks."""
    _mode = "i2v"

    def _setup(self, model_names: Sequence[str], depths_per_model: List[List[int]], engine, graph_builder,
               weight_seed, whole_module: Optional[List[bool]] = None):
        self.model_names = list(model_names)
        self._depths = depths_per_model
        # per model: hook the whole module (list depths of the adaptive attack, TPAMI_attack.py:176-200) instead of
        # the scalar-depth lookup's tensor -- the two differ for SqueezeNet (Fire output vs its 3x3 branch)
        self._whole = list(whole_module) if whole_module is not None else [False] * len(self.model_names)
        self._engine = engine
        self._builder = graph_builder or _graphs.build
        self._wseed = weight_seed
        self._nets = None
        self._net_key = None
        self.loss_info = {}
        # validate names / hook depths now, as the reference does at construction
        # (unknown model -> UnboundLocalError image_attacks.py:103; densenet -> AttributeError :291)
        for m, ds in zip(self.model_names, depths_per_model):
            g = self._builder(m, (224, 224))
            for d in ds:
                if d not in g.hooks:
                    raise KeyError(d)

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def _get_nets(self, frames: int, hw):
        key = (hw, )
        if self._nets is not None and self._net_key == key and self._max_frames >= frames:
            return self._nets
        for old in (self._nets or []):          # re-planning for a larger batch / other resolution
            old.close()
        nets = []
        for m, ds, whole in zip(self.model_names, self._depths, self._whole):
            g = self._builder(m, hw)
            sd = _weights.load_state_dict(g, self._wseed)
            nets.append(self.engine.build_net(g, sd, [g.hook_for(d, whole) for d in ds], frames))
        self._nets, self._net_key, self._max_frames = nets, key, frames
        return nets

    # hooks for the adaptive subclass
    def _begin(self, L, dev):
        pass

    def _run(self, videos: torch.Tensor, video_names, state=None, steps=None):
        """`state` / `steps` serve teacher forcing only (`forced_step`): start from a given (delta, exp_avg,
        exp_avg_sq) after `t` Adam steps -- and, for the adaptive attack, given coefficients -- and run `steps`
        iterations instead of `self.steps`."""
        eng = self.engine
        dev = eng.device
        videos = videos.detach().to(device=dev, dtype=torch.float32).contiguous()
        b, c, f, h, w = videos.shape
        N = b * f
        nets = self._get_nets(N, (h, w))
        L = sum(len(n.hooks) for n in nets)
        steps, eps = (self.steps if steps is None else int(steps)), float(self.epsilon)
        mode = self._mode
        t0 = 0
        kw = dict(dtype=torch.float32, device=dev)
        x = torch.empty(N, 3, h, w, **kw)
        u = torch.empty_like(x)
        eng.frames_from_video(videos, x, u)                             # image_attacks.py:300-301,308
        delta = torch.full((N, 3, h, w), 0.01 / 255, **kw)              # :304
        m = torch.zeros_like(delta)
        v = torch.zeros_like(delta)
        gx = torch.empty_like(delta)
        xadv = torch.empty_like(delta)
        if state is not None:
            delta.copy_(state["delta"].reshape(delta.shape))
            m.copy_(state["m"].reshape(delta.shape))
            v.copy_(state["v"].reshape(delta.shape))
            t0 = int(state["t"])
        scratch = torch.empty(max(n.scratch_bytes(N) for n in nets), dtype=torch.uint8, device=dev)
        init = []
        if mode != "std":
            for net in nets:                                            # clean pass on the RAW input, :318-323
                net.forward(x)
                init.append([net.save_hook(i, N) for i in range(len(net.hooks))])
        vals = torch.zeros(max(steps, 1), L, N if mode != "std" else 1, **kw)
        aens = mode == "aens"
        if aens:
            if self.coeffs is None or self.coeffs.device != dev:
                self.coeffs = torch.ones(L, **kw)                       # TPAMI_attack.py:165
            prev = torch.ones(L, **kw)                                  # :257
            # per step ONE contiguous pair [sum_frames cos_l | coeff_l * sum_frames cos_l]: what `aens_reduce_kernel` writes is what
            # the all-reduce works on in place and what the next step's `aens_coeffs_kernel` reads -- no pack / unpack kernels between
            exch = torch.empty(steps, 2, L, **kw)
            weighted = exch[:, 1]
            wts = torch.empty(steps, L, **kw)
        begin = time.time()
        for i in range(steps):
            if aens:
                if state is not None and i == 0 and state.get("coeffs") is not None:
                    self.coeffs.copy_(state["coeffs"].to(dev))             # forced coefficients of this step
                else:
                    eng.aens_coeffs(prev, self.coeffs, float(self.momentum))   # :265
                wts[i].copy_(self.coeffs)
            eng.compose(u, delta, xadv, b, f, eps)                      # image_attacks.py:331-332
            l = 0
            for n, net in enumerate(nets):
                net.forward(xadv)                                       # :334
                for k in range(len(net.hooks)):
                    if mode == "std":
                        net.stdloss(k, vals[i, l], scratch, N, self._std_exchange())   # :218
                    elif aens:
                        net.cossim(k, init[n][k], vals[i, l], scratch, N, coef_dev=self.coeffs, coef_index=l,
                                   coef_host=1.0 / L)                   # TPAMI_attack.py:289-291
                    else:
                        net.cossim(k, init[n][k], vals[i, l], scratch, N)   # image_attacks.py:341-347
                    l += 1
                net.backward(gx, accumulate=n > 0)                      # :352 (input gradient only)
            if aens:
                eng.aens_reduce(vals[i], self.coeffs, exch[i, 0], exch[i, 1])
                self._exchange(exch[i])                                  # sharded runs: 2L floats summed over the ranks, in place
                prev = exch[i, 1] if self.coef_CE else exch[i, 0]        # TPAMI_attack.py:293-297 (this step's row is not written again)
            eng.adam_step(delta, m, v, gx, u, eps, float(self.step_size), t0 + i + 1)   # :351-353
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        self.used_time = time.time() - begin
        out = torch.empty(b, 3, f, h, w, **kw)
        eng.compose(u, delta, out, b, f, eps, video_layout=True)        # :360-364
        # one read-back for the whole call
        if aens:
            costs = weighted.mean(dim=1).cpu().numpy().astype(np.float32)   # TPAMI_attack.py:291
            self.weights = [wts[i].cpu().numpy() for i in range(steps)]     # :266
        elif mode == "std":
            costs = vals[:steps].sum(dim=(1, 2)).cpu().numpy().astype(np.float32)   # image_attacks.py:218 (one value per step)
        else:
            costs, self.last_clip_costs = _canonical_costs(vals[:steps].cpu().numpy(), b, f)   # image_attacks.py:347
        self.last_costs = costs
        self.last_values = vals
        for vid_name in video_names:                                    # :355-358 (batch-total cost per name)
            if vid_name not in self.loss_info.keys():
                self.loss_info[vid_name] = {}
            for i in range(steps):
                self.loss_info[vid_name][i] = {"cost": str(costs[i])}
        self._delta, self._m, self._v = delta, m, v
        self._gx = gx                        # d cost / d composed frames of the LAST iteration (before the compose backward)
        if aens:
            self._prev = prev.clone()        # previous_cs_loss after the last iteration (TPAMI_attack.py:293-297)
        return out

    def forced_step(self, videos, delta, m, v, t, coeffs=None):
        """Teacher forcing (tests / diagnostics): ONE iteration of the loop from the optimiser state `(delta, m, v)`
        reached after `t` Adam steps (`image_attacks.py:325-358`, `TPAMI_attack.py:258-312`), always in one lane.
        Returns `(delta', m', v', cost)`.  `coeffs` forces the adaptive attack's layer weights for this iteration."""
        saved = dict(self.loss_info)
        try:
            self._run(videos, [], state=dict(delta=delta, m=m, v=v, t=t, coeffs=coeffs), steps=1)
        finally:
            self.loss_info = saved
        return self._delta, self._m, self._v, float(self.last_costs[0])

    def _exchange(self, pair):
        pass

    def _std_exchange(self):
        return None

    # ---- clip lanes -------------------------------------------------------------------------------------------
    # Frames are independent in I2V / ENS-I2V (the loss is a sum of per-frame terms, Adam is elementwise), so a batch
    # may be cut into lanes of whole clips that run CONCURRENTLY on separate HIP streams, each with its own planned
    # nets: the tails and under-filled launches of one lane overlap with the other's.  Measured on the headline
    # workload: 695 -> 734 frames/s with 2 lanes of 2 clips (tools/lanes_fresh.py).  The perturbed clips are
    # bit-identical to the single-lane run; only the reported batch cost is summed in a different order.
    clip_lanes = None          # None: $I2V_CLIP_LANES (default 2) on a GPU engine, 1 on the host simulation

    def _lane_count(self, b, f=1):
        """Lanes for a batch of b clips of f frames: whole clips per lane.  A single clip (the reference CLI's default
        `--batch_size 1`) is cut along its frames only when lanes are asked for explicitly (`clip_lanes`): frame lanes paid with the
        round-2 kernels (519 -> 548 frames/s) and no longer do with round 3's -- fresh processes, 1 / 2 / 3 frame lanes: 577 / 569 / 550
        (tools/lanes_single.py)."""
        if self._mode != "i2v":
            return 1
        n = self.clip_lanes
        if n is None and b == 1:
            return 1
        if n is None:
            # Two lanes, whatever the batch.  Measured on the headline shape in fresh processes (tools/lanes_fresh.py, frames/s with
            # 1 / 2 / 4 lanes): 695 / 734 / 630 -- the HIP runtime multiplexes a process's streams onto 4 hardware queues by
            # default, and with 4 lane streams beside torch's own two lanes end up sharing a queue and serialise (with
            # GPU_MAX_HW_QUEUES=8: 694 / 734 / 731, nothing gained over two).  8 clips: 717 / 737 / 734; 1 clip: 495 / 504 / 488.
            n = int(os.environ.get("I2V_CLIP_LANES", "2")) if self.engine.device.type == "cuda" else 1
        units = b if b > 1 else f // 8            # a lane of fewer than 8 frames does not pay
        return max(1, min(int(n), units))

    def _make_lanes(self, n_lanes):
        if len(getattr(self, "_lanes", [])) != n_lanes:
            self._lanes = []
            for _ in range(n_lanes):
                lane = copy.copy(self)
                lane._nets, lane._net_key, lane.loss_info, lane._lanes = None, None, {}, []
                self._lanes.append(lane)

    def reserve(self, clips, frames, hw):
        """Plan (pack, upload, autotune) now for the LARGEST batch the caller will send -- `clips` clips of `frames` frames at
        `hw` -- so that a run whose batches grow (a CLI grouping whatever its loader has ready) never re-plans on the way."""
        n_lanes = self._lane_count(clips, frames)
        if n_lanes > 1:
            self._make_lanes(n_lanes)
            share = -(-clips // n_lanes) * frames if clips > 1 else -(-frames // n_lanes)
            for lane in self._lanes:
                lane._get_nets(share, tuple(hw))
        else:
            self._get_nets(clips * frames, tuple(hw))

    def _run_lanes(self, videos, video_names, n_lanes):
        eng = self.engine
        dev = eng.device
        cuda = dev.type == "cuda"
        b, _, f, h, w = videos.shape
        video_names = list(video_names)
        self._make_lanes(n_lanes)
        for lane in self._lanes:                        # the caller may have changed these between calls (legal on the
            lane.steps, lane.step_size, lane.epsilon = self.steps, self.step_size, self.epsilon    # reference classes)
        by_frames = b == 1                              # one clip: the lanes take frame ranges of it
        total = f if by_frames else b
        cuts = [(total * k) // n_lanes for k in range(n_lanes + 1)]

        def part_of(k):
            if by_frames:
                return videos[:, :, cuts[k]:cuts[k + 1]].contiguous(), video_names
            return videos[cuts[k]:cuts[k + 1]], video_names[cuts[k]:cuts[k + 1]]
        for k, lane in enumerate(self._lanes):          # plan (and autotune) one after the other, before anything runs
            lane._get_nets((cuts[k + 1] - cuts[k]) * (1 if by_frames else f), (h, w))
        if cuda:
            torch.cuda.current_stream(dev).synchronize()        # the lanes read `videos` on their own streams
        outs, errors = [None] * n_lanes, []

        def work(k):
            try:
                lane = self._lanes[k]
                if cuda:
                    torch.cuda.set_device(dev)          # per-thread state
                    with torch.cuda.stream(self._lane_streams[k]):
                        outs[k] = lane._run(*part_of(k))
                else:
                    outs[k] = lane._run(*part_of(k))
            except BaseException as e:
                errors.append(e)
        if cuda and len(getattr(self, "_lane_streams", [])) != n_lanes:
            self._lane_streams = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
        begin = time.time()
        threads = [threading.Thread(target=work, args=(k,)) for k in range(n_lanes)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        self.used_time = time.time() - begin
        # the lanes' per-frame loss values side by side in frame order = what one lane would have produced: the reported cost
        # does not depend on the split either
        vals = np.concatenate([lane.last_values[:self.steps].cpu().numpy() for lane in self._lanes], axis=2)
        costs, self.last_clip_costs = _canonical_costs(vals, b, f)
        self.last_costs = costs
        self._delta = torch.cat([lane._delta for lane in self._lanes])      # frame order: (clip, frame) in both splits
        for vid_name in video_names:                                    # image_attacks.py:355-358 (batch-total cost per name)
            self.loss_info.setdefault(vid_name, {})
            for i in range(self.steps):
                self.loss_info[vid_name][i] = {"cost": str(costs[i])}
        return torch.cat(outs, dim=2 if by_frames else 0)

    def forward(self, videos, labels, video_names):
        n_lanes = self._lane_count(videos.shape[0], videos.shape[2])
        if n_lanes > 1:
            return self._run_lanes(videos.detach().to(device=self.engine.device, dtype=torch.float32), video_names, n_lanes)
        return self._run(videos, video_names)

    def forward_grouped(self, batches):
        """Several loader batches `(videos, labels, video_names)` in ONE engine call (the reference CLI's default is
        `--batch_size 1`: a single 32-frame clip leaves the 14x14 layers with 1.5 blocks per CU).  Frames are independent
        in I2V / ENS-I2V and no kernel's summation order depends on the batch, so every clip comes out bit-identical to its
        own call; `loss_info` gets, per batch, the cost of THAT batch's frames (what its own call would have logged).
        Returns the list of per-batch outputs.  Attacks that couple the batch (adaptive ENS, DR) run batch by batch."""
        batches = list(batches)
        if self._mode != "i2v" or len(batches) == 1:
            return [self(v, l, n) for v, l, n in batches]
        sizes = [int(v.shape[0]) for v, _, _ in batches]
        dev = self.engine.device
        videos = torch.cat([v.detach().to(device=dev, dtype=torch.float32) for v, _, _ in batches])
        labels = torch.cat([torch.as_tensor(l).reshape(-1) for _, l, _ in batches])
        names = [n for _, _, ns in batches for n in ns]
        out = self(videos, labels, names)
        cc, at, outs = self.last_clip_costs, 0, []
        for (v, l, ns), nb in zip(batches, sizes):
            cost = _sum_in_order(cc[:, at:at + nb])
            for n in ns:
                for i in range(self.steps):
                    self.loss_info[n][i] = {"cost": str(cost[i])}
            outs.append(out[at:at + nb])
            at += nb
        return outs


def _sum_in_order(a):
    """fp32 sum over the last axis, left to right (a fixed order, whatever the length)."""
    acc = np.zeros(a.shape[:-1], np.float32)
    for k in range(a.shape[-1]):
        acc = (acc + a[..., k]).astype(np.float32)
    return acc


def _canonical_costs(vals, b, f):
    """`cost = sum(stack(losses))` (image_attacks.py:347) from the per-frame loss values `vals` (steps, L, b*f), in an order
    that does not depend on how the batch was executed: every clip's (L, f) values are summed on their own (numpy's
    pairwise sum of a contiguous run of L*f floats), then the clips left to right.  Returns (costs (steps,), per-clip
    costs (steps, b)) -- clip lanes, grouped loader batches and a plain call all log the same strings."""
    steps, L, N = vals.shape
    per_clip = np.ascontiguousarray(vals.reshape(steps, L, b, f).transpose(0, 2, 1, 3)).reshape(steps, b, L * f)
    clip_costs = per_clip.sum(axis=2, dtype=np.float32)
    return _sum_in_order(clip_costs), clip_costs


class ImageGuidedFMDirection_Adam(_ImageGuided):
    """The I2V attack (`/root/reference/image_attacks.py:236-364`)."""

    def __init__(self, model_name_lists, depth, step_size, epsilon=16 / 255, steps=10, *, engine=None,
                 graph_builder=None, weight_seed=None):
        super().__init__("ImageGuidedFMDirection_Adam")
        self.epsilon, self.steps, self.step_size, self.depth = epsilon, steps, step_size, depth
        self.model_name = model_name_lists[0]
        self._setup(model_name_lists[:1], [[depth]], engine, graph_builder, weight_seed)


class ImageGuidedStd_Adam(_ImageGuided):
    """Dispersion-Reduction baseline (`/root/reference/image_attacks.py:129-234`): minimises the
    unbiased std of the hooked activation over the whole (N,C,H,W) tensor; no clean pass."""
    _mode = "std"

    def __init__(self, model_name_lists, depth, step_size, epsilon=16 / 255, steps=10, *, engine=None,
                 graph_builder=None, weight_seed=None, process_group=None, distributed=None):
        super().__init__("ImageGuidedStd_Adam")
        self.epsilon, self.steps, self.step_size, self.depth = epsilon, steps, step_size, depth
        self.model_name = model_name_lists[0]
        self._setup(model_name_lists[:1], [[depth]], engine, graph_builder, weight_seed)
        self._pg, self._dist = process_group, distributed

    def _std_exchange(self):
        """Clip-sharded runs: the std couples every frame of the GLOBAL batch (image_attacks.py:218), so the
        local (sum, sum of squares, count) are all-reduced once per step -- 3 doubles over RCCL."""
        import torch.distributed as dist
        on = self._dist if self._dist is not None else (dist.is_available() and dist.is_initialized()
                                                          and dist.get_world_size(self._pg) > 1)
        if not on:
            return None

        counts = self.__dict__.setdefault("_global_counts", {})

        def exchange(sums, count):
            # the element count is fixed per (hook, local count): all-reduced once, then no host sync per step
            if count not in counts:
                c = torch.tensor([float(count)], dtype=torch.float64, device=sums.device)
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self._pg)
                counts[count] = int(round(c.item()))
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self._pg)
            return counts[count]
        return exchange


class ImageGuidedFML2_Adam_MultiModels(_ImageGuided):
    """ENS-I2V (`/root/reference/image_attacks.py:366-496`): several backbones, one hook each,
    cost = sum over models and frames; lr is hard-wired to 0.005 (`:376`)."""

    def __init__(self, model_name_lists, depths, epsilon=16 / 255, steps=60, *, engine=None, graph_builder=None,
                 weight_seed=None):
        super().__init__("ImageGuidedFML2_Adam_MultiModels")
        self.epsilon, self.steps, self.step_size, self.depths = epsilon, steps, 0.005, depths
        self._setup(model_name_lists, [[depths[m]] for m in model_name_lists], engine, graph_builder, weight_seed)


class AENS_I2V_MF(_ImageGuided):
    """Adaptive ENS-I2V (`/root/reference/TPAMI_attack.py:141-320`): several models x several
    layers, per-step re-weighting; returns `(adv, used_time, cost_saved)`.  `coeffs` persists on
    the object across calls as in the reference (`:165,265`).

    Multi-GPU: clips are sharded over ranks; `feat_sum`/`weighted` (2L floats) are all-reduced
    (RCCL) every step so the weights follow the GLOBAL batch, as on one device."""
    _mode = "aens"

    def __init__(self, model_name_lists, depths, step_size, momentum=0, coef_CE=False, epsilon=16 / 255, steps=60,
                 *, engine=None, graph_builder=None, weight_seed=None, process_group=None, distributed=None):
        super().__init__("AENS_I2V_MF")
        self.epsilon, self.steps, self.step_size, self.depths = epsilon, steps, step_size, depths
        self.momentum, self.coef_CE = momentum, coef_CE
        self.coeffs = None
        self.weights = []
        per_model = [list(depths[m]) if isinstance(depths[m], (list, tuple)) else [depths[m]] for m in model_name_lists]
        whole = [isinstance(depths[m], (list, tuple)) for m in model_name_lists]
        self._setup(model_name_lists, per_model, engine, graph_builder, weight_seed, whole)
        self._pg = process_group
        self._dist = distributed

    def _exchange(self, pair):
        """The path's one collective: `pair` = this step's (2, L) floats, summed over the ranks IN PLACE.
        Ordering, with RCCL (`backend="nccl"`): the engine launches on torch's current stream S (`Engine.stream()`); `all_reduce`
        records an event on S behind `aens_reduce_kernel`, makes the process group's own stream wait for it, enqueues the RCCL
        kernel there, and -- a synchronous op -- makes S wait for the collective's completion EVENT (a stream wait, `work.wait()` of
        ProcessGroupNCCL; the host does not block unless TORCH_NCCL_BLOCKING_WAIT asks for it).  So the chain is
        aens_reduce_kernel -> [event] -> RCCL all-reduce (2L floats) -> [event] -> next step's aens_coeffs_kernel,
        entirely on the device; the host runs ahead queueing the next step's compose / forward launches, which only the
        coefficient kernel's consumers (the cosine gradient) wait behind.  With gloo (CPU tensors, the tests) it is a blocking call."""
        import torch.distributed as dist
        on = self._dist if self._dist is not None else (dist.is_available() and dist.is_initialized()
                                                          and dist.get_world_size(self._pg) > 1)
        if on:
            dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=self._pg)

    def forward(self, videos, labels, video_names):
        adv = self._run(videos, video_names)
        cost_saved = np.asarray(self.last_costs, dtype=np.float64)     # np.zeros(steps) of float64, :256
        return adv, self.used_time, cost_saved
