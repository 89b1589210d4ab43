#!/usr/bin/env python
"""bench.py -- adversarial frames/s of the I2V attack on MI355X (BASELINE.json metric).

One "step" = one complete I2V attack over one batch of synthetic clips: flatten + un-normalise,
clean feature pass, S=10 iterations of (compose, ResNet-50 forward to layer3, cosine loss, input
gradient, Adam) and the final compose -- i.e. `ImageGuidedFMDirection_Adam.forward`
(/root/reference/image_attacks.py:294-364).  Workload at N=1: BASELINE.json configs[1], batch = 4
clips of 32 x 224^2 (128 frames), ResNet-50 hook layer3, eps = 16/255, lr = 0.005, inputs
resident in HBM before the timed region.  With --gpus N every rank attacks its own 4 clips (weak
scaling, no data-path collective: frames are independent).

Prints ONE JSON line (rank 0) with `roofline` (conv_igemm, fp32 MFMA) and `cpu_baseline`
(the CPU oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "image-to-video-i2v-attack_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32-input MFMA (= vector) peak
PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured on a float4 copy)
ATTACK_STEPS = 10
CLIPS_PER_GPU = 4
FRAMES, HW = 32, 224
MODEL, DEPTH = "resnet50", 3


def synthetic_clips(b, seed0=1000):
    """uint8 noise clips (exact 0/1 pixels occur), ImageNet-normalised."""
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1)
    clips = []
    for i in range(b):
        gen = torch.Generator().manual_seed(seed0 + i)
        u8 = torch.randint(0, 256, (1, 3, FRAMES, HW, HW), generator=gen, dtype=torch.uint8)
        clips.append((u8.float() / 255 - mean) / std)
    return torch.cat(clips)


def build_id():
    """Identity of the kernel/engine SOURCES the running library was built from (the .git directory does not travel to
    the GPU box, the sources do): sha256 over csrc/ and the ABI header, first 16 hex digits.  A PMC summary is only
    evidence about THIS binary if it carries the same id."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "image-to-video-i2v-attack_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".cpp", ".h"))]
    for f in files + [os.path.join(ROOT, "include", "i2v_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r6_traffic_by_instantiation.json")
PMC_MFMA_BUSY = None      # MFMA-busy fraction of the conv launches from the same PMC summary (same build-id rule as the traffic)


def measured_traffic():
    """HBM bytes per conv_igemm launch from the rocprofv3 PMC passes over this same command (tools/profile.sh ->
    tools/summarise_profile.py: 2 x FETCH_SIZE per the gfx950 correction + WRITE_SIZE, separate --pmc runs).  PMC
    collection cannot run inside the timed process, so the committed summary is reported -- but only when it was
    collected on a build of exactly these sources (`build_id`); otherwise null."""
    try:
        with open(TRAFFIC_JSON) as fh:
            t = json.load(fh)
        if t.get("build_id") != build_id():
            return None, f"profiles/r6_traffic_by_instantiation.json is from build {t.get('build_id')}, running {build_id()}"
        global PMC_MFMA_BUSY
        PMC_MFMA_BUSY = t["conv_igemm"].get("mfma_busy_frac")
        return round(t["conv_igemm"]["hbm_bytes_per_launch"]), None
    except Exception as e:       # noqa: BLE001
        return None, f"no PMC summary ({type(e).__name__})"


def cpu_baseline():
    """The CPU oracle (a port of the reference path; `oracle/restate.py`) on a bounded sample of
    the same workload: ResNet-50 layer3, one 32-frame 224^2 clip (BASELINE configs[0]); as SURVEY.md section 8(d)
    prescribes, 1 untimed warm-up iteration, then the WHOLE attack timed -- clean pass + all 10 iterations --
    (about 20 s of host work): frames / t_attack, nothing extrapolated.
    Thread count: the best of {16, 32, 64} on a short probe (ATen/oneDNN slows down badly when
    over-subscribed: 256 threads on the GPU box's 2x64-core host is >20x slower than 16)."""
    from i2v_amd import graphs, weights
    from oracle import restate, size_parity
    g = graphs.build(MODEL, (HW, HW))
    net = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[DEPTH]])
    vid = synthetic_clips(1)
    x = restate.flatten_frames(vid).contiguous()
    ncpu = os.cpu_count() or 1
    best, cores = None, 1
    for thr in sorted({min(t, ncpu) for t in (16, 32, 64)}):
        torch.set_num_threads(thr)
        t0 = time.time()
        net.forward(x[:8])
        dt = time.time() - t0
        if best is None or dt < best:
            best, cores = dt, thr
    torch.set_num_threads(cores)
    ora = size_parity.oracle_attack(net, vid, steps=ATTACK_STEPS, lr=0.005, warmup=True)
    t_clean, t_iters = ora["t_clean"], ora["t_iters"]
    fps = FRAMES / (t_clean + t_iters)
    base = {"value": round(fps, 3), "unit": "adversarial frames/s", "cores": cores, "kind": "port",
            "sample": f"1 clip x {FRAMES} frames x 224^2, ResNet-50 layer3: 1 warm-up iteration, then the whole attack timed -- "
                      f"clean pass {t_clean:.2f}s + {ATTACK_STEPS} iterations {t_iters:.2f}s on {cores} of {ncpu} host threads"}
    return base, ora


def framework_baseline_child():
    """`bench.py --framework-baseline-child` (started by the bench BEFORE it touches the GPU, idle until told to go): the oracle's
    restatement of the reference loop (`oracle/restate.py`, `oracle/size_parity.oracle_attack`: the same torch ops the reference's
    modules execute, minus autograd's weight gradients) with every tensor on cuda:0 -- ATen -> MIOpen / rocBLAS, i.e. what
    PyTorch-ROCm does with this attack on this very GPU.  One 4-clip attack (configs[1]) after one warm-up iteration.  A BASELINE beside
    `cpu_baseline`: never the target, never imported by the package."""
    if sys.stdin.readline().strip() != "go":
        return 0
    t_start = time.time()
    from i2v_amd import graphs, weights
    from oracle import restate, size_parity
    torch.backends.cudnn.benchmark = False            # MIOpen immediate mode (no exhaustive find): what a default PyTorch run does
    g = graphs.build(MODEL, (HW, HW))
    net = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[DEPTH]], device="cuda:0")
    vid = synthetic_clips(CLIPS_PER_GPU).to("cuda:0")
    t0 = time.time()
    size_parity.oracle_attack(net, vid, steps=1, lr=0.005)        # first use at the timed batch size: MIOpen picks (and caches) its solutions per shape
    torch.cuda.synchronize()
    t_first = time.time() - t0
    ora = size_parity.oracle_attack(net, vid, steps=ATTACK_STEPS, lr=0.005, warmup=True)
    torch.cuda.synchronize()
    t = ora["t_clean"] + ora["t_iters"]
    print(json.dumps({"value": round(CLIPS_PER_GPU * FRAMES / t, 2), "unit": "adversarial frames/s",
                      "kind": "oracle port on PyTorch-ROCm/MIOpen (ATen conv2d / conv2d_input, eager, fp32)",
                      "seconds": {"first_touch_clean_pass_and_one_step": round(t_first, 2), "clean_pass": round(ora["t_clean"], 3),
                                  "iterations": round(ora["t_iters"], 3), "child_total": round(time.time() - t_start, 1)},
                      "costs": [float(f"{c:.7g}") for c in ora["costs"]],
                      "note": f"{CLIPS_PER_GPU} clips x {FRAMES} frames x 224^2, ResNet-50 layer3, {ATTACK_STEPS} steps, tensors resident on cuda:0, one warm-up "
                              "iteration, then the whole attack timed; torch " + torch.__version__ + ", torch.backends.cudnn.benchmark=False (MIOpen "
                              "immediate mode, no find); a baseline, not the target"}), flush=True)
    return 0


def start_framework_baseline():
    """The child of `framework_baseline_child`, started before this process initialises HIP; it waits on its stdin."""
    import subprocess
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--framework-baseline-child"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, text=True, cwd=ROOT)


def collect_framework_baseline(child, budget_s):
    """Tell the child to go and wait at most `budget_s` for its line (a fresh box may have MIOpen compile its kernels first)."""
    import threading
    if child is None:
        return None
    out = {}

    def reader():
        out["line"] = child.stdout.readline()
    try:
        child.stdin.write("go\n"); child.stdin.flush()
    except OSError:
        return {"value": None, "note": "the baseline process had already exited"}
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    th.join(budget_s)
    if th.is_alive() or not out.get("line", "").startswith("{"):
        child.kill()                              # exactly the process started above
        child.wait()
        return {"value": None, "unit": "adversarial frames/s", "kind": "oracle port on PyTorch-ROCm/MIOpen",
                "note": f"no result within the {budget_s:.0f} s budget of a default run (MIOpen building its kernels on a fresh box?); "
                        "the measured figure of a longer run is in profiles/ (tools/run_r6final.sh: --framework-budget 900)"}
    child.wait()
    return json.loads(out["line"])


def parity_check(ora, costs, delta, adv, f64=False):
    """The free-running rung of the parity ladder from the oracle run `cpu_baseline` pays for anyway:
    the device's 10-step attack on the SAME clip (seed 1000) against the oracle's -- cost of every step and mean|delta_10| (the
    well-conditioned quantities), and the perturbed pixels (chaotic under Adam's +-lr steps in ANY pair of fp32
    implementations), held to the fp32 oracle's own distance from the float64 oracle: against the committed float64 run by default
    (round 5), against a live one with `--parity-f64` (`oracle/size_parity.py`, tests/test_gpu_size_parity.py).  The per-step
    atol-1e-4 contract is the teacher-forced rung's."""
    from oracle import size_parity
    st = size_parity.compare(costs, delta.cpu(), adv.cpu(), ora)
    st64 = y32 = None
    # default run: the COMMITTED float64 run of this clip (tests/golden/size_parity_f64_seed1000.npz: costs, mean|delta|, a 1-in-41
    # sample of the perturbed clip) is the yardstick -- the device's and the live fp32 oracle's distance from it over the same sample
    yard = None if f64 else size_parity.load_yardstick(os.path.join(ROOT, "tests", "golden"), seed=1000, steps=ATTACK_STEPS, lr=0.005)
    if yard is not None:
        st64 = size_parity.compare_sampled(costs, delta.cpu(), adv.cpu(), yard)
        y32 = size_parity.compare_sampled(ora["costs"], ora["delta"], ora["adv"], yard)
    if f64:
        from i2v_amd import graphs, weights
        from oracle import restate
        g = graphs.build(MODEL, (HW, HW))
        net64 = restate.OracleNet(g, weights.synthetic_state_dict(g, 0), [g.hooks[DEPTH]], dtype=torch.float64)
        ora64, y32 = size_parity.yardstick(net64, synthetic_clips(1), ora, steps=ATTACK_STEPS)
        st64 = size_parity.compare(costs, delta.cpu(), adv.cpu(), ora64)
    ok, bad = size_parity.within_bounds(st, st64, y32)
    rnd = lambda d: {k: float(f"{v:.4g}") for k, v in d.items()}          # noqa: E731
    out = rnd(st)
    out.update({"against": "fp32 CPU oracle, free-running", "clip": "seed 1000, 1 x 32 x 224^2 (BASELINE.json configs[0])",
                "steps": ATTACK_STEPS, "costs_device": [float(f"{c:.7g}") for c in costs],
                "costs_oracle": [float(f"{c:.7g}") for c in ora["costs"]],
                "bounds": {"max_rel_cost_err": size_parity.COST_RTOL, "mean_abs_delta_ratio": f"1 +- {size_parity.DELTA_MEAN_RTOL}",
                           "pixel statistics": "device's distance from the float64 oracle held to the live fp32 oracle's own distance from it "
                                               f"(x{size_parity.ADV_DIFF_MARGIN}, -{size_parity.PIXEL_FRAC_SLACK}); float64 run: "
                                               + ("live (--parity-f64)" if f64 else "committed fixture, 1-in-41 sample of the clip" if yard is not None
                                                  else "none at hand: pixel statistics unbounded")},
                "within_bounds": bool(ok), "failures": bad})
    if st64 is not None:
        out["device_vs_f64_oracle"], out["fp32_oracle_vs_f64_oracle"] = rnd(st64), rnd(y32)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=None,
                    help=f"clips per GPU and engine call (default {CLIPS_PER_GPU}; --workload ilaf: 8; --workload aens: 8 = BASELINE.json configs[3], batch 64 over 8 GPUs)")
    ap.add_argument("--white_model", default="slowfast_resnet50", help="--workload ilaf: video backbone (graphs.build_video)")
    ap.add_argument("--streams", type=int, default=3, help="--workload ilaf: engine calls in flight on separate HIP streams")
    ap.add_argument("--ilaf_clips", type=int, default=None,
                    help="--workload ilaf: the same as --clips (kept for older command lines; giving both is an error): clips per "
                         "engine call (default 8), attacked as INDEPENDENT one-clip problems (ILAF.forward_independent: "
                         "per-clip loss segments, each clip bit-identical to its one-clip call; the reference runs one clip per call)")
    ap.add_argument("--workload", default="i2v", choices=["i2v", "ens", "aens", "config2", "ilaf"],
                    help="i2v = the headline metric (default); ens / aens = BASELINE configs[2]/[3]-style extras on the "
                         "reference's own model list (resnet101+vgg16+squeezenet1_1+alexnet); aens carries the one data-path "
                         "collective (2L floats all-reduced per step, TPAMI_attack.py:265,293-297) inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-framework-baseline", action="store_true",
                    help="skip `gpu_framework_baseline` (the oracle's restatement of the reference loop run by PyTorch-ROCm / MIOpen on cuda:0, after the timed regions)")
    ap.add_argument("--framework-budget", type=float, default=75.0, help="seconds the default run waits for `gpu_framework_baseline` before reporting null")
    ap.add_argument("--framework-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-split-bf16", action="store_true",
                    help="skip the extra, clearly separated `split_bf16_mode` measurement (the opt-in I2V_MATH=bf16x3 arithmetic; never `value`)")
    ap.add_argument("--parity-f64", action="store_true",
                    help="parity_check: also run the float64 oracle on the same clip (about a minute of host time) and hold the "
                         "device's perturbed pixels to the fp32 oracle's own distance from it")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket backbone launches with HIP events in the timed region")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (measurements); gloo only for functional checks of the rank plumbing")
    ap.add_argument("--share-device", action="store_true",
                    help="FUNCTIONAL CHECK ONLY: every rank uses device 0 (a 1-GPU box); the line is labelled a non-measurement")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="the N=1 `value` of this same workload (e.g. from BENCH_rNN.json): adds efficiency_vs_n1 = value / (N * n1)")
    ap.add_argument("--selftest-hostsim", action="store_true",
                    help="FUNCTIONAL CHECK ONLY (no GPU): the rank plumbing on the CPU host simulation of the kernel backend "
                         "(tests/hostsim, tiny backbone, gloo); the line is labelled a non-measurement")
    args = ap.parse_args(argv)
    if args.clips is not None and args.ilaf_clips is not None and args.clips != args.ilaf_clips:
        ap.error("--clips and --ilaf_clips both given with different values")
    if args.workload == "ilaf":
        args.clips = max(1, args.clips if args.clips is not None else (args.ilaf_clips if args.ilaf_clips is not None else 8))
    elif args.clips is None:      # configs[2]: batch = 8; configs[3]: batch 64 over 8 GPUs = 8 per GPU; configs[1]: batch = 4
        args.clips = 8 if args.workload in ("aens", "config2") else CLIPS_PER_GPU
    return args


def under_launcher():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per device) BEFORE this process
    touches the GPU in any way -- never re-exec a process that has initialised HIP -- wait for them, pass rank 0's
    JSON line through.  (Under `python -m torch.distributed.run ... bench.py --gpus N` the launcher has already made
    the ranks and this function is not used.)  A failed rank fails the run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        bad = []
        while any(p.poll() is None for p in procs) and not bad:
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            time.sleep(0.05)
        if bad:                                  # a dead rank leaves the others waiting in a collective: stop exactly them
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
        bad = bad or [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    if bad:
        print(f"bench.py: rank(s) failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.framework_baseline_child:
        return framework_baseline_child()
    if args.gpus > 1 and not under_launcher():
        return spawn_ranks(args, argv)
    if args.selftest_hostsim:
        return selftest_hostsim(args)
    # the same-GPU framework baseline runs in a child started NOW, before this process initialises HIP (it idles until the timed regions are over)
    alone = args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
    child = start_framework_baseline() if (alone and args.workload == "i2v" and not args.no_cpu_baseline and not args.no_framework_baseline) else None
    try:
        return run_rank(args, child)
    finally:
        if child is not None and child.poll() is None:
            child.kill()
            child.wait()


CPU_PIN = None        # what affinity.pin_rank did for this rank (reported in the JSON line)


def init_ranks(args):
    """(dist or None, rank, local_rank, world, device string, sum of rank ids); one all-reduce of the rank ids
    proves that N ranks really are talking to each other.  FIRST thing, before any GPU call: the rank's CPU cores
    (`i2v_amd.affinity`: NUMA-aware share of the allowed cores; lane / reader / writer threads inherit the mask)."""
    global CPU_PIN
    from i2v_amd import affinity
    CPU_PIN = affinity.pin_rank()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = args.selftest_hostsim
    dev_index = 0 if (args.share_device or cpu) else local_rank
    if not cpu:
        torch.cuda.set_device(dev_index)
    dev = "cpu" if cpu else f"cuda:{dev_index}"
    dist, rank_sum = None, 0
    if world > 1 or under_launcher():
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl" and not cpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)], device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        rank_sum = int(t.item())
        if rank_sum != world * (world - 1) // 2:
            raise RuntimeError(f"rank-id all-reduce gave {rank_sum}, expected {world * (world - 1) // 2}")
    return dist, rank, local_rank, world, dev, rank_sum


def reduce_max(dist, value, dev):
    """max over ranks of a host scalar (the slowest rank's wall time) and the list of every rank's own value."""
    if dist is None:
        return value, [value]
    on = dev if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], device=on, dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)
    vals = [float(e.item()) for e in every]
    return max(vals), vals


def aens_fields(atk, dist, world, dev):
    """What a multi-GPU AENS line carries beside the contract (the path's ONE collective): the layer weights of the
    last step -- functions of the GLOBAL batch, so identical on every rank (gathered and compared bit for bit) and equal to a
    one-device run over the same clips --, the size of the process group as the collective library sees it, and the measured device
    time of one (2, L) in-place all-reduce on the launch stream (HIP events around 20 back-to-back exchanges, after the timed region)."""
    import numpy as np
    w = np.stack(atk.weights).astype(np.float32)                      # (steps, L) of the last call
    f = {"layer_weights_last_step": [float(x) for x in w[-1]], "layers": int(w.shape[1]),
         "exchange": "one in-place all-reduce of 2L floats per attack step (TPAMI_attack.py:265,293-297)"}
    if dist is None:
        f["rccl_ranks"] = 1
        f["note"] = "single process, no process group: the exchange is a no-op"
        return f
    f["rccl_ranks"] = dist.get_world_size()
    f["backend"] = "rccl" if dist.get_backend() == "nccl" else dist.get_backend()
    on = dev if dist.get_backend() == "nccl" else "cpu"
    mine = torch.from_numpy(w).to(on)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    f["weights_identical_on_all_ranks"] = bool(all(torch.equal(e, every[0]) for e in every))
    pair = torch.zeros(2, w.shape[1], device=dev)
    atk._exchange(pair)                                               # warm the path
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        atk._exchange(pair)
    e1.record()
    torch.cuda.synchronize()
    f["allreduce_device_us"] = round(1e3 * e0.elapsed_time(e1) / reps, 2)
    f["allreduce_bytes"] = int(pair.numel() * 4)
    return f


def add_scaling_fields(line, args):
    """`per_gpu_min` / `per_gpu_max` (the spread between ranks: `value` is set by the slowest) and, given `--n1-value`,
    `efficiency_vs_n1` = value / (N x the N=1 value) -- a convenience for reading a scaling run; the driver computes its own."""
    pg = line.get("per_gpu") or []
    if pg:
        line["per_gpu_min"], line["per_gpu_max"] = min(pg), max(pg)
    if args.n1_value:
        line["efficiency_vs_n1"] = round(line["value"] / (line["n_gpus"] * args.n1_value), 4)


def selftest_hostsim(args):
    """FUNCTIONAL CHECK of the multi-rank plumbing without a GPU (CPU container, `tests/test_cli_and_dist_cpu.py`):
    gloo ranks, each attacking its own clips on the HOST SIMULATION of the kernel backend (tests/hostsim -- test
    infrastructure) with a tiny backbone; `--workload aens` all-reduces its 2L floats per step.  Never a
    measurement: the line says so."""
    from tests.hostsim_util import hostsim_engine
    from i2v_amd import attacks, graphs
    if os.environ.get("I2V_BENCH_SELFTEST_FAIL_RANK") == os.environ.get("RANK", "0"):     # test hook: a rank that dies early
        return 3
    dist, rank, _, world, dev, rank_sum = init_ranks(args)
    eng = hostsim_engine()
    b, f, hw = 2, 2, 32
    kw = dict(engine=eng, graph_builder=graphs.build_tiny, weight_seed=0)
    if args.workload == "aens":
        atk = attacks.AENS_I2V_MF(["resnet", "vgg"], depths={"resnet": [2, 3], "vgg": [2, 3]}, step_size=0.005, steps=2, **kw)
    else:
        atk = attacks.ImageGuidedFMDirection_Adam(["resnet"], depth=3, step_size=0.005, steps=2, **kw)
    vids = torch.cat([synthetic_clips(1, seed0=1000 + rank * b + i)[:, :, :f, :hw, :hw] for i in range(b)]).contiguous()
    lab = torch.zeros(b, dtype=torch.long)
    names = [f"clip{rank * b + i}" for i in range(b)]
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        atk(vids, lab, names)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed, per_rank = reduce_max(dist, elapsed, dev)
    extra = {}
    if args.workload == "aens":
        extra["aens_weights_last"] = [round(float(x), 6) for x in atk.weights[-1]]     # identical on every rank: global batch
        extra["aens_weights"] = [[float(x) for x in wrow] for wrow in atk.weights]      # every step of the last call
        if dist is not None:          # ... proved: every rank's whole weight trajectory gathered and compared bit for bit
            import numpy as np
            mine = torch.from_numpy(np.stack(atk.weights).astype(np.float32))
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            extra["aens_weights_identical_on_all_ranks"] = bool(all(torch.equal(e, every[0]) for e in every))
    line = {"metric": "SELFTEST (host simulation, not a measurement)", "value": round(args.steps * b * f * world / elapsed, 3),
            "unit": "adversarial frames/s", "n_gpus": world, "steps": args.steps, "warmup": 0, "data": "synthetic",
            "ranks_proved_by_allreduce": {"sum_of_rank_ids": rank_sum, "expected": world * (world - 1) // 2},
            "per_gpu": [round(args.steps * b * f / t, 3) for t in per_rank], "config": {"workload": f"selftest {args.workload}"},
            "cpu_affinity_rank0": CPU_PIN, **extra}
    add_scaling_fields(line, args)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def split_bf16_measurement(args, eng, attacks, videos, labels, names, b, value, flop_per_frame):
    """The opt-in split-bf16 arithmetic of an EXPERIMENTAL build, AFTER and APART from the headline: a fresh attack object planned in
    that mode, the same K steps un-instrumented, its own parity check against the committed float64 run."""
    atk3 = attacks.ImageGuidedFMDirection_Adam([MODEL], depth=DEPTH, step_size=0.005, steps=ATTACK_STEPS, engine=eng, weight_seed=0)
    atk3.clip_lanes = 1
    atk3(videos, labels, names)
    torch.cuda.synchronize()
    b0 = eng.capi.i2v_backend_stat(b"bf3_launches")
    atk3(videos, labels, names)                # (the launch counter around ONE call)
    torch.cuda.synchronize()
    per_call = int(eng.capi.i2v_backend_stat(b"bf3_launches") - b0)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        atk3(videos, labels, names)
    torch.cuda.synchronize()
    el3 = time.perf_counter() - t1
    atk3.clip_lanes = None                 # product default: two concurrent clip lanes (bit-identical output)
    atk3(videos, labels, names)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        atk3(videos, labels, names)
    torch.cuda.synchronize()
    el3b = time.perf_counter() - t1
    adv3 = atk3(videos[:1].contiguous(), labels[:1], names[:1])
    torch.cuda.synchronize()
    from oracle import size_parity
    yard = size_parity.load_yardstick(os.path.join(ROOT, "tests", "golden"), seed=1000, steps=ATTACK_STEPS, lr=0.005)
    st3 = size_parity.compare_sampled(atk3.last_costs, atk3._delta.cpu(), adv3.cpu(), yard) if yard is not None and b >= 1 else None
    v3 = args.steps * b * FRAMES / el3
    return {"math": "I2V_MATH=bf16x3 (opt-in, experimental build): conv launches on three-term bf16 operands, 6 bf16 MFMAs per 16 K rows, fp32 accumulation; "
                    "product terms kept down to 2^-26 |w||x|",
            "value": round(v3, 2), "unit": "adversarial frames/s", "ms_per_step": round(1e3 * el3 / args.steps, 3),
            "speedup_vs_value": round(v3 / value, 3), "end_to_end_tflops_equivalent": round(v3 * flop_per_frame / 1e12, 2),
            "value_product_default_lanes": round(args.steps * b * FRAMES / el3b, 2),
            "bf3_launches_per_step": per_call,
            "device_vs_f64_oracle": {k: float(f"{v:.4g}") for k, v in st3.items()} if st3 else None,
            "note": "NOT the headline: `value` is the exact-fp32 path.  Same clips, same K steps, one clip lane, no per-launch events; "
                    "parity of this mode: tests/test_gpu_split_bf16.py, DESIGN_HISTORY.md R5-12"}


def run_rank(args, framework_child=None):
    dist, rank, local_rank, world, dev, rank_sum = init_ranks(args)

    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB) and rank == 0:      # normally built by __graft_entry__.build(); a no-op otherwise
        ge.build()
    if dist is not None:
        dist.barrier()
    from i2v_amd import attacks, graphs
    eng = attacks.get_engine(dev)
    names4 = ["resnet", "vgg", "squeezenet", "alexnet"]
    if args.workload == "i2v":
        atk = attacks.ImageGuidedFMDirection_Adam([MODEL], depth=DEPTH, step_size=0.005, steps=ATTACK_STEPS, engine=eng, weight_seed=0)
    elif args.workload == "ens":
        atk = attacks.ImageGuidedFML2_Adam_MultiModels(names4, depths={"resnet": 2, "vgg": 3, "squeezenet": 2, "alexnet": 3},
                                                       steps=ATTACK_STEPS, engine=eng, weight_seed=0)
    elif args.workload == "config2":      # BASELINE.json configs[2]: ResNet-50 + VGG-16 + DenseNet-121
        names3 = ["resnet50", "vgg", "densenet121"]
        atk = attacks.ImageGuidedFML2_Adam_MultiModels(names3, depths={"resnet50": 3, "vgg": 3, "densenet121": 3},
                                                       steps=ATTACK_STEPS, engine=eng, weight_seed=0)
    elif args.workload == "ilaf":         # BASELINE.json configs[4]: ILAF fine-tuning, 60 steps (image_attacks.py:502), 1 clip per call
        from i2v_amd import sign_attacks, video
        atk = sign_attacks.ILAF(video.VideoModel(args.white_model, (FRAMES, HW, HW), weight_seed=0), args.white_model, engine=eng)
    else:
        atk = attacks.AENS_I2V_MF(names4, depths={n: [2, 3] for n in names4}, step_size=0.005, steps=ATTACK_STEPS, engine=eng, weight_seed=0)
    b = args.clips
    if args.workload == "ilaf":
        atk.independent_clips = True
    if args.workload == "aens" and world > 1:
        out_parallelism = f"clips sharded over {world} GPUs, one all-reduce of 2L floats per attack step (RCCL)"
    videos = synthetic_clips(b, seed0=1000 + rank * b).to(dev)         # resident in HBM before timing
    labels = torch.zeros(b, dtype=torch.long)
    names = [f"clip{rank * b + i}" for i in range(b)]
    if args.workload == "ilaf":           # (existing adversarial clip, clean clip): the clean clip + a +-10/255 perturbation
        gen = torch.Generator().manual_seed(77 + rank)
        std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1)
        noise = (torch.randint(-10, 11, videos.shape, generator=gen).float() / 255 / std).to(dev)
        ori = videos
        _ilaf = atk
        videos = (ori + noise).contiguous()
        # `--streams` clip streams: each worker thread owns an ILAF object (its own planned net) and a HIP stream and
        # works on its own clip pair; a step = one ILAF call on every stream (sign_attacks.run_concurrent).
        import threading
        lanes = [(_ilaf, videos, ori, torch.cuda.Stream(device=dev))]
        for k in range(1, max(1, args.streams)):
            o2 = synthetic_clips(b, seed0=5000 + rank * 64 + k * b).to(dev)
            n2 = (torch.randint(-10, 11, o2.shape, generator=gen).float() / 255 / std).to(dev)
            a2 = sign_attacks.ILAF(video.VideoModel(args.white_model, (FRAMES, HW, HW), weight_seed=0), args.white_model, engine=eng)
            a2.independent_clips = True
            lanes.append((a2, (o2 + n2).contiguous(), o2, torch.cuda.Stream(device=dev)))

        for a_, v_, _, _ in lanes:            # plan every stream's net now, one after the other, on a quiet device (plan-time autotuning
            a_.plan_for(v_)                   # under the other streams' traffic picked different kernels per stream)
        torch.cuda.synchronize()

        def lane_calls(lane, reps):
            a, v, o, st = lane
            torch.cuda.set_device(dev)                    # per-thread state (a new thread starts on device 0)
            with torch.cuda.stream(st):
                for _ in range(reps):
                    a(v, o, labels, names)
                st.synchronize()

        def atk(v, l, n, reps=1):
            ths = [threading.Thread(target=lane_calls, args=(ln, reps)) for ln in lanes]
            for t in ths:
                t.start()
            for t in ths:
                t.join()

    timing = not args.no_kernel_timing
    # The attack classes cut an I2V / ENS batch into two concurrently executed clip lanes by default (attacks.py,
    # +4 % frames/s).  Under concurrency a launch's event-measured duration is no longer its own, so the timed,
    # per-launch-instrumented region runs ONE lane (the roofline then means what it says and agrees with rocprofv3);
    # the product-default configuration is timed afterwards, un-instrumented, and reported as `product_default`.
    lanes_capable = args.workload in ("i2v", "ens", "config2")
    if lanes_capable and timing:
        atk.clip_lanes = 1
    for _ in range(max(args.warmup, 0)):
        atk(videos, labels, names)
    torch.cuda.synchronize()
    # ILAF on one clip is ~3000 launches of ~30 us per call: bracketing each with an event pair costs ~20 %, so its
    # per-kernel times come from ONE extra call after the timed region instead of from inside it
    timing_outside = timing and args.workload == "ilaf"
    # Headline workloads: the timed region carries one HIP event pair per SEGMENT (a run of consecutive launches of one kind on the
    # launch stream: ~70 records per attack instead of ~1000, which cost 1.7 % of the figure they were measuring) -- conv_igemm's
    # device time, flops, launches and bytes per pass come from there; the per-LAUNCH split into MFMA-bound and HBM-bound launches
    # (`by_bound`) comes from one extra attack after the timed region, instrumented launch by launch.
    seg_mode = timing and not timing_outside and args.workload in ("i2v", "ens", "config2", "aens")
    if timing and not timing_outside:
        eng.timing_enable("segments" if seg_mode else True)
        atk(videos, labels, names)          # pre-create the event pool outside the timed region
        eng.timing_collect()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    if args.workload == "ilaf":
        atk(videos, labels, names, reps=args.steps)       # every stream runs `steps` calls back to back
    else:
        for _ in range(args.steps):
            atk(videos, labels, names)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    own_elapsed = elapsed
    elapsed, per_rank_s = reduce_max(dist, elapsed, dev)
    if timing_outside:                      # one stream, instrumented
        eng.timing_enable(True)
        _ilaf(videos, ori, labels, names); eng.timing_collect()      # event pool
        _ilaf(videos, ori, labels, names)
    kt = eng.timing_collect() if timing else None
    kt_launch = None
    if timing and seg_mode:                 # one more attack, per launch: the by-bound split (after the timed region)
        eng.timing_enable(True)
        atk(videos, labels, names); eng.timing_collect()      # event pool
        atk(videos, labels, names)
        kt_launch = eng.timing_collect()
    if timing:
        eng.timing_enable(False)
    product_default = None
    if lanes_capable and timing:
        atk.clip_lanes = None                 # product default: $I2V_CLIP_LANES or 2
        atk(videos, labels, names)            # plans the lanes' nets, warms up
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            atk(videos, labels, names)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        barrier()
        el2, _ = reduce_max(dist, el2, dev)
        product_default = {"clip_lanes": atk._lane_count(b, FRAMES), "value": round(args.steps * b * FRAMES * world / el2, 2),
                           "unit": "adversarial frames/s", "ms_per_step": round(1e3 * el2 / args.steps, 3),
                           "note": "same K steps, no per-launch events, batch cut into concurrent clip lanes (bit-identical output)"}

    # the reference CLI's default `--batch_size 1` (image_main.py:22; BASELINE.json configs[0] shape): one clip per call, product
    # default lanes (a single clip is cut along its frames), after the headline regions -- an extra field, never `value`
    single_clip = None
    if args.workload == "i2v" and b > 1 and timing:       # (--no-kernel-timing = the rocprofv3 passes: only the headline launches)
        atk.clip_lanes = None
        v1 = videos[:1].contiguous()
        atk(v1, labels[:1], names[:1])           # plans the 32-frame nets
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = max(3, args.steps)
        for _ in range(reps):
            atk(v1, labels[:1], names[:1])
        torch.cuda.synchronize()
        el1 = time.perf_counter() - t1
        single_clip = {"value": round(reps * FRAMES / el1, 2), "unit": "adversarial frames/s", "ms_per_clip": round(1e3 * el1 / reps, 3),
                       "clip_lanes": atk._lane_count(1, FRAMES), "note": "one 32-frame clip per call (the reference CLI's default batch), this rank only"}
    n_lanes = len(lanes) if args.workload == "ilaf" else 1
    frames_total = args.steps * b * FRAMES * world * n_lanes
    value = frames_total / elapsed
    g = graphs.build(MODEL, (HW, HW)).truncated([graphs.build(MODEL, (HW, HW)).hooks[DEPTH]])
    mac = g.macs_per_frame()
    flop_per_frame = (4 * ATTACK_STEPS + 2) * mac
    out = {
        "metric": "adversarial frames/sec (10-step I2V, ResNet-50 layer3, 32x224^2 clips)",
        "value": round(value, 2), "unit": "adversarial frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"I2V ResNet-50 layer3, batch={b} synthetic 32-frame 224^2 clips per GPU, "
                               f"{ATTACK_STEPS} steps eps=16/255 lr=0.005 (BASELINE.json configs[1])",
                   "frames_per_gpu": b * FRAMES, "attack_steps": ATTACK_STEPS, "weights": "seeded synthetic (seed 0)",
                   "math": ("fp32-input MFMA (exact fp32: the default)" if os.environ.get("I2V_MATH", "") != "bf16x3" else
                            "NOT THE DEFAULT: the caller set I2V_MATH=bf16x3 (split-bf16 operands, fp32 accumulation) for the whole run"),
                   "parallelism": f"clips sharded over {world} GPU(s), no collective"},
        "end_to_end_tflops_per_gpu": round(value / world * flop_per_frame / 1e12, 2),
        "algorithmic_gflop_per_frame": round(flop_per_frame / 1e9, 2),
        # every rank's own frames/s over ITS wall time of the same K steps (`value` divides all frames by the slowest)
        "per_gpu": [round(args.steps * b * FRAMES * n_lanes / t, 2) for t in per_rank_s],
    }
    add_scaling_fields(out, args)
    if CPU_PIN is not None:
        out["cpu_affinity_rank0"] = CPU_PIN
    if dist is not None:
        out["ranks_proved_by_allreduce"] = {"sum_of_rank_ids": rank_sum, "expected": world * (world - 1) // 2,
                                            "backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend()}
    if args.share_device:
        out["metric"] = "FUNCTIONAL CHECK (all ranks share device 0, not a measurement): " + out["metric"]
    if os.environ.get("I2V_MATH", "") == "bf16x3":
        out["metric"] = "OPT-IN MATH MODE I2V_MATH=bf16x3 (not the headline configuration): " + out["metric"]
    if kt is not None and kt["conv_igemm_fwd"]["launches"]:
        # every instantiation of conv_igemm: backbone fwd + dgrad, and the class-packed image gradient
        parts = ("conv_igemm_fwd", "conv_igemm_dgrad", "conv_igemm_imggrad")
        c = {k: sum(kt[p][k] for p in parts) for k in kt["conv_igemm_fwd"]}
        tf = lambda d: round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2) if d["ms"] else None
        ach = c["flops"] / (c["ms"] * 1e-3) / 1e12
        traffic, why_not = measured_traffic() if args.workload == "i2v" and b == CLIPS_PER_GPU else (None, "not the headline workload")
        # the launches whose algorithmic flops/byte is below the machine balance (157.3 TFLOP/s / 8 TB/s = 19.7): the
        # low-K "expand" convolutions and their input gradients -- those are bounded by the HBM roofline, the rest by MFMA
        if kt_launch is not None:           # by-bound split from the per-launch pass (one attack), totals above from the timed region's segments
            cl = {k: sum(kt_launch[p][k] for p in parts) for k in kt_launch["conv_igemm_fwd"]}
            for k in ("lowi_ms", "lowi_bytes", "lowi_launches", "lowi_flops"):
                c[k] = cl[k]
            hi = {k: cl[k] - cl["lowi_" + k] for k in ("ms", "flops", "launches", "bytes")}
        else:
            hi = {k: c[k] - c["lowi_" + k] for k in ("ms", "flops", "launches", "bytes")}
        out["roofline"] = {"kernel": "conv_igemm (fp32 MFMA implicit GEMM, fwd + dgrad)", "bound": "mfma",
                           "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                           "traffic": traffic,
                           "traffic_unit": "B per launch (rocprofv3 PMC passes of this build: 2*FETCH_SIZE + WRITE_SIZE, profiles/r6_traffic_by_instantiation.json)",
                           "launches": int(c["launches"]), "avg_launch_us": round(1e3 * c["ms"] / c["launches"], 2),
                           "avg_gflop_per_launch": round(c["flops"] / c["launches"] / 1e9, 3),
                           "algorithmic_bytes_per_launch": round(c["bytes"] / c["launches"]),
                           "achieved_by_pass": {"forward": tf(kt["conv_igemm_fwd"]), "input_grad": tf(kt["conv_igemm_dgrad"]),
                                                "image_grad": tf(kt["conv_igemm_imggrad"])},
                           "by_bound": {
                               "mfma_bound_launches": {"launches": int(hi["launches"]), "device_ms": round(hi["ms"], 2),
                                                       "achieved_tflops": round(hi["flops"] / (hi["ms"] * 1e-3) / 1e12, 2) if hi["ms"] else None,
                                                       "frac_of_mfma_peak": round(hi["flops"] / (hi["ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if hi["ms"] else None},
                               "hbm_bound_launches": {"launches": int(c["lowi_launches"]), "device_ms": round(c["lowi_ms"], 2),
                                                      "rule": "algorithmic flops/byte < 19.7 (= 157.3 TFLOP/s / 8 TB/s)",
                                                      "achieved_gbps": round(c["lowi_bytes"] / (c["lowi_ms"] * 1e-3) / 1e9, 1) if c["lowi_ms"] else None,
                                                      "peak_gbps": PEAK_HBM_GBPS,
                                                      "frac_of_hbm_peak": round(c["lowi_bytes"] / (c["lowi_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if c["lowi_ms"] else None,
                                                      "achieved_tflops": round(c["lowi_flops"] / (c["lowi_ms"] * 1e-3) / 1e12, 2) if c["lowi_ms"] else None}},
                           "device_ms_by_kernel": {k: round(v["ms"], 2) for k, v in kt.items()},
                           "wall_ms_timed_region": round(1e3 * elapsed, 2),
                           "event_granularity": ("one HIP event pair per segment (consecutive launches of one kind) over the timed region; `by_bound` from one "
                                                 "extra attack instrumented per launch") if kt_launch is not None else "one HIP event pair per launch",
                           "build_id": build_id()}
        if why_not:
            out["roofline"]["traffic_note"] = why_not
        elif PMC_MFMA_BUSY is not None:      # same PMC summary, same build: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x time x 2.4 GHz), time-weighted
            out["roofline"]["mfma_busy_frac_pmc"] = PMC_MFMA_BUSY
        if timing_outside:
            out["roofline"]["note"] = "per-kernel times from one extra instrumented call after the timed region"
    if args.workload == "ilaf":
        vg = _ilaf.model.graph_for((FRAMES, HW, HW))
        vmac = vg.truncated(_ilaf.model.hook_tensors(vg)).macs_per_frame()          # per clip
        out["metric"] = f"adversarial frames/sec ({_ilaf.steps}-step ILAF, {args.white_model} white-box, 32x224^2 clips)"
        out["config"] = {"workload": f"ILAF fine-tune (BASELINE.json configs[4]): {b} independent one-clip problem(s) per engine call "
                                     f"(per-clip loss segments), {n_lanes} call(s) in flight per GPU on separate HIP streams, "
                                     f"{_ilaf.steps} sign steps of 0.005, eps=16/255, hooks of {args.white_model} (synthetic weights)",
                         "frames_per_gpu": b * FRAMES * n_lanes, "attack_steps": _ilaf.steps, "streams": n_lanes,
                         "parallelism": f"{n_lanes} clip stream(s) per GPU x {world} GPU(s), replicas only"}
        out["algorithmic_gflop_per_frame"] = round((4 * _ilaf.steps + 4) * vmac / FRAMES / 1e9, 2)
        out["end_to_end_tflops_per_gpu"] = round(value / world * (4 * _ilaf.steps + 4) * vmac / FRAMES / 1e12, 2)
    elif args.workload != "i2v":
        if args.workload == "aens" and world > 1:
            out["config"]["parallelism"] = out_parallelism
        models = "resnet50+vgg16+densenet121" if args.workload == "config2" else "resnet101+vgg16+squeezenet1_1+alexnet"
        out["metric"] = f"adversarial frames/sec (10-step {args.workload.upper()}-I2V, {models}, 32x224^2 clips)"
        out["config"]["workload"] = f"{args.workload} ensemble ({models}), batch={b} clips per GPU"
        for k in ("end_to_end_tflops_per_gpu", "algorithmic_gflop_per_frame"):
            out.pop(k, None)
    # which BASELINE.json entry this line measures, verbatim (and what, if anything, differs from it as run)
    idx = {"i2v": 1, "config2": 2, "aens": 3, "ilaf": 4}.get(args.workload)
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            entries = json.load(fh)["configs"]
    except Exception:       # noqa: BLE001
        entries = None
    if idx is not None and entries:
        differs = {"i2v": None if b == CLIPS_PER_GPU else f"batch={b}",
                   "config2": None if b == 8 else f"batch={b}",
                   "aens": f"this GPU's share: {b} clips per GPU x {world} GPU(s)" + ("" if world == 8 and b == 8 else " (the entry: 64 over 8)")
                           + "; model list = the reference CLI's (resnet101+vgg16+squeezenet1_1+alexnet, depths [2,3])",
                   "ilaf": f"{args.white_model} feature guide, {world} GPU(s), replicas only" + ("" if "slowfast" in args.white_model else " (the entry names SlowFast-R50)")}[args.workload]
        out["config"]["baseline_entry"] = f"configs[{idx}]: {entries[idx]}"
        out["config"]["differs_from_entry"] = ((differs + "; ") if differs else "") + "synthetic clips and seeded synthetic weights (no dataset / checkpoints offline)"
    elif args.workload == "ens":
        out["config"]["baseline_entry"] = "none: ENS-I2V on the reference CLI's own model list (an extra beside configs[2])"
    if args.workload == "aens":
        out["aens"] = aens_fields(atk, dist, world, dev)
    if args.workload == "i2v" and getattr(atk, "_nets", None) and eng.capi.i2v_backend_stat(b"experimental") == 1:       # what the autotuner did with the fusable 3x3 -> pointwise pairs (headline plan)
        fi = atk._nets[0].fusion_info()
        out["fused_pairs"] = {"eligible_fwd": fi[0], "eligible_bwd": fi[1], "fused_fwd": fi[2], "fused_bwd": fi[3],
                              "note": "pairs run as one conv_fused_kernel launch at the planned batch size (autotuned per pair; experimental build, DESIGN.md section 12)"}
    out["plan_ms"] = {"total": round(eng.plan_ms, 1), "plans": eng.plans,
                      "note": "host wall time of building the planned backbones (weight packing, upload, launch lists, plan-time "
                              "autotuning of every convolution launch); paid once per (backbone, resolution, max batch), never inside a timed region"}
    if product_default is not None:
        out["product_default"] = product_default
    if single_clip is not None:
        out["single_clip"] = single_clip
    # The opt-in split-bf16 arithmetic (I2V_MATH=bf16x3: six bf16 MFMAs per 16 K rows on three-term operands, fp32 accumulation), measured
    # AFTER and APART from the headline: a fresh attack object planned in that mode, the same K steps un-instrumented, and its own parity
    # check against the committed float64 run.  `value` above is the default exact-fp32 path and stays so.
    split = None
    experimental = eng.capi.i2v_backend_stat(b"experimental") == 1
    if args.workload == "i2v" and not args.no_split_bf16 and timing and world == 1 and os.environ.get("I2V_MATH", "") != "bf16x3":
        if not experimental:
            split = "not built: the split-bf16 K loop is an EXPERIMENTAL kernel (csrc/i2v_conv_exp.hip, -DI2V_EXPERIMENTAL), not part of the product library"
        else:
            os.environ["I2V_MATH"] = "bf16x3"
            try:
                split = split_bf16_measurement(args, eng, attacks, videos, labels, names, b, value, flop_per_frame)
            except Exception as e:       # noqa: BLE001 -- an optional side measurement must never cost the headline line
                split = {"error": f"{type(e).__name__}: {e}"}
            finally:
                os.environ.pop("I2V_MATH", None)
    if split is not None:
        out["split_bf16_mode"] = split
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.workload == "i2v":
            # device run of the clip the oracle is about to attack (seed 1000 = this rank's first clip), outside every timed region
            atk.clip_lanes = None
            adv1 = atk(videos[:1].contiguous(), labels[:1], names[:1])
            torch.cuda.synchronize()
            dev_costs, dev_delta = atk.last_costs.copy(), atk._delta.clone()
            out["gpu_framework_baseline"] = collect_framework_baseline(framework_child, args.framework_budget)
            if out["gpu_framework_baseline"] is None:
                out.pop("gpu_framework_baseline")
            out["cpu_baseline"], ora = cpu_baseline()
            out["parity_check"] = parity_check(ora, dev_costs, dev_delta, adv1, f64=args.parity_f64)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
