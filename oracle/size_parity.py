"""TEST INFRASTRUCTURE -- the free-running rung of the parity ladder at BASELINE size.

`oracle_attack` runs the CPU restatement's whole I2V attack (`restate.run_attack` arithmetic, spelled out here so that
the clean pass and the iterations can be timed separately: `/root/reference/image_attacks.py:294-364`) on one synthetic
clip of BASELINE.json configs[0] -- ResNet-50 layer3, 32 x 224^2, 10 steps -- and keeps what the ladder compares: the cost
of every step, the final Adam variable and the perturbed clip.  `compare` turns a device run of the same clip into the
four statistics of the rung.  Used by `tests/test_gpu_size_parity.py` and by `bench.py`'s `cpu_baseline` leg (which
times the same run), never by the product package.

Why statistics and not atol 1e-4: the free-running fp32 loop is chaotic (the reference against ITSELF with
another summation order differs by > 1e-4 on 40 % of the pixels after 10 Adam steps, the cost trajectory agrees to ~1e-5);
the per-step atol contract is held by the teacher-forced rungs (tests/golden/tf_*.npz, test_gpu_parity.py).
"""
import time

import numpy as np
import torch

from . import restate

MEAN = torch.tensor(restate.MEAN).view(1, 3, 1, 1, 1)
STD = torch.tensor(restate.STD).view(1, 3, 1, 1, 1)

# The rung's bounds.  Costs and mean|delta| are well-conditioned (cost rel-err <= 1e-4...2e-4 per step,
# mean|delta| within 1 %).  The perturbed PIXELS are not: Adam's first steps move every pixel by +-lr whatever the size of its
# gradient, so each sign flip of a gradient component that is zero to rounding puts that pixel 2*lr away, in ANY pair of
# correct fp32 implementations.  The yardstick for the pixel statistics is therefore the reference arithmetic's own distance
# from exact arithmetic -- the fp32 oracle against the SAME oracle in float64 (`yardstick`) -- and the device run is held to
# that distance times the margins below.
COST_RTOL = 2e-4            # every step's cost, against the fp32 oracle and against the float64 oracle
DELTA_MEAN_RTOL = 0.01      # mean|delta_S| of the two runs
ADV_DIFF_MARGIN = 1.25      # mean|adv - adv_f64| of the device <= margin x mean|adv_f32oracle - adv_f64|
PIXEL_FRAC_SLACK = 0.02     # share of pixel-channels within 2*lr of the f64 run: device >= fp32 oracle's share - slack


def synthetic_clip(seed, frames=32, hw=224):
    """uint8 noise (exact 0 / 255 pixels occur), ImageNet-normalised, (1,3,f,h,w)."""
    gen = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (1, 3, frames, hw, hw), generator=gen, dtype=torch.uint8)
    return (u8.float() / 255 - MEAN) / STD


def oracle_attack(net, vid, steps=10, lr=0.005, eps=16 / 255, warmup=False):
    """One whole I2V attack of the oracle on `vid` (1,3,f,h,w).  Returns costs (steps,) float32, delta (N,3,h,w), adv
    (1,3,f,h,w) and the wall time of the clean pass / the iterations.  `warmup`: one untimed iteration first."""
    b, _, f, _, _ = vid.shape
    x = restate.flatten_frames(vid).contiguous()
    u = restate.unnormalise(x)                                          # image_attacks.py:308
    t0 = time.time()
    init = [t.clone() for t in net.forward(x)]                          # :318-323 (raw x)
    t_clean = time.time() - t0

    def iteration(delta, opt):
        xn, mask = restate.compose(u, delta, eps)                       # :331-332
        feats = net.forward(xn)                                         # :334
        cs, gr = restate.cosine_fwd_bwd(feats[0], init[0])              # :341-347
        opt.step(delta, restate.compose_backward(net.backward([gr]), mask))   # :351-353
        return float(cs.sum())

    if warmup:
        d0 = torch.full_like(x, 0.01 / 255)
        iteration(d0, restate.AdamState(d0, lr))
    delta = torch.full_like(x, 0.01 / 255)                              # :304
    opt = restate.AdamState(delta, lr)
    costs = np.zeros(steps, np.float32)
    t0 = time.time()
    for i in range(steps):
        costs[i] = iteration(delta, opt)
    t_iters = time.time() - t0
    xn, _ = restate.compose(u, delta, eps)                              # :360-361
    return {"costs": costs, "delta": delta, "adv": restate.unflatten_frames(xn, b, f).contiguous(), "t_clean": t_clean,
            "t_iters": t_iters, "lr": lr}


def compare(costs, delta, adv, ora):
    """The rung's statistics for a device run (`costs`, final `delta` (N,3,h,w), `adv` (b,3,f,h,w), CPU tensors) against the
    oracle's `ora` of the same clip(s)."""
    costs = np.asarray(costs, np.float64)
    oc = np.asarray(ora["costs"], np.float64)
    rel = np.abs(costs - oc) / np.abs(oc)
    # (an oracle run that was saved by a worker process carries mean|delta| as a number instead of the tensor: oracle/fooling_worker.py)
    d_dev = float(delta) if isinstance(delta, float) else float(delta.float().abs().mean())
    d_ora = float(ora["mean_abs_delta"]) if "mean_abs_delta" in ora else float(ora["delta"].float().abs().mean())
    diff = (adv.float() - ora["adv"].float()).abs()
    un = diff * STD                                                     # back to [0,1] pixel units
    return {"max_rel_cost_err": float(rel.max()), "rel_cost_err_last_step": float(rel[-1]),
            "mean_abs_delta_ratio": d_dev / d_ora, "mean_abs_adv_diff": float(diff.mean()),
            "frac_pixels_within_2lr": float((un <= 2 * ora["lr"]).float().mean()),
            "max_abs_adv_diff_pixel_units": float(un.max())}


def yardstick(net64, vid, ora32, steps=10, lr=0.005, eps=16 / 255):
    """The fp32 oracle's own distance from exact arithmetic on this clip: the same attack by the float64 oracle (same fp32
    input values, same weights cast up) and `compare(fp32 oracle, f64 oracle)`.  Returns (ora64, stats32)."""
    ora64 = oracle_attack(net64, vid.double(), steps=steps, lr=lr, eps=eps)
    ora64["delta"], ora64["adv"] = ora64["delta"].float(), ora64["adv"].float()
    return ora64, compare(ora32["costs"], ora32["delta"], ora32["adv"], ora64)


YARDSTICK_NPZ = "size_parity_f64_seed1000.npz"      # tests/golden/, written by oracle/make_size_yardstick.py


def load_yardstick(golden_dir, seed=1000, steps=10, lr=0.005):
    """The committed float64 run of configs[0] on clip `seed` (`oracle/make_size_yardstick.py`): its costs, mean|delta_S| and every
    STRIDE-th element of its perturbed clip -- or None when the request is not the configuration the file was made for."""
    import os
    path = os.path.join(golden_dir, YARDSTICK_NPZ)
    if not os.path.isfile(path):
        return None
    z = np.load(path)
    if int(z["seed"]) != seed or int(z["steps"]) != steps or float(z["lr"]) != lr:
        return None
    return {k: z[k] for k in z.files}


def compare_sampled(costs, delta, adv, yard, lr=0.005):
    """`compare` against the committed float64 run: the pixel statistics over its sample of the clip (every `stride`-th element of the
    flattened (1,3,f,h,w) tensor: 117 k of 4.8 M values)."""
    costs = np.asarray(costs, np.float64)
    rel = np.abs(costs - yard["costs"]) / np.abs(yard["costs"])
    flat = adv.float().reshape(-1)
    assert flat.numel() == int(yard["numel"]), (flat.numel(), int(yard["numel"]))
    stride = int(yard["stride"])
    idx = torch.arange(0, flat.numel(), stride)
    diff = (flat[idx] - torch.from_numpy(yard["adv_sample"])).abs()
    chan = idx // (flat.numel() // 3)                                   # (1, 3, f, h, w): channel of every sampled element
    un = diff * torch.tensor(restate.STD)[chan]
    return {"max_rel_cost_err": float(rel.max()), "rel_cost_err_last_step": float(rel[-1]),
            "mean_abs_delta_ratio": float(delta.float().abs().mean()) / float(yard["mean_abs_delta"]),
            "mean_abs_adv_diff": float(diff.mean()), "frac_pixels_within_2lr": float((un <= 2 * lr).float().mean()),
            "max_abs_adv_diff_pixel_units": float(un.max())}


def effective_cpus():
    """CPUs this process may really use at once: the affinity mask capped by the cgroup's CFS quota (the GPU box shows 256 CPUs and
    grants 16: `cpu.max` = `1600000 100000`)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def start_oracle_workers(rows, out_dir, workers=None, threads=1, steps=10, lr=0.005, extra=(), f64_rows=(), f64_threads=4, pin=False):
    """Start CPU child processes of `oracle.fooling_worker` over `rows` (dealt round-robin, so the early rows of every worker finish
    first) and, for `f64_rows`, one more with `f64_threads` threads that runs the float64 oracle on those rows; returns the Popen
    objects.  Children of the caller -- never an exec of the caller itself, which may hold the GPU.
    Default: ONE THREAD per worker and as many workers as the host really grants CPUs (`effective_cpus`, minus the float64 worker's).
    Measured on the GPU box (tools/oracle_scaling_probe.py; 256 CPUs visible, CFS quota 16): seconds per clip with 1 x 32 threads
    16.0, 2 x 16 11.0, 4 x 8 6.9, 4 x 4 4.8, 8 x 2 4.2, 16 x 1 3.3 -- ATen's convolutions scale poorly over threads, processes do not
    care; anything beyond the quota is throttled (8 x 28 threads beside a 32-thread caller: 43 s per clip)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ncpu = effective_cpus()
    if workers is None:
        workers = max(1, (ncpu - (f64_threads if f64_rows else 0)) // max(1, threads))
    workers = max(1, min(workers, len(rows)))

    def launch(mine, thr, slot, f64):
        cmd = [sys.executable, "-m", "oracle.fooling_worker", "--rows", ",".join(str(r) for r in mine), "--threads", str(thr),
               "--out", out_dir, "--steps", str(steps), "--lr", str(lr)] + (["--slot", str(slot)] if pin else []) + list(extra)
        if f64:
            cmd += ["--f64_rows", ",".join(str(r) for r in f64)]
        return subprocess.Popen(cmd, cwd=root, env=dict(os.environ, OMP_NUM_THREADS=str(thr), MKL_NUM_THREADS=str(thr)))

    procs = [launch(rows[w::workers], threads, w, ()) for w in range(workers)]
    if f64_rows:
        procs.append(launch([], max(1, min(f64_threads, ncpu)), workers, list(f64_rows)))
    return procs


def wait_oracle_row(out_dir, row, procs, timeout=1800.0, lr=0.005, tag="oracle"):
    """Block until row `row` of a `start_oracle_workers` run is complete; returns its `ora` dict (costs, mean_abs_delta, adv (1,3,f,h,w), lr).
    Raises if a worker died without producing it.  `tag="oracle64"`: the float64 run of an `f64_rows` row."""
    import os
    done = os.path.join(out_dir, f"{row}-{tag}.npz")
    t0 = time.time()
    while not os.path.exists(done):
        if all(p.poll() is not None for p in procs) and not os.path.exists(done):
            raise RuntimeError(f"oracle workers exited ({[p.returncode for p in procs]}) without row {row}")
        if time.time() - t0 > timeout:
            raise TimeoutError(f"oracle row {row} not ready after {timeout} s")
        time.sleep(0.2)
    z = np.load(done)
    adv = torch.from_numpy(np.load(os.path.join(out_dir, f"{row}-{tag}-adv.npy")))[None]
    return {"costs": z["costs"], "mean_abs_delta": float(z["mean_abs_delta"]), "adv": adv, "lr": lr, "seconds": float(z["seconds"])}


def within_bounds(dev_vs_32, dev_vs_64=None, ora32_vs_64=None):
    """The rung: costs and mean|delta| against the fp32 oracle (and, when the float64 run is at hand, against it too); the pixel
    statistics against the float64 run, held to the fp32 oracle's own distance from it.  Returns (ok, list of failures)."""
    bad = []
    for tag, st in (("fp32 oracle", dev_vs_32), ("f64 oracle", dev_vs_64)):
        if st is None:
            continue
        if not st["max_rel_cost_err"] <= COST_RTOL:
            bad.append(f"cost trajectory vs {tag}: {st['max_rel_cost_err']:.3g} > {COST_RTOL}")
        if not abs(st["mean_abs_delta_ratio"] - 1) <= DELTA_MEAN_RTOL:
            bad.append(f"mean|delta| ratio vs {tag}: {st['mean_abs_delta_ratio']:.5f}")
    if dev_vs_64 is not None and ora32_vs_64 is not None:
        if not dev_vs_64["mean_abs_adv_diff"] <= ADV_DIFF_MARGIN * ora32_vs_64["mean_abs_adv_diff"]:
            bad.append(f"mean|adv - adv_f64|: device {dev_vs_64['mean_abs_adv_diff']:.4g} > {ADV_DIFF_MARGIN} x fp32 oracle's "
                       f"{ora32_vs_64['mean_abs_adv_diff']:.4g}")
        if not dev_vs_64["frac_pixels_within_2lr"] >= ora32_vs_64["frac_pixels_within_2lr"] - PIXEL_FRAC_SLACK:
            bad.append(f"pixels within 2*lr of the f64 run: device {dev_vs_64['frac_pixels_within_2lr']:.4f}, fp32 oracle "
                       f"{ora32_vs_64['frac_pixels_within_2lr']:.4f}")
    return not bad, bad
