"""Summarise gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>_*.  Run in the build container."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def find(sub, suffix):
    for root, _, files in os.walk(os.path.join(src, sub)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    raise FileNotFoundError((sub, suffix))


shutil.copy(find("stats", "kernel_stats.csv"), f"profiles/{tag}_kernel_stats.csv")
try:
    shutil.copy(find("ilaf", "kernel_stats.csv"), f"profiles/{tag}_ilaf_kernel_stats.csv")
except FileNotFoundError:
    pass


def short(name):
    return name.split("(")[0].replace("void ", "")


ATTACK_MARKS = ("frames_from_video", "compose_kernel", "adam_kernel", "sign_delta")
probes_skipped = {}


def per_kernel(sub, counters):
    """Per-kernel sums over the launches of the ATTACKS: everything dispatched before the first attack kernel of the process is
    planning (uploads, arena clears and the autotuner's probe launches -- since round 3 at four batch sizes per convolution, more
    launches than the attacks themselves and much smaller ones) and is left out of the per-launch averages."""
    rows = list(csv.DictReader(open(find(sub, "counter_collection.csv"))))
    trace = list(csv.DictReader(open(find(sub, "kernel_trace.csv"))))
    kt = {r["Dispatch_Id"]: r for r in trace}
    marks = [int(r["Dispatch_Id"]) for r in trace if any(m in r["Kernel_Name"] for m in ATTACK_MARKS)]
    first = min(marks) if marks else 0
    probes_skipped[sub] = sum(1 for r in trace if int(r["Dispatch_Id"]) < first and "conv_igemm" in r["Kernel_Name"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in rows:
        if int(r["Dispatch_Id"]) < first:
            continue
        n = short(r["Kernel_Name"])
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        d = r["Dispatch_Id"]
        if d not in seen and d in kt:
            seen.add(d)
            agg[n]["_ns"] += int(kt[d]["End_Timestamp"]) - int(kt[d]["Start_Timestamp"])
            agg[n]["_calls"] += 1
    return agg


def attack_only_stats(sub, dst):
    """The --kernel-trace --stats pass again, from its kernel trace, WITHOUT the plan-time launches (same rule as per_kernel):
    name, calls, total ns, average ns -- the average that has to agree with bench.py's live `roofline.avg_launch_us`."""
    trace = list(csv.DictReader(open(find(sub, "kernel_trace.csv"))))
    marks = [int(r["Dispatch_Id"]) for r in trace if any(m in r["Kernel_Name"] for m in ATTACK_MARKS)]
    first = min(marks) if marks else 0
    agg = collections.defaultdict(lambda: [0, 0])
    for r in trace:
        if int(r["Dispatch_Id"]) < first:
            continue
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(a[1] for a in agg.values())
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 2)])
    conv = [a for k, a in agg.items() if "conv_igemm" in k]
    n, t = sum(a[0] for a in conv), sum(a[1] for a in conv)
    return {"conv_igemm_launches": n, "conv_igemm_avg_us": round(t / max(n, 1) / 1e3, 2), "plan_time_launches_left_out": sum(1 for r in trace if int(r["Dispatch_Id"]) < first)}


attack_stats = attack_only_stats("stats", f"profiles/{tag}_kernel_stats_attacks.csv")
try:
    attack_only_stats("ilaf", f"profiles/{tag}_ilaf_kernel_stats_attacks.csv")
except FileNotFoundError:
    pass
fetch, write, mfma = per_kernel("fetch", ["FETCH_SIZE"]), per_kernel("write", ["WRITE_SIZE"]), per_kernel("mfma", [])
out = {}
for n in sorted(fetch, key=lambda k: -fetch[k]["_ns"]):
    calls = fetch[n]["_calls"]
    f_kb, w_kb = fetch[n]["FETCH_SIZE"], write.get(n, {}).get("WRITE_SIZE", 0.0)
    m = mfma.get(n, {})
    t = m.get("_ns", 0) * 1e-9
    out[n] = {"calls": int(calls), "avg_us": round(fetch[n]["_ns"] / calls / 1e3, 2),
              "FETCH_SIZE_KB_per_launch_raw": round(f_kb / calls, 1), "WRITE_SIZE_KB_per_launch": round(w_kb / max(write.get(n, {}).get("_calls", 1), 1), 1),
              "mfma_busy_frac": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * t * 2.4e9), 4) if t else None,
              "clock_GHz": round(m.get("GRBM_GUI_ACTIVE", 0) / 8 / t / 1e9, 3) if t else None}
try:
    bid = open(os.path.join(src, "build_id.txt")).read().strip()
except OSError:
    bid = None
for n, v in out.items():
    m = mfma.get(n, {})
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        v["wave_cycles_waiting_frac"] = round(m.get("SQ_WAIT_ANY", 0) / wc, 3)          # parked at s_waitcnt / s_barrier
        v["wave_cycles_issue_stalled_frac"] = round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3)  # ready but the pipe is busy
    v["hbm_bytes_per_launch"] = round((2 * v["FETCH_SIZE_KB_per_launch_raw"] + v["WRITE_SIZE_KB_per_launch"]) * 1024)
json.dump({"build_id": bid, "stats_pass_attacks_only": attack_stats, "kernels": out}, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
conv = {k: v for k, v in out.items() if k.startswith("conv_igemm")}
calls = sum(v["calls"] for v in conv.values())
fk = sum(v["FETCH_SIZE_KB_per_launch_raw"] * v["calls"] for v in conv.values()) / calls
wk = sum(v["WRITE_SIZE_KB_per_launch"] * v["calls"] for v in conv.values()) / calls
# time-weighted MFMA-busy fraction over every conv_igemm launch of the profiled run (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * t * 2.4 GHz))
tsum = sum(v["avg_us"] * v["calls"] for v in conv.values() if v["mfma_busy_frac"] is not None)
busy = sum(v["mfma_busy_frac"] * v["avg_us"] * v["calls"] for v in conv.values() if v["mfma_busy_frac"] is not None) / tsum if tsum else None
json.dump({"build_id": bid,
           "note": "HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B fetch requests at 64 B: MI355X_MICROARCH.md, HBM); "
                   "separate --pmc passes of `bench.py --steps 2 --warmup 1 --no-kernel-timing`; the plan-time launches in front of the first "
                   "attack kernel (autotuner probes) are left out: " + json.dumps(probes_skipped),
           "conv_igemm": {"launches": calls, "fetch_bytes_per_launch_raw": fk * 1024, "write_bytes_per_launch": wk * 1024,
                          "hbm_bytes_per_launch": (2 * fk + wk) * 1024, "mfma_busy_frac": round(busy, 4) if busy is not None else None},
           "by_instantiation": {k: {"calls": v["calls"], "avg_us": v["avg_us"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                                    "hbm_gbps": round(v["hbm_bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1),
                                    "mfma_busy_frac": v["mfma_busy_frac"]} for k, v in conv.items()}},
          open(f"profiles/{tag}_traffic_by_instantiation.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in list(out)[:8]}, indent=1))
