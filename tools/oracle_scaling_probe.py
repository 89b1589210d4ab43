#!/usr/bin/env python3
"""Developer tool (round 5, GPU box's HOST): how many CPU oracle processes the box really runs side by side -- one clip's whole fp32 attack
(oracle/fooling_worker.py) per configuration: 1 x 32 threads, 2 x 16, 4 x 8, 2 x 32, pinned to consecutive CPUs or not; prints wall
seconds per configuration and the per-clip seconds the workers report.  The measurement behind the worker counts of
tests/test_gpu_size_parity.py and tools/fooling_parity.py."""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "image-to-video-i2v-attack_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import size_parity  # noqa: E402

print("allowed CPUs:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count(), "effective (quota):", size_parity.effective_cpus())
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(path, open(path).read().strip())
    except OSError:
        pass
configs = [(1, 32, False), (2, 16, False), (4, 8, False), (2, 32, False), (2, 16, True)]
if len(sys.argv) > 1:
    configs = [tuple(int(x) for x in c.split("x")) + (False,) for c in sys.argv[1:]]
row = 300
for workers, threads, pin in configs:
    out = tempfile.mkdtemp(prefix="oscale_")
    rows = list(range(row, row + workers)); row += workers
    t0 = time.time()
    procs = size_parity.start_oracle_workers(rows, out, workers=workers, threads=threads, pin=pin)      # (no float64 rows)
    secs = [size_parity.wait_oracle_row(out, r, procs, timeout=1200)["seconds"] for r in rows]
    wall = time.time() - t0
    for p in procs:
        p.wait()
    print(f"{workers} x {threads} threads{' pinned' if pin else ''}: wall {wall:.1f} s for {workers} clip(s) = {wall / workers:.1f} s per clip; per-clip attack seconds {[round(s, 1) for s in secs]}", flush=True)
    shutil.rmtree(out, ignore_errors=True)
