#!/usr/bin/env python3
"""Developer tool: where the wall time of an attack goes BETWEEN kernels.  Reads a rocprofv3 --kernel-trace csv, keeps the dispatches from
the first Adam step (`adam_kernel`; ILAF: the second `sign_delta_gx_kernel`) on -- planning, autotuning and the first step's compose / forward / backward are left out, so the window
is "steady state from the end of step 1" (`profiles/r4_gap_probe.txt` was produced with this cut) -- and reports -- per queue and overall -- the span, the sum of kernel durations, and the gaps
between the end of one dispatch and the start of the next (median / mean / total), plus the durations by kernel name.
    python3 tools/gap_probe.py <dir-with-*_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
# everything before the first optimiser step -- Adam's, or ILAF's second sign step (its first call is the warm-up) -- is planning / autotuning
marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]] or [i for i, r in enumerate(rows) if "sign_delta_gx_kernel" in r[2]][1:]
first = marks[0]
rows = rows[first:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _, _ in rows)
print(f"dispatches {len(rows)}, span {span / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms ({100.0 * busy / span:.1f} % of the span)")
# union of busy intervals (kernels of different queues overlap)
cur_s, cur_e, union = rows[0][0], rows[0][1], 0
gaps = []
for s, e, _, _ in rows[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
gaps.sort()
if gaps:
    print(f"device busy (union) {union / 1e6:.2f} ms = {100.0 * union / span:.1f} % of the span; idle gaps: {len(gaps)}, total {sum(gaps) / 1e6:.2f} ms, "
          f"median {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[len(gaps) * 9 // 10] / 1e3:.2f} us, max {gaps[-1] / 1e3:.1f} us")
by = defaultdict(lambda: [0, 0])
for s, e, n, _ in rows:
    k = n.split("(")[0][:70]
    by[k][0] += e - s
    by[k][1] += 1
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {t / 1e6:9.2f} ms  n={n:6d}  avg {t / n / 1e3:8.1f} us  {k}")
