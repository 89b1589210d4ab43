"""Aggregate an I2V_TIMING_DUMP trace (one line per backbone launch: kind Cd K pixels-per-frame frames pointwise ms GFLOP [MB])
by launch shape.  kind: 0 conv fwd, 5 conv input-gradient, 1 image gradient, 2/3 pooling, 4 addmask/memset.
    I2V_TIMING_DUMP=/tmp/d python bench.py --steps 3 --no-cpu-baseline; python tools/timing_dump_agg.py /tmp/d [rows]"""
import collections
import sys

agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for line in open(sys.argv[1]):
    f = line.split()
    k, Cd, K, HWg, frames, pw, ms, gf = f[:8]
    mb = float(f[8]) if len(f) > 8 else 0.0
    key = (int(k), int(Cd), int(K), int(HWg), int(frames), int(pw))
    a = agg[key]; a[0] += 1; a[1] += float(ms); a[2] += float(gf); a[3] += mb
tot = sum(a[1] for a in agg.values())
print("total ms", round(tot, 2))
print("(kind, Cd, K, px/frame, frames, pointwise)  launches  ms  share  avg_us  TFLOP/s  algorithmic TB/s  FLOP/B")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    tbs = a[3] / a[1] * 1e-3 if a[1] else 0          # MB / ms = GB/s -> TB/s
    print(key, "n=%d ms=%.2f (%.1f%%) avg_us=%.1f TF=%.1f TB/s=%.2f FLOP/B=%.1f" %
          (a[0], a[1], 100 * a[1] / tot, 1e3 * a[1] / a[0], a[2] / a[1] if a[1] else 0, tbs, (a[2] * 1e3 / a[3]) if a[3] else 0))
