import sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for line in open(sys.argv[1]):
    k, Cd, K, HWg, frames, pw, ms, gf = line.split()
    key = (int(k), int(Cd), int(K), int(HWg), int(frames), int(pw))
    a = agg[key]; a[0] += 1; a[1] += float(ms); a[2] += float(gf)
tot = sum(a[1] for a in agg.values())
print("total ms", round(tot, 2))
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(key, "n=%d ms=%.2f (%.1f%%) avg_us=%.1f TF=%.1f" % (a[0], a[1], 100 * a[1] / tot, 1e3 * a[1] / a[0], a[2] / a[1] if a[1] else 0))
