O=gpurun_out/r6f; mkdir -p $O
for d in 0 2 4 6; do
I2V_FB_DELAY=$d I2V_FUSE_DEBUG=1 timeout 600 python bench.py --workload ilaf --streams 1 --steps 2 --warmup 1 2> $O/err_$d.txt | tail -1 > $O/bench_s1_delay$d.json
grep "i2v fastblock" $O/err_$d.txt | grep -v "launch [0-9]*:" | grep "128 frames" | sort | uniq > $O/autotune_delay$d.txt
done
