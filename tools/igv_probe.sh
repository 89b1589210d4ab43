# Round 6 probe (GPU box, EXPERIMENTAL build): conv_igvfma_kernel with its weight rows pinned to one row (bit 0: every scalar load a cache hit) and / or
# without operand loads (bit 1) -- timing only, the results are wrong.     python __graft_entry__.py --experimental; bash tools/igv_probe.sh
O=gpurun_out/igv_probe; mkdir -p $O
export I2V_LIB=$PWD/image-to-video-i2v-attack_amd/i2v_amd/libi2v_hip_exp.so
for pr in 0 1 2 3; do
I2V_IGV_PROBE=$pr I2V_TIMING_DUMP=/tmp/dp$pr timeout 600 python bench.py --workload ilaf --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; echo "I2V_IGV_PROBE=$pr: $(python tools/timing_dump_agg.py /tmp/dp$pr 40 | grep '(1, 12, 640')"
done > $O/summary.txt
cat $O/summary.txt
