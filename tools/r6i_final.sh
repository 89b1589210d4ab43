O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
bash tools/run_r6final.sh > $O/r6final.log 2>&1
