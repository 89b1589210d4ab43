# Round 6, last pass (GPU box): the whole GPU suite, the smoke test, then everything profiles/r6_* is made from.     bash tools/r6_final_all.sh
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
bash tools/run_r6final.sh > $O/r6final.log 2>&1
bash tools/pmc_fastblock.sh > $O/pmc_fastblock.log 2>&1; find gpurun_out/pmc_fastblock -name "*.csv" -delete
