mkdir -p gpurun_out
LOG=gpurun_out/r05_c39.log; : > $LOG
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "leaves_the_strict_layout or streamed" 2>&1 | tail -15 >> $LOG
if grep -q "failed" $LOG; then exit 1; fi
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_pytest_gpu.log
cat gpurun_out/r05_pytest_gpu.log >> $LOG
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $LOG
python bench.py --steps 3 --warmup 1 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
cut -c1-400 gpurun_out/r05_bench.json >> $LOG
