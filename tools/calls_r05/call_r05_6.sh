mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -15 > gpurun_out/r05_c6_pytest.log
timeout 600 bash tools/exit_ab_r05.sh > gpurun_out/r05_c6_exit_ab.txt 2>&1
timeout 600 python bench.py > gpurun_out/r05_c6_bench.json 2> gpurun_out/r05_c6_bench.err
