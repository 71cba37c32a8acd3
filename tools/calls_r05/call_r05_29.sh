mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -12 > gpurun_out/r05_c29_pytest.log
timeout 600 python bench.py > gpurun_out/r05_c29_bench.json 2> gpurun_out/r05_c29_bench.err
