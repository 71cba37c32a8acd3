mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streamed" --timeout 600 ) 2>&1 | tail -8 > gpurun_out/r05_c22_pytest.log
timeout 600 python bench.py > gpurun_out/r05_c22_bench.json 2> gpurun_out/r05_c22_bench.err
