mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_c7_pytest.log
timeout 900 bash tools/ab_r05.sh "|T1K_HOST_CHAIN=1|T1K_PIPELINES=2|T1K_PIPELINES=2 T1K_HOST_CHAIN=1||T1K_HOST_CHAIN=1" 10000000 2 > gpurun_out/r05_c7_ab.log 2>&1
timeout 600 bash tools/cold_r05.sh > gpurun_out/r05_c7_cold.txt 2>&1
