mkdir -p gpurun_out
timeout 900 bash tools/cold_r05.sh "|||" > gpurun_out/r05_c32_cold.txt 2>&1
cp gpurun_out/r05_cold.log gpurun_out/r05_c32_cold.log
timeout 600 bash tools/ab_r05.sh "|" 10000000 2 > gpurun_out/r05_c32_ab.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_windows or pipeline or golden" --timeout 300 2>&1 | tail -4 > gpurun_out/r05_c32_pytest.log
