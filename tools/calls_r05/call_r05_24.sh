mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streamed" --timeout 300 ) 2>&1 | tail -8 > gpurun_out/r05_c24_pytest.log
timeout 1500 bash tools/size_curve_r05.sh "30000000 50000000" > gpurun_out/r05_c24_size.txt 2>&1
