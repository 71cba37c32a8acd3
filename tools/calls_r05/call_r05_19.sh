mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gzip or gz or lanes" --timeout 600 ) 2>&1 | tail -6 > gpurun_out/r05_c19_pytest.log
timeout 1200 bash tools/gz_r05.sh > gpurun_out/r05_c19_gz.txt 2>&1
