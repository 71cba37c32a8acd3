mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gzip or gz or odd_but_legal or several_files" --timeout 600 ) 2>&1 | tail -15 > gpurun_out/r05_c13_pytest.log
timeout 1200 bash tools/gz_r05.sh > gpurun_out/r05_c13_gz.txt 2>&1
