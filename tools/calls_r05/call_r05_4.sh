mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_c4_pytest.log
timeout 900 bash tools/ab_r05.sh "|T1K_RADIX_SORTS=1||T1K_RADIX_SORTS=1" 10000000 2 > gpurun_out/r05_c4_ab.log 2>&1
timeout 600 bash tools/ranks8_r05.sh > gpurun_out/r05_c4_ranks8.log 2>&1
