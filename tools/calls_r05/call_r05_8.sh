mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "beyond_the_hit_masks or long_read or a_few_long" --timeout 600 --durations=8 ) 2>&1 | tail -25 > gpurun_out/r05_c8_pytest_long.log
( timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain|k_collect|k_dp_dense|k_near|k_gather|k_arena|k_csort|k_extend|k_general" ) > gpurun_out/r05_c8_kstats.log 2>&1
( T1K_HOST_CHAIN=1 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain|k_collect|k_dp_dense|k_near|k_gather|k_arena|k_csort|k_extend|k_general" ) >> gpurun_out/r05_c8_kstats.log 2>&1
timeout 900 bash tools/cold_r05.sh "|T1K_COVERAGE=eager|T1K_PIPELINES=1||T1K_COVERAGE=eager" > gpurun_out/r05_c8_cold.txt 2>&1
