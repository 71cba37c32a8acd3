mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 900 ) 2>&1 | tail -12 > gpurun_out/r05_c17_pytest.log
( timeout 300 bash tools/kstats_r05.sh "main" 1 "k_chain_general|k_near_hits|k_gather|k_chain_wave|k_csort_scatter<Group" ) > gpurun_out/r05_c17_kstats.log 2>&1
( T1K_NO_SIMPLE_KERNEL=1 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_chain_general|k_near_hits|k_gather|k_chain_wave|k_csort_scatter<Group" ) >> gpurun_out/r05_c17_kstats.log 2>&1
timeout 900 bash tools/ab_r05.sh "|T1K_NO_SIMPLE_KERNEL=1||T1K_NO_SIMPLE_KERNEL=1" 10000000 2 > gpurun_out/r05_c17_ab.log 2>&1
