mkdir -p gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_c5_pytest.log
( timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain_fast|k_collect|k_select|k_extend" ) > gpurun_out/r05_c5_kstats.log 2>&1
( T1K_XCD_AFFINITY=0 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain_fast|k_collect|k_select|k_extend" ) >> gpurun_out/r05_c5_kstats.log 2>&1
T1K_DEBUG_PHASES=1 T1K_PIPELINES=1 python bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check 2>&1 | grep "again with capacities" | head -5 > gpurun_out/r05_c5_caps.log
timeout 900 bash tools/ab_r05.sh "|T1K_XCD_AFFINITY=0||T1K_XCD_AFFINITY=0" 10000000 2 > gpurun_out/r05_c5_ab.log 2>&1
