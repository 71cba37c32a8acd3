mkdir -p gpurun_out
( timeout 300 bash tools/kstats_r05.sh "main" 1 "k_chain_general|k_chain_wave" ) > gpurun_out/r05_c18_kstats.log 2>&1
( T1K_WAVE_SMALL=0 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_chain_general|k_chain_wave" ) >> gpurun_out/r05_c18_kstats.log 2>&1
( T1K_WAVE_SMALL=128 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_chain_general|k_chain_wave" ) >> gpurun_out/r05_c18_kstats.log 2>&1
timeout 900 bash tools/ab_r05.sh "|T1K_WAVE_SMALL=0||T1K_WAVE_SMALL=0" 10000000 2 > gpurun_out/r05_c18_ab.log 2>&1
timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q --timeout 600 2>&1 | tail -3 > gpurun_out/r05_c18_pytest.log
