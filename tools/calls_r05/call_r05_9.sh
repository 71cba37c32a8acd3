mkdir -p gpurun_out
timeout 900 bash tools/ab_r05.sh "|T1K_HOST_CHAIN=1||T1K_HOST_CHAIN=1|T1K_PIPELINES=2|T1K_PIPELINES=4" 10000000 2 > gpurun_out/r05_c9_ab.log 2>&1
timeout 900 bash tools/cold_r05.sh "|T1K_NO_WARM=1||T1K_NO_WARM=1|T1K_HOST_CHAIN=1" > gpurun_out/r05_c9_cold.txt 2>&1
