mkdir -p gpurun_out
timeout 900 bash tools/cold_r05.sh "|T1K_NO_PREPIN=1||T1K_NO_PREPIN=1||T1K_NO_PREPIN=1" > gpurun_out/r05_c31_cold.txt 2>&1
cp gpurun_out/r05_cold.log gpurun_out/r05_c31_cold.log
timeout 600 bash tools/ab_r05.sh "|T1K_NO_PREPIN=1" 10000000 2 > gpurun_out/r05_c31_ab.log 2>&1
