mkdir -p gpurun_out
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag > gpurun_out/r05_c12_thp.txt 2>&1
timeout 900 bash tools/cold_r05.sh "|T1K_NO_EXIT_RELEASE=1|T1K_NO_THP=1||T1K_NO_EXIT_RELEASE=1|T1K_NO_THP=1 T1K_NO_EXIT_RELEASE=1" > gpurun_out/r05_c12_cold.txt 2>&1
cp gpurun_out/r05_cold.log gpurun_out/r05_c12_cold.log
timeout 900 bash tools/ab_r05.sh "|T1K_NO_THP=1||T1K_NO_THP=1" 10000000 2 > gpurun_out/r05_c12_ab.log 2>&1
