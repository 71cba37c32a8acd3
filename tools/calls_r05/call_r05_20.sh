mkdir -p gpurun_out
timeout 1200 bash tools/ab_r05.sh "|T1K_BATCH=32768|T1K_BATCH=24576|T1K_BATCH=12288|T1K_PAIR_BATCH=131072|T1K_PAIR_BATCH=32768|" 10000000 2 > gpurun_out/r05_c20_ab.log 2>&1
