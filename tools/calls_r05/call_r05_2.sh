mkdir -p gpurun_out
R=$(pwd)
export T1K_DEBUG_PHASES=1
( timeout 300 bash tools/kstats_r05.sh "main fw4" 1 "k_seed|k_chain_fast|k_collect|rocprim" ) > gpurun_out/r05_c2_kstats.log 2>&1
( T1K_SORT_SLOW=1 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain_fast|k_collect|rocprim" ) >> gpurun_out/r05_c2_kstats.log 2>&1
( T1K_FUSE_SEED=0 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain_fast|k_collect|rocprim" ) >> gpurun_out/r05_c2_kstats.log 2>&1
( T1K_FUSE_SEED=0 T1K_SORT_SLOW=1 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain_fast|k_collect|rocprim" ) >> gpurun_out/r05_c2_kstats.log 2>&1
# counts of one range, fused and not
T1K_PIPELINES=1 python bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check 2>&1 | grep "gap walk" | head -3 > gpurun_out/r05_c2_counts.log
T1K_FUSE_SEED=0 T1K_PIPELINES=1 python bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check 2>&1 | grep "gap walk" | head -3 >> gpurun_out/r05_c2_counts.log
unset T1K_DEBUG_PHASES
timeout 900 bash tools/ab_r05.sh "|T1K_FUSE_SEED=0" 10000000 2 > gpurun_out/r05_c2_ab.log 2>&1
