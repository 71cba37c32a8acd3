mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_c1_pytest.log
R=$(pwd)
timeout 600 bash tools/kstats_r05.sh "main fw4 fw6" 1 "k_seed|k_chain_fast|k_collect" > gpurun_out/r05_c1_kstats.log 2>&1
T1K_FUSE_SEED=0 timeout 300 bash tools/kstats_r05.sh "main" 1 "k_seed|k_chain_fast|k_collect" >> gpurun_out/r05_c1_kstats.log 2>&1
timeout 900 bash tools/ab_r05.sh "|T1K_FUSE_SEED=0|T1K_GPU_LIB=$R/t1k_amd/lib/variants/libt1k_fw4.so|T1K_GPU_LIB=$R/t1k_amd/lib/variants/libt1k_fw6.so" 10000000 2 > gpurun_out/r05_c1_ab.log 2>&1
