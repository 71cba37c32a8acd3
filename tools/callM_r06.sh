#!/bin/bash
# Round 6: the bench step with fewer atomics on hot words, four rounds of three builds (noho = round-5 hand-outs + k_pair's statistics per workgroup; main = + truncate small x4, select small x2; pho4r32 = main + k_pair hand-out x4, row cursor 32 at a time)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callM.log; : > $L
R=$(pwd); V=$R/t1k_amd/lib/variants
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
tools/ab_r06.sh "T1K_GPU_LIB=$V/libt1k_noho.so||T1K_GPU_LIB=$V/libt1k_pho4r32.so|T1K_GPU_LIB=$V/libt1k_noho.so||T1K_GPU_LIB=$V/libt1k_pho4r32.so|T1K_GPU_LIB=$V/libt1k_noho.so||T1K_GPU_LIB=$V/libt1k_pho4r32.so|T1K_GPU_LIB=$V/libt1k_noho.so||T1K_GPU_LIB=$V/libt1k_pho4r32.so" 10000000 3 2>&1 | tee -a $L
