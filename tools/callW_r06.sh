#!/bin/bash
# Round 6: the cursors of an arena's 32 stripes 128 bytes apart (a cache line each) instead of 64
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callW.log; : > $L
echo "== kernels alone (1 M pairs, one pipeline): main = stripes 64 bytes apart, stripe16 = 128" | tee -a $L
tools/kstats_r06.sh "main stripe16 main stripe16" 1 "k_seed_groups|k_chain_fast<5, [01]|k_extend\(" 2>&1 | tee -a $L
R=$(pwd); V=$R/t1k_amd/lib/variants
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
echo "== bench step, 10 M pairs, three pipelines" | tee -a $L
tools/ab_r06.sh "|T1K_GPU_LIB=$V/libt1k_stripe16.so||T1K_GPU_LIB=$V/libt1k_stripe16.so||T1K_GPU_LIB=$V/libt1k_stripe16.so||T1K_GPU_LIB=$V/libt1k_stripe16.so" 10000000 3 2>&1 | cut -c1-150 | tee -a $L
