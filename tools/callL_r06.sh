#!/bin/bash
# Round 6: atomics on hot words -- k_pair's hand-out in chunks, its statistics per workgroup (main), row-cursor reservations; chunked hand-out of read-ends in k_collect / k_select / k_truncate
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callL.log; : > $L
echo "== k_pair alone (1 M pairs, one pipeline; round 5's kernel: 1.343 ms): main = statistics per workgroup, hand-out 1; phoN = hand-out N at a time; rM = row cursor reserved M entries at a time" | tee -a $L
tools/kstats_r06.sh "main pho2 pho4 pho4r32 pho4r64 pho8r64 main pho4 pho4r32 pho4r64" 1 "k_pair" 2>&1 | tee -a $L
echo "== read-end hand-out N at a time (k_collect, k_select, k_truncate)" | tee -a $L
tools/kstats_r06.sh "main reho2 reho4 reho8 main reho2 reho4 reho8" 1 "k_collect|k_select|k_truncate" 2>&1 | tee -a $L
R=$(pwd); V=$R/t1k_amd/lib/variants
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
echo "== bench step, 10 M pairs, three pipelines" | tee -a $L
tools/ab_r06.sh "|T1K_GPU_LIB=$V/libt1k_pho4r32.so|T1K_GPU_LIB=$V/libt1k_pho4r64.so|T1K_GPU_LIB=$V/libt1k_reho4.so||T1K_GPU_LIB=$V/libt1k_pho4r32.so|T1K_GPU_LIB=$V/libt1k_pho4r64.so|T1K_GPU_LIB=$V/libt1k_reho4.so" 10000000 2 2>&1 | tee -a $L
