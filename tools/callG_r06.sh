#!/bin/bash
# Round 6, seventh GPU call: register budgets against what each kernel's LDS admits (one-pipeline kernel times; occA / occB: tools/build_variant.sh lines in DESIGN 9.0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callG.log; : > $L
echo "== kernels alone (1 M pairs, one pipeline): occA = collect 7, extend 7, closed form 7, truncate small 6 / large 8, select small 7 wavefronts per SIMD; occB = collect 8, extend 8, closed form 8, truncate small 7 / large 8, select small 7" | tee -a $L
tools/kstats_r06.sh "main occA occB main occA occB" 1 "k_collect|k_extend\(|k_chain_fast<5, 0|k_truncate|k_select<2048" 2>&1 | tee -a $L
R=$(pwd); V=$R/t1k_amd/lib/variants
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
echo "== bench step, 10 M pairs, three pipelines" | tee -a $L
tools/ab_r06.sh "|T1K_GPU_LIB=$V/libt1k_occA.so|T1K_GPU_LIB=$V/libt1k_occB.so||T1K_GPU_LIB=$V/libt1k_occA.so|T1K_GPU_LIB=$V/libt1k_occB.so" 10000000 2 2>&1 | tee -a $L
