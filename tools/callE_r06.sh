#!/bin/bash
# Round 6, fifth GPU call: k_seed_groups with 7 / 6 wavefronts' worth of registers (the LDS admits 7 workgroups a compute unit) and with the read
# offsets packed into the list lengths (560 bytes of LDS less: 8 workgroups a compute unit); then the whole -m gpu suite.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callE.log; : > $L
say() { echo "$@" | tee -a $L; }
R=$(pwd)
say "== k_seed_groups alone (1 M pairs, one pipeline)"
tools/kstats_r06.sh "main seedw7 seedw6 seedpq main seedw7 seedw6 seedpq" 1 "k_seed_groups" 2>&1 | tee -a $L
say "== bench step, 10 M pairs, three pipelines"
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
V=$R/t1k_amd/lib/variants
tools/ab_r06.sh "|T1K_GPU_LIB=$V/libt1k_seedw7.so|T1K_GPU_LIB=$V/libt1k_seedpq.so|T1K_GPU_LIB=$V/libt1k_seedw6.so||T1K_GPU_LIB=$V/libt1k_seedw7.so|T1K_GPU_LIB=$V/libt1k_seedpq.so" 10000000 2 2>&1 | tee -a $L
say "== pytest -m gpu (whole suite)"
( time timeout 1700 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r06_pytest_gpu.log 2>&1
tail -22 gpurun_out/r06_pytest_gpu.log | tee -a $L
