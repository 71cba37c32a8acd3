#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
echo "== seeding chunk size (alleles per LDS pass): 512 (main, 7 wavefronts) / 256 (7) / 1024 (4 wavefronts' worth of registers: its 35 KB of LDS admit four workgroups); 1 M pairs, one pipeline" | tee gpurun_out/r06_callX.log
tools/kstats_r06.sh "main chunk256 chunk1024 main chunk256 chunk1024" 1 "k_seed_groups|k_collect|k_chain_fast<5, 0" 2>&1 | tee -a gpurun_out/r06_callX.log
