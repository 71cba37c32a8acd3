#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
echo "== k_seed_groups alone (1 M pairs, one pipeline; 72 VGPRs): postings in flight per lane 4 (main) / 2 / 6 / 8" | tee gpurun_out/r06_callU.log
tools/kstats_r06.sh "main seedif2 seedif6 seedif8 main seedif6 seedif8" 1 "k_seed_groups" 2>&1 | tee -a gpurun_out/r06_callU.log
