#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
echo "== k_seed_groups alone (1 M pairs, one pipeline): main = compare-and-swap per posting; seedpeek = a plain LDS read first, the compare-and-swap only where the accumulator still looks empty" | tee gpurun_out/r06_callY.log
tools/kstats_r06.sh "main seedpeek main seedpeek" 1 "k_seed_groups" 2>&1 | tee -a gpurun_out/r06_callY.log
