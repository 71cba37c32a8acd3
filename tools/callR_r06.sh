#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for env in "T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_BATCH=48 T1K_PAIR_BATCH=96" "T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_BATCH=48" "T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_PAIR_BATCH=96" "T1K_BATCH=48 T1K_PAIR_BATCH=96"; do
  echo "=== $env"
  env $env T1K_FUZZ_SEED=644 T1K_DEBUG_TRACE=1 python tools/repro_fuzz_r06.py ref-dna 8 -s 0.9 --relaxIntronAlign 2>&1 | grep -v "^\[t1k trace\] pair\|^\[t1k\] " | tail -12 | cut -c1-300
done
