#!/bin/bash
# Round 6: the fuzz file on the final code with fresh seeds at twice the size, its executable cases through many small windows as well
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_fuzz_fresh_seeds.log; : > $L
for seed in 611 622 633; do
  echo "== T1K_FUZZ_SEED=$seed T1K_FUZZ_SCALE=2" | tee -a $L
  T1K_FUZZ_SEED=$seed T1K_FUZZ_SCALE=2 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -3 | tee -a $L
done
echo "== seed 644, small windows and ranges" | tee -a $L
T1K_FUZZ_SEED=644 T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_BATCH=48 T1K_PAIR_BATCH=96 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "executable" 2>&1 | tail -3 | tee -a $L
