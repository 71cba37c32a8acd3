#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_fuzz_small_windows_failure.log; : > $L
T1K_FUZZ_SEED=644 T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_BATCH=48 T1K_PAIR_BATCH=96 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "executable and ref-dna-flags1-8" 2>&1 | tail -60 | cut -c1-400 | tee -a $L
echo "== same seed, default windows" | tee -a $L
T1K_FUZZ_SEED=644 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "executable" 2>&1 | tail -3 | tee -a $L
echo "== small windows, k_pair reserve/hand-out irrelevant? T1K_REF_TRANSPOSE=0" | tee -a $L
T1K_REF_TRANSPOSE=0 T1K_FUZZ_SEED=644 T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_BATCH=48 T1K_PAIR_BATCH=96 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "executable" 2>&1 | tail -4 | tee -a $L
echo "== small windows, seed 101 (a seed round 5 ran this way)" | tee -a $L
T1K_FUZZ_SEED=101 T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_BATCH=48 T1K_PAIR_BATCH=96 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "executable" 2>&1 | tail -4 | tee -a $L
