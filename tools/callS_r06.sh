#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=gpurun_out/r06_ensure_fix.log; : > $L
echo "== new test on the fixed library" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "later_range_one_entry" 2>&1 | tail -3 | tee -a $L
echo "== the same test with the library as it was (LD_LIBRARY_PATH=t1k_amd/lib/old): must fail" | tee -a $L
LD_LIBRARY_PATH=$(pwd)/t1k_amd/lib/old timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "later_range_one_entry" 2>&1 | grep -E "invalid argument|passed|failed" | head -4 | cut -c1-300 | tee -a $L
echo "== fuzz seed 644 through small windows and ranges, fixed library" | tee -a $L
T1K_FUZZ_SEED=644 T1K_FIRST_WINDOW=64 T1K_WINDOW=512 T1K_BATCH=48 T1K_PAIR_BATCH=96 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "executable" 2>&1 | tail -3 | tee -a $L
for seed in 655 666; do
echo "== fuzz seed $seed, small windows and ranges" | tee -a $L
T1K_FUZZ_SEED=$seed T1K_FIRST_WINDOW=24 T1K_WINDOW=384 T1K_BATCH=40 T1K_PAIR_BATCH=64 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "executable" 2>&1 | tail -3 | tee -a $L
done
