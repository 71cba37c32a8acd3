mkdir -p gpurun_out
W=/tmp/t1k_bench; P=10000000; LOG=gpurun_out/r05_c38_windows.log; : > $LOG; mkdir -p $W
for v in "T1K_FIRST_WINDOW=131072" "T1K_FIRST_WINDOW=65536" ""; do
  echo "== bench [$v]" >> $LOG
  env $v T1K_DEBUG_TASKS=1 python bench.py --pairs $P --steps 2 --warmup 1 --no-cpu-baseline --no-executable-check 2> $W/b.err | tail -1 | cut -c1-180 >> $LOG
  grep "window . done" $W/b.err | cut -c1-260 >> $LOG
  sleep 15
done
timeout 900 python -m pytest tests/test_gpu_coverage.py -m gpu -x -q 2>&1 | tail -3 >> $LOG
