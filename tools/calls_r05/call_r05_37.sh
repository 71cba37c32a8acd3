mkdir -p gpurun_out
W=/tmp/t1k_bench; P=10000000; LOG=gpurun_out/r05_c37_windows.log; : > $LOG; mkdir -p $W
for v in "T1K_FIRST_WINDOW=131072" ""; do
  echo "== bench [$v]" >> $LOG
  env $v T1K_DEBUG_TASKS=1 python bench.py --pairs $P --steps 1 --warmup 1 --no-cpu-baseline --no-executable-check 2> $W/b.err | tail -1 | cut -c1-180 >> $LOG
  grep "prep window\|window . done" $W/b.err | cut -c1-260 >> $LOG
done
