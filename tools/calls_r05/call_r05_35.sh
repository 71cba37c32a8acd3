mkdir -p gpurun_out
W=/tmp/t1k_bench; P=10000000; LOG=gpurun_out/r05_c35_windows.log; : > $LOG
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)"
for v in "" "" "" "T1K_FIRST_WINDOW=131072" ""; do
  t0=$(date +%s%N)
  env $v T1K_DEBUG_PHASES=1 T1K_DEBUG_TASKS=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/fw 2> $W/fw.err
  echo "== [$v] $(( ($(date +%s%N) - t0) / 1000000 )) ms wall" >> $LOG
  grep "prep window\|windows, " $W/fw.err | cut -c1-200 >> $LOG
  sleep 5
done
echo "== bench, T1K_FIRST_WINDOW=131072, three steps" >> $LOG
T1K_FIRST_WINDOW=131072 T1K_DEBUG_TASKS=1 python bench.py --pairs $P --steps 2 --warmup 1 --no-cpu-baseline --no-executable-check 2> $W/b.err | tail -1 | cut -c1-200 >> $LOG
grep "prep window" $W/b.err | cut -c1-200 >> $LOG
