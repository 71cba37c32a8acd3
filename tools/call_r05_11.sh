mkdir -p gpurun_out
LOG=gpurun_out/r05_c11_exit.txt; : > $LOG
ms() { echo $(( ($(date +%s%N) - $1) / 1000000 )); }
for v in "" "HH_STREAMS=8" "HH_STREAMS=8 HH_MAPS=1" "HH_ANON_GB=3" "HH_STREAMS=8 HH_ANON_GB=3" "HH_STREAMS=16" "" "HH_STREAMS=8" "HH_ANON_GB=3"; do
  sleep 2; t0=$(date +%s%N); env $v tools/hip_hello 2>> $LOG; echo "  [$v] hip_hello wall: $(ms $t0) ms" >> $LOG
done
