mkdir -p gpurun_out
LOG=gpurun_out/r05_c11_exit.txt; : > $LOG
ms() { echo $(( ($(date +%s%N) - $1) / 1000000 )); }
for v in "" "HH_ANON_GB=1" "HH_ANON_GB=3" "HH_ANON_GB=3 HH_ZAP=1" "HH_ANON_GB=3 HH_ZAP=8" "HH_ANON_GB=3 HH_ZAP=32" "HH_ANON_GB=3" "HH_ANON_GB=3 HH_ZAP=16" ""; do
  sleep 2; t0=$(date +%s%N); env $v tools/hip_hello 2>> $LOG; echo "  [$v] hip_hello wall: $(ms $t0) ms" >> $LOG
done
