#!/bin/bash
# Round 5: where a cold `genotyper` process spends the time its in-process step does not.  Stopwatch around: /bin/true, the smallest
# HIP process -- alone, holding 32 / 128 GB of device memory at its end, and having handed them back before it --, the executable (usage only,
# then the bench's job with its phase lines under a few settings).   -> gpurun_out/r05_cold.log
W=/tmp/t1k_bench; P=10000000
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || exit 1
LOG=gpurun_out/r05_cold.log; : > $LOG
ms() { echo $(( ($(date +%s%N) - $1) / 1000000 )); }
t0=$(date +%s%N); /bin/true; echo "/bin/true: $(ms $t0) ms" >> $LOG
for a in ""; do sleep 3; t0=$(date +%s%N); tools/hip_hello $a 2>> $LOG; echo "  hip_hello $a wall: $(ms $t0) ms" >> $LOG; done
t0=$(date +%s%N); t1k_amd/bin/genotyper > /dev/null 2>&1; echo "genotyper without arguments (usage), wall: $(ms $t0) ms" >> $LOG
IFS='|' read -ra VARS <<< "${1:-|}"
for v in "${VARS[@]}"; do
  sleep 15
  t0=$(date +%s%N)
  env $v T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/exe_ph 2> $W/ph.txt
  echo "== ${v:-default}: $(ms $t0) ms wall; md5 $(md5sum < $W/exe_ph_genotype.tsv | cut -c1-8); $(grep 'main:' $W/ph.txt | cut -c11-140); $(grep 'device memory:' $W/ph.txt | cut -c11-120)" >> $LOG
  grep -E "reference: parse|read files mapped|windows, |address space|Threads:|VmPeak|VmHWM|VmPTE|device\+download" $W/ph.txt | cut -c1-300 >> $LOG
done
