#!/bin/bash
# Round 5: where the cold executable's time beyond main goes, and whether handing the device memory back from several threads before leaving
# (T1K_EXIT_FREE=<threads>, then _exit) beats the process's own teardown.  Stopwatch around the process, 20 s apart (the driver wipes what the
# previous process returned).   -> gpurun_out/r05_exit_ab.log
W=/tmp/t1k_bench; P=10000000
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || exit 1
LOG=gpurun_out/r05_exit_ab.log; : > $LOG
for v in "" "T1K_EXIT_FREE=1" "T1K_EXIT_FREE=8" "T1K_EXIT_FREE=32" "" "T1K_EXIT_FREE=8"; do
  sleep 20
  t0=$(date +%s%N)
  env $v T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/exe_ph 2> $W/ph.txt
  echo "${v:-plain exit}: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall; $(grep 'main:' $W/ph.txt | cut -c1-120) $(grep 'exit:' $W/ph.txt | cut -c1-120); md5 $(md5sum < $W/exe_ph_genotype.tsv | cut -c1-8)" | tee -a $LOG
done
