#!/bin/bash
# Round 6 (same script as round 5: regression run of the round-4 size curve on the device-driven loop): job size against time on ONE GPU, past the memory budget of the kept read sets (half the device, T1K_ARCHIVE_GB): windows beyond it
# fall back to the per-range coverage updates.  Cold executable per size, phase lines kept.  tools/size_curve_r04.sh "1000000 3000000 ..."
W=/tmp/t1k_bench; mkdir -p $W gpurun_out
LOG=gpurun_out/r06_size_curve.log; : > $LOG
for P in ${1:-1000000 3000000 10000000 20000000 30000000 50000000}; do
  need=$(( P * 1000 )); free=$(df --output=avail -B1 $W | tail -1)
  if [ "$free" -lt "$need" ]; then echo "$P pairs: skipped, $free bytes free on $W" >> $LOG; continue; fi
  python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || { echo "$P pairs: input generation failed" >> $LOG; continue; }
  t0=$(date +%s%N)
  T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/sz 2> $W/sz.err; rc=$?
  ms=$(( ($(date +%s%N) - t0) / 1000000 ))
  echo "== $P pairs: rc $rc, $ms ms wall = $(( P * 1000 / ms )) pairs/s, genotype md5 $(md5sum < $W/sz_genotype.tsv | cut -c1-8)" >> $LOG
  grep "windows,\|read sets of\|read sets are not kept\|device memory\|device+download\|read files mapped\|^genotyper:" $W/sz.err | cut -c1-260 >> $LOG
  [ $rc -ne 0 ] && grep -v "^\[t1k\] \(fullalign\|range\|band\|equal\)" $W/sz.err | tail -6 | cut -c1-400 >> $LOG
  rm -f $W/sz_aligned_1.fa $W/sz_aligned_2.fa
  [ $P -gt 10000000 ] && rm -f $W/reads_g24_s1.0_p${P}_seed2_*.fq $W/reads_g24_s1.0_p${P}_seed2_truth.tsv
  sleep 10
done
cat $LOG
