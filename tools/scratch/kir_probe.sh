#!/bin/bash
# 10 M pairs vs the KIR-like dna reference (kir-wgs flags): wall time of the executable with the default windows and with fixed 2 M-fragment ones
W=/tmp/t1k_fs; mkdir -p $W; cd /root/repo
tools/t1k_synth ref-dna --genes 17 --scale 1.0 --seed 20250614 > $W/kir.fa
tools/t1k_synth reads --ref $W/kir.fa --pairs 10000000 --len 150 --seed 3 --out $W/k
for cfg in "A=0" "A=1" "T1K_WINDOW=2097152 T1K_WINDOW_GROWTH=1 T1K_FIRST_WINDOW=262144" "A=2"; do
  ( time env $cfg T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/kir.fa -1 $W/k_1.fq -2 $W/k_2.fq -s 0.9 --relaxIntronAlign -o $W/kp 2> $W/kp.err ) 2>&1 | grep real | tr '\n' ' '
  echo "$cfg | $(grep -o '[0-9]* windows.*device loop [0-9.]* ms' $W/kp.err) | $(md5sum < $W/kp_genotype.tsv | cut -c1-8)"
done
