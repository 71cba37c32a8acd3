#!/bin/bash
mkdir -p /tmp/fl && cd /tmp/fl
/root/repo/tools/t1k_synth ref-rna --genes 24 --scale 1.0 --seed 20250614 > hla.fa
/root/repo/tools/t1k_synth reads --ref hla.fa --pairs 100000 --len 150 --seed 2 --out h
/root/repo/tools/t1k_synth ref-dna --genes 17 --scale 1.0 --seed 20250614 > kir.fa
/root/repo/tools/t1k_synth reads --ref kir.fa --pairs 100000 --len 150 --seed 2 --out k
ulimit -c 0
for i in $(seq 1 30); do
  /root/repo/t1k_amd/bin/genotyper -f hla.fa -1 h_1.fq -2 h_2.fq -s 0.97 -o oh$i 2> eh$i.log; rc1=$?
  /root/repo/t1k_amd/bin/genotyper -f kir.fa -1 k_1.fq -2 k_2.fq -s 0.9 --relaxIntronAlign -o ok$i 2> ek$i.log; rc2=$?
  echo "run $i hla rc=$rc1 $(md5sum < oh${i}_genotype.tsv 2>/dev/null | cut -c1-8) kir rc=$rc2 $(md5sum < ok${i}_genotype.tsv 2>/dev/null| cut -c1-8)"
  if [ $rc1 != 0 ]; then tail -5 eh$i.log; fi
  if [ $rc2 != 0 ]; then tail -5 ek$i.log; fi
done
dmesg 2>/dev/null | tail -5
