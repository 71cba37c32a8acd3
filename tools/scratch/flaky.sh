#!/bin/bash
mkdir -p /tmp/fl && cd /tmp/fl
/root/repo/tools/t1k_synth ref-dna --genes 17 --scale 1.0 --seed 20250614 > ref.fa
/root/repo/tools/t1k_synth reads --ref ref.fa --pairs 100000 --len 150 --seed 2 --out r
for i in $(seq 1 24); do
  T1K_DEBUG_PHASES=1 /root/repo/t1k_amd/bin/genotyper -f ref.fa -1 r_1.fq -2 r_2.fq -s 0.9 --relaxIntronAlign -o o$i 2> e$i.log; T1K_DEBUG_PHASES=1 true; echo "run $i rc=$? $(md5sum < o${i}_genotype.tsv | cut -c1-8) $(md5sum < o${i}_allele.tsv | cut -c1-8) $(md5sum < o${i}_aligned_1.fa | cut -c1-8) $(grep -c . e$i.log)"
done
grep -h "genotyper:" e*.log | sort | uniq -c
grep -h "again with" e*.log | sort | uniq -c | sort -k1,1nr | head -8
