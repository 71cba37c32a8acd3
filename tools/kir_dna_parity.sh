#!/bin/bash
# KIR-like dna reference (BASELINE.json configs[2] shape, reduced read count): this build vs the reference binary, byte for byte
set -e
cd "$(dirname "$0")/.."
W=${1:-/tmp/t1k_kir}; PAIRS=${2:-200000}
mkdir -p $W
tools/t1k_synth ref-dna --seed 7 --genes 17 --scale 1.0 > $W/kir.fa
tools/t1k_synth reads --ref $W/kir.fa --seed 8 --pairs $PAIRS --len 150 --out $W/r --fragmean 420
grep -c ">" $W/kir.fa
SECONDS=0; t1k_amd/bin/genotyper -f $W/kir.fa -1 $W/r_1.fq -2 $W/r_2.fq -s 0.8 --relaxIntronAlign -o $W/ours > $W/ours.log 2>&1; echo "this build: $SECONDS s"
SECONDS=0; oracle/_ref/genotyper -f $W/kir.fa -1 $W/r_1.fq -2 $W/r_2.fq -s 0.8 --relaxIntronAlign -t 64 -o $W/ref > $W/ref.log 2>&1; echo "reference -t 64: $SECONDS s"
cmp $W/ours_genotype.tsv $W/ref_genotype.tsv && echo "genotype.tsv IDENTICAL"
cmp $W/ours_allele.tsv $W/ref_allele.tsv && echo "allele.tsv IDENTICAL"
tail -3 $W/ours.log
