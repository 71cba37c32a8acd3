#!/bin/bash
# this build (drop-in executable) vs the reference binary on a generated workload, byte for byte.
# usage: stress_parity.sh <workdir> <ref-rna|ref-dna> "<ref args>" "<reads args>" "<genotyper flags>"   (BC=1: pass --barcode <workdir>/r_bc.fa)
set -e
cd "$(dirname "$0")/.."
W=$1; KIND=$2; REFARGS=$3; READARGS=$4; FLAGS=$5
mkdir -p $W
tools/t1k_synth $KIND $REFARGS > $W/ref.fa
tools/t1k_synth reads --ref $W/ref.fa --out $W/r $READARGS
echo "alleles: $(grep -c '>' $W/ref.fa)"
if [ -n "$BC" ]; then FLAGS="$FLAGS --barcode $W/r_bc.fa"; fi
SECONDS=0; t1k_amd/bin/genotyper -f $W/ref.fa -1 $W/r_1.fq -2 $W/r_2.fq $FLAGS -o $W/ours > $W/ours.log 2>&1 || { tail -3 $W/ours.log; echo "THIS BUILD FAILED"; exit 1; }
echo "this build: $SECONDS s"
SECONDS=0; oracle/_ref/genotyper -f $W/ref.fa -1 $W/r_1.fq -2 $W/r_2.fq $FLAGS -t 64 -o $W/ref > $W/ref.log 2>&1; echo "reference -t 64: $SECONDS s"
for f in $W/ref_*; do o=$W/ours_${f#$W/ref_}; cmp $o $f && echo "${f#$W/ref_} IDENTICAL ($(stat -c %s $f) bytes)"; done
grep "can be assigned" $W/ours.log | cut -c28-
