#!/bin/bash
# Full-size parity of round 2 (run on the GPU box): this build's executables against the reference binaries built by oracle/Makefile,
# every output file byte for byte.
#   A. BASELINE configs[1]: 1 M 2x150 pairs vs the HLA-like rna reference, -s 0.97             (single GPU, and 2 ranks sharing the GPU)
#   B. BASELINE configs[2]: 10 M 2x150 pairs vs the KIR-like dna reference, --preset kir-wgs = -s 0.9 --relaxIntronAlign (run-t1k:300-304)
cd "$(dirname "$0")/.."
W=/tmp/t1k_fs; mkdir -p $W gpurun_out
L=gpurun_out/r02_full_size_parity${PARTS:+_$PARTS}.log; : > $L
say() { echo "$@" | tee -a $L; }
cmpall() { for s in _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa; do if cmp -s $1$s $2$s; then say "   $s IDENTICAL ($(stat -c %s $1$s) bytes)"; else say "   $s DIFFERS"; fi; done; }
tools/t1k_synth ref-rna --genes 24 --scale 1.0 --seed 20250614 > $W/hla.fa
tools/t1k_synth reads --ref $W/hla.fa --pairs 1000000 --len 150 --seed 2 --out $W/h
say "A. 1 M pairs, HLA-like rna ($(grep -c '>' $W/hla.fa) alleles), -s 0.97"
SECONDS=0; t1k_amd/bin/genotyper -f $W/hla.fa -1 $W/h_1.fq -2 $W/h_2.fq -s 0.97 -o $W/ours 2> $W/ours.log; say "   this build, 1 GPU: $SECONDS s (rc $?)"
SECONDS=0; T1K_GPUS=0,0 t1k_amd/bin/genotyper -f $W/hla.fa -1 $W/h_1.fq -2 $W/h_2.fq -s 0.97 -o $W/ours2 2> $W/ours2.log; say "   this build, 2 ranks on one GPU: $SECONDS s (rc $?)"
SECONDS=0; T1K_GPUS=0,0,0 T1K_SHARD_INPUT=1 t1k_amd/bin/genotyper -f $W/hla.fa -1 $W/h_1.fq -2 $W/h_2.fq -s 0.97 -o $W/ours3 2> $W/ours3.log; say "   this build, 3 ranks on one GPU, each indexing and writing only its own reads: $SECONDS s (rc $?)"
SECONDS=0; oracle/_ref/genotyper -f $W/hla.fa -1 $W/h_1.fq -2 $W/h_2.fq -s 0.97 -t 64 -o $W/ref 2> $W/ref.log; say "   reference -t 64: $SECONDS s"
say "  1 GPU vs reference:"; cmpall $W/ours $W/ref
say "  2 ranks vs reference:"; cmpall $W/ours2 $W/ref
say "  3 ranks with their own input vs reference:"; cmpall $W/ours3 $W/ref
say "  EM iterations: ours $(grep -o 'in [0-9]* EM' $W/ours.log) / sharded $(grep -o 'in [0-9]* EM' $W/ours2.log) / reference $(grep -o 'in [0-9]* EM' $W/ref.log)"
[ "$PARTS" = "A" ] && exit 0
tools/t1k_synth ref-dna --genes 17 --scale 1.0 --seed 20250614 > $W/kir.fa
tools/t1k_synth reads --ref $W/kir.fa --pairs 10000000 --len 150 --seed 3 --out $W/k
say "B. 10 M pairs, KIR-like dna ($(grep -c '>' $W/kir.fa) alleles), -s 0.9 --relaxIntronAlign (kir-wgs)"
SECONDS=0; t1k_amd/bin/genotyper -f $W/kir.fa -1 $W/k_1.fq -2 $W/k_2.fq -s 0.9 --relaxIntronAlign -o $W/kours 2> $W/kours.log; say "   this build: $SECONDS s (rc $?)"
SECONDS=0; oracle/_ref/genotyper -f $W/kir.fa -1 $W/k_1.fq -2 $W/k_2.fq -s 0.9 --relaxIntronAlign -t 64 -o $W/kref 2> $W/kref.log; say "   reference -t 64: $SECONDS s"
cmpall $W/kours $W/kref
say "  EM iterations: ours $(grep -o 'in [0-9]* EM' $W/kours.log) / reference $(grep -o 'in [0-9]* EM' $W/kref.log)"
