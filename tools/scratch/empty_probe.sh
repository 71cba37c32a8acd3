#!/bin/bash
# empty read files / one pair / reads shorter than k: this build against the reference binary
cd /root/repo; W=/tmp/t1k_empty; mkdir -p $W
zcat tests/golden/cyp2d6_rna_seq.fa.gz > $W/ref.fa
zcat tests/golden/cyp_rna_2x100/reads_1.fq.gz > $W/r1.fq; zcat tests/golden/cyp_rna_2x100/reads_2.fq.gz > $W/r2.fq
: > $W/e1.fq; : > $W/e2.fq
head -4 $W/r1.fq > $W/o1.fq; head -4 $W/r2.fq > $W/o2.fq
printf "@s\nACGTACG\n+\nIIIIIII\n" > $W/s1.fq; printf "@s\nTTTTACG\n+\nIIIIIII\n" > $W/s2.fq
for c in e o s; do
  oracle/_ref/genotyper -f $W/ref.fa -1 $W/${c}1.fq -2 $W/${c}2.fq -o $W/ref_$c > $W/ref_$c.log 2>&1; rr=$?
  t1k_amd/bin/genotyper -f $W/ref.fa -1 $W/${c}1.fq -2 $W/${c}2.fq -o $W/gpu_$c > $W/gpu_$c.log 2>&1; rg=$?
  echo "case $c: reference rc $rr, this build rc $rg"
  for s in _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa; do
    if [ -e $W/ref_$c$s ] && [ -e $W/gpu_$c$s ]; then cmp -s $W/ref_$c$s $W/gpu_$c$s && echo "   $s identical ($(stat -c %s $W/gpu_$c$s) B)" || echo "   $s DIFFERS"; else echo "   $s: reference $( [ -e $W/ref_$c$s ] && echo yes || echo no ), this build $( [ -e $W/gpu_$c$s ] && echo yes || echo no )"; fi
  done
  tail -2 $W/gpu_$c.log | cut -c1-200
done
