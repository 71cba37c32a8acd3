#!/bin/bash
# empty inputs through the neighbours of the path: fastq-extractor and analyzer, this build against the reference binaries
cd /root/repo; W=/tmp/t1k_empty2; mkdir -p $W
zcat tests/golden/cyp2d6_rna_seq.fa.gz > $W/ref.fa
: > $W/e1.fq; : > $W/e2.fq; : > $W/allele.tsv
for who in ref gpu; do
  if [ $who = ref ]; then X=oracle/_ref/fastq-extractor; A=oracle/_ref/analyzer; else X=t1k_amd/bin/fastq-extractor; A=t1k_amd/bin/analyzer; fi
  $X -f $W/ref.fa -1 $W/e1.fq -2 $W/e2.fq -o $W/x_$who > $W/x_$who.log 2>&1; echo "$who extractor rc $?: $(ls $W | grep "^x_${who}_" | tr '\n' ' ')"
  $A -f $W/ref.fa -a $W/allele.tsv -1 $W/e1.fq -2 $W/e2.fq -o $W/a_$who > $W/a_$who.log 2>&1; echo "$who analyzer rc $?: $(ls $W | grep "^a_${who}_" | tr '\n' ' ')"
done
for f in $(ls $W | grep "^x_ref_\|^a_ref_"); do g=${f/_ref_/_gpu_}; cmp -s $W/$f $W/$g && echo "$f identical ($(stat -c %s $W/$f) B)" || echo "$f DIFFERS or missing"; done
tail -2 $W/a_gpu.log; tail -2 $W/x_gpu.log
: > $W/bc.fa
oracle/_ref/analyzer -f $W/ref.fa -a $W/allele.tsv -1 $W/e1.fq -2 $W/e2.fq --barcode $W/bc.fa -o $W/b_ref > $W/b_ref.log 2>&1; echo "ref analyzer+barcode rc $?: $(ls -la $W | grep "b_ref_" | awk '{print $9":"$5}' | tr '\n' ' ')"
stat -c "%n %s" $W/a_ref_allele.vcf; tail -3 $W/b_ref.log
# non-empty reads but empty allele list
zcat tests/golden/cyp_rna_2x100/reads_1.fq.gz > $W/r1.fq; zcat tests/golden/cyp_rna_2x100/reads_2.fq.gz > $W/r2.fq
oracle/_ref/analyzer -f $W/ref.fa -a $W/allele.tsv -1 $W/r1.fq -2 $W/r2.fq -o $W/c_ref > $W/c_ref.log 2>&1; echo "ref analyzer (reads, no alleles) rc $?: $(ls -la $W | grep "c_ref_" | awk '{print $9":"$5}' | tr '\n' ' ')"
