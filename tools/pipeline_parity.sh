#!/bin/bash
# Both stages back to back, as run-t1k chains them (run-t1k:377-434): raw reads -> fastq-extractor -> candidate reads -> genotyper,
# ours (t1k_amd/bin/*) against the reference's own binaries (oracle/_ref/*).  Every file of both stages must be identical.
# Usage: tools/pipeline_parity.sh [pairs]
set -e
cd "$(dirname "$0")/.."
W=${W:-/tmp/pipeline_parity}; mkdir -p $W
P=${1:-300000}
tools/t1k_synth ref-rna --seed 51 --genes 10 --scale 0.3 > $W/ref.fa
tools/t1k_synth reads --ref $W/ref.fa --out $W/raw --seed 52 --pairs $P --len 150 --bg 0.9 --sub 0.004
fail=0
t0=$(date +%s.%N)
timeout 600 t1k_amd/bin/fastq-extractor -f $W/ref.fa -1 $W/raw_1.fq -2 $W/raw_2.fq -t 8 -o $W/ours_cand 2> $W/ours_x.log
timeout 600 t1k_amd/bin/genotyper -f $W/ref.fa -1 $W/ours_cand_1.fq -2 $W/ours_cand_2.fq -t 8 -s 0.97 -o $W/ours > $W/ours_g.log 2>&1
t1=$(date +%s.%N)
timeout 900 oracle/_ref/fastq-extractor -f $W/ref.fa -1 $W/raw_1.fq -2 $W/raw_2.fq -t 64 -o $W/ref_cand 2> $W/ref_x.log
timeout 1800 oracle/_ref/genotyper -f $W/ref.fa -1 $W/ref_cand_1.fq -2 $W/ref_cand_2.fq -t 64 -s 0.97 -o $W/ref > $W/ref_g.log 2>&1
t2=$(date +%s.%N)
for f in cand_1.fq cand_2.fq genotype.tsv allele.tsv aligned_1.fa aligned_2.fa; do
  if cmp -s $W/ours_$f $W/ref_$f; then echo "identical: $f ($(wc -l < $W/ref_$f) lines)"; else echo "DIFF: $f"; fail=1; fi
done
echo "ours: $(python3 -c "print('%.1f' % ($t1 - $t0))") s   reference (-t 64): $(python3 -c "print('%.1f' % ($t2 - $t1))") s   ($P raw pairs, $(( $(wc -l < $W/ref_cand_1.fq) / 4 )) candidates)"
exit $fail
