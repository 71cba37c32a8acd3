#!/bin/bash
# Edge-case option sets through the executable, every output file compared with oracle/_ref/genotyper.
set -e
cd /root/repo
W=${W:-/tmp/edge}; mkdir -p $W
tools/t1k_synth ref-rna --seed 21 --genes 8 --scale 0.2 > $W/ref.fa
tools/t1k_synth reads --ref $W/ref.fa --out $W/r --seed 22 --pairs 30000 --len 120 --nrate 0.01 --sub 0.01
run() { # name, args...
  n=$1; shift
  t1k_amd/bin/genotyper "$@" -o $W/ours_$n > $W/ours_$n.log 2>&1 || echo "ours rc=$?"
  oracle/_ref/genotyper "$@" -t 8 -o $W/ref_$n > $W/ref_$n.log 2>&1 || echo "ref rc=$?"
  ok=1; for f in $W/ref_${n}_*; do o=$W/ours_${n}_${f#$W/ref_${n}_}; cmp -s $o $f || { ok=0; echo "DIFF $n ${f##*/}"; }; done; [ $ok = 1 ] && echo "$n: all $(ls $W/ref_${n}_* | wc -l) files identical"
}
run single -f $W/ref.fa -u $W/r_1.fq -s 0.8
run nmax -f $W/ref.fa -1 $W/r_1.fq -2 $W/r_2.fq -s 0.9 -n 50
run frac -f $W/ref.fa -1 $W/r_1.fq -2 $W/r_2.fq -s 0.8 --frac 0.3 --cov 2.0 --crossGeneRate 0.01
run alpha -f $W/ref.fa -1 $W/r_1.fq -2 $W/r_2.fq -s 0.8 --squaremMinAlpha -0.5
head -c 0 /dev/null > $W/empty_1.fq; cp $W/empty_1.fq $W/empty_2.fq
run empty -f $W/ref.fa -1 $W/empty_1.fq -2 $W/empty_2.fq
head -400 $W/r_1.fq > $W/s_1.fq; head -400 $W/r_2.fq > $W/s_2.fq
run tiny -f $W/ref.fa -1 $W/s_1.fq -2 $W/s_2.fq -s 0.8 --outputReadAssignment
