#!/bin/bash
# What an odd long read costs: 1 M synthetic 2x150 bp pairs against the HLA-like reference, with and without ten 2x500 bp pairs in the
# middle of the files (one window of the job then takes the long path).  Prints the job lines of both runs.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
D=${TMPDIR:-/tmp}/t1k_lrc; mkdir -p $D; cd $D
$R/tools/t1k_synth ref-rna > ref.fa
$R/tools/t1k_synth reads --ref ref.fa --out s --pairs 1000000 --len 150 --seed 3
$R/tools/t1k_synth reads --ref ref.fa --out l --pairs 10 --len 500 --seed 4 --fragmean 1040
for m in 1 2; do
  head -n 2000000 s_$m.fq > mix_$m.fq; cat l_$m.fq >> mix_$m.fq; tail -n +2000001 s_$m.fq >> mix_$m.fq
done
for tag in s mix; do
  for rep in 1 2; do
    t0=$(date +%s%N)
    T1K_DEBUG_PHASES=1 $R/t1k_amd/bin/genotyper -f ref.fa -1 ${tag}_1.fq -2 ${tag}_2.fq -s 0.97 -o out_$tag 2> err_$tag.txt || { tail -5 err_$tag.txt; exit 1; }
    echo "$tag run $rep: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall"
    grep -E "windows,|device memory" err_$tag.txt | tail -3
    sleep 20  # (the driver zeroes the memory the process gave back: the next start would wait for it)
  done
done
md5sum out_s_genotype.tsv out_mix_genotype.tsv
