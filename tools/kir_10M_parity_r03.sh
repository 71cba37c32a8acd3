#!/bin/bash
# BASELINE configs[2] at size, round 3 (run on the GPU box through gpurun): 10 M 2x150 bp pairs against the KIR-like dna reference with
# --preset kir-wgs = -s 0.9 --relaxIntronAlign (run-t1k:300-304) -- this build (coverage deferred to selection, eager, two ranks on one
# GPU) against the reference binary built by oracle/Makefile, every output file byte for byte.  Round 2 compared this workload before
# the coverage was deferred (profiles/r02_full_size_parity.log part B); with --relaxIntronAlign the near-best alignments still run in
# every range (the relaxed counts feed pairing), only their coverage updates moved behind selection.
# The reference's md5 sums land in gpurun_out/r03_kir_10M_md5.json (merged into tests/golden/full_size_md5.json as "kir_10M").
cd "$(dirname "$0")/.."
W=/tmp/t1k_kir; mkdir -p $W gpurun_out
L=gpurun_out/r03_kir_10M_parity.log; : > $L
say() { echo "$@" | tee -a $L; }
timed() { python3 - "$@" <<'PY'
import resource, subprocess, sys, time
t = time.time(); rc = subprocess.call(sys.argv[1:], stdout=subprocess.DEVNULL)
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print("%.1f s, peak RSS %.1f GB, rc %d" % (time.time() - t, ru.ru_maxrss / 1048576.0, rc))
PY
}
SUF="_genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa"
tools/t1k_synth ref-dna --genes 17 --scale 1.0 --seed 20250614 > $W/kir.fa
tools/t1k_synth reads --ref $W/kir.fa --pairs 10000000 --len 150 --seed 3 --out $W/k
say "K. 10 M pairs, KIR-like dna ($(grep -c '>' $W/kir.fa) alleles), -s 0.9 --relaxIntronAlign (kir-wgs)"
A="-f $W/kir.fa -1 $W/k_1.fq -2 $W/k_2.fq -s 0.9 --relaxIntronAlign"
say "   this build: $(T1K_DEBUG_PHASES=1 timed t1k_amd/bin/genotyper $A -o $W/o 2>$W/o.log)"
grep -h "coverage deferred\|coverage of the" $W/o.log | head -3 | tee -a $L
say "   this build, T1K_COVERAGE=eager: $(T1K_COVERAGE=eager timed t1k_amd/bin/genotyper $A -o $W/e 2>$W/e.log)"
say "   this build, two ranks on one GPU: $(T1K_GPUS=0,0 timed t1k_amd/bin/genotyper $A -o $W/t 2>$W/t.log)"
say "   reference -t 32: $(timed oracle/_ref/genotyper $A -t 32 -o $W/r 2>$W/r.log)"
for m in o e t; do
  for s in $SUF; do
    if cmp -s $W/$m$s $W/r$s; then say "   [$m] $s IDENTICAL ($(stat -c %s $W/r$s) bytes, md5 $(md5sum < $W/r$s | cut -c1-32))"; else say "   [$m] $s DIFFERS"; fi
  done
done
say "   EM iterations: ours $(grep -o 'in [0-9]* EM' $W/o.log | head -1) / reference $(grep -o 'in [0-9]* EM' $W/r.log | head -1)"
if [ -s $W/r_genotype.tsv ]; then
  cp $W/r_genotype.tsv gpurun_out/r03_kir_10M_ref_genotype.tsv
  echo "{\"kir_10M\": {\"pairs\": 10000000, \"seed\": 3, \"genes\": 17, \"flags\": \"-s 0.9 --relaxIntronAlign\", $(for s in _genotype.tsv _allele.tsv _aligned_1.fa; do printf '"%s": "%s", ' $s $(md5sum < $W/r$s | cut -c1-32); done) \"_aligned_2.fa\": \"$(md5sum < $W/r_aligned_2.fa | cut -c1-32)\"}}" > gpurun_out/r03_kir_10M_md5.json
  cat gpurun_out/r03_kir_10M_md5.json
fi
