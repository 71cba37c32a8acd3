#!/bin/bash
# Full-size parity of round 3 (run on the GPU box through gpurun): this build's executables against the reference binaries built by
# oracle/Makefile, every output file byte for byte, on the two BASELINE workloads round 2 never compared at size.
#   T. the reference genotyper's best thread count on the 100 k-pair sample bench.py's cpu_baseline uses (-t 32 / 64 / 128 / nproc)
#   B. BASELINE configs[4] on one GPU: 1 M 2x150 pairs with 100 k barcodes (log-uniform = Zipf-like usage), genotyper -> analyzer
#   H. the headline workload itself: 10 M 2x150 pairs vs the HLA-like rna reference, -s 0.97 (bench.py's input, seed 2)
# The md5 sums of the reference's files land in gpurun_out/r03_full_size_md5.json (committed as tests/golden/full_size_md5.json).
# PARTS=TBH (default) selects the parts.
cd "$(dirname "$0")/.."
PARTS=${PARTS:-TBH}
W=/tmp/t1k_fs; mkdir -p $W gpurun_out
L=gpurun_out/r03_full_size_parity.log; : > $L
J=gpurun_out/r03_full_size_md5.json
say() { echo "$@" | tee -a $L; }
REFG=oracle/_ref/genotyper; REFA=oracle/_ref/analyzer
OURG=t1k_amd/bin/genotyper; OURA=t1k_amd/bin/analyzer
# run a command, print wall seconds and the peak resident set of its process tree (GB)
timed() { python3 - "$@" <<'PY'
import resource, subprocess, sys, time
t = time.time(); rc = subprocess.call(sys.argv[1:], stdout=subprocess.DEVNULL)
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print("%.1f s, peak RSS %.1f GB, rc %d" % (time.time() - t, ru.ru_maxrss / 1048576.0, rc))
PY
}
cmpall() { # ours ref suffixes...
  local a=$1 b=$2; shift 2
  for s in "$@"; do
    if [ ! -f $b$s ]; then say "   $s: reference file missing"; elif cmp -s $a$s $b$s; then say "   $s IDENTICAL ($(stat -c %s $a$s) bytes, md5 $(md5sum < $b$s | cut -c1-32))"; else say "   $s DIFFERS"; fi
  done; }
say "box: $(nproc) hardware threads, $(free -g | awk '/Mem:/{print $2" GB RAM, "$7" GB available"}'), $(df -h /tmp | awk 'NR==2{print $4}') free in /tmp"
MEMKB=$(awk '/MemTotal/{print $2}' /proc/meminfo)
tools/t1k_synth ref-rna --genes 24 --scale 1.0 --seed 20250614 > $W/hla.fa
say "reference: $(grep -c '>' $W/hla.fa) alleles"

if [[ $PARTS == *T* ]]; then
  say "T. reference genotyper on the first 100 k pairs of the headline input (cpu_baseline's sample), by thread count"
  tools/t1k_synth reads --ref $W/hla.fa --pairs 100000 --len 150 --seed 2 --out $W/t
  BEST=1000000
  for t in 32 64 128 $(nproc); do
    R=$(timed $REFG -f $W/hla.fa -1 $W/t_1.fq -2 $W/t_2.fq -s 0.97 -t $t -o $W/tref 2>/dev/null); say "   -t $t: $R"
    S=$(echo $R | awk '{print int($1 * 10)}'); if [ $S -lt $BEST ]; then BEST=$S; BT=$t; fi
  done
  say "   fastest: -t $BT"
fi

if [[ $PARTS == *B* ]]; then
  say "B. 1 M pairs x 100 k barcodes (configs[4] on one GPU), -s 0.97, genotyper -> analyzer"
  tools/t1k_synth reads --ref $W/hla.fa --pairs 1000000 --len 150 --seed 5 --barcodes 100000 --sub 0.0 --out $W/b
  say "   distinct barcodes in the file: $(grep -v '>' $W/b_bc.fa | sort -u | wc -l)"
  say "   this build genotyper: $(timed $OURG -f $W/hla.fa -1 $W/b_1.fq -2 $W/b_2.fq --barcode $W/b_bc.fa -s 0.97 -o $W/bo 2>$W/bo.log)"
  R=$(timed $REFG -f $W/hla.fa -1 $W/b_1.fq -2 $W/b_2.fq --barcode $W/b_bc.fa -s 0.97 -t ${BT:-64} -o $W/br 2>$W/br.log); say "   reference genotyper -t ${BT:-64}: $R"
  RSS1M=$(echo $R | grep -o 'RSS [0-9.]*' | awk '{print $2}')
  cmpall $W/bo $W/br _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa _aligned_bc.fa
  say "   this build analyzer: $(timed $OURA -f $W/hla.fa -a $W/br_allele.tsv -1 $W/br_aligned_1.fa -2 $W/br_aligned_2.fa --barcode $W/br_aligned_bc.fa -s 0.97 -o $W/bao 2>$W/bao.log)"
  say "   reference analyzer -t 64: $(timed $REFA -f $W/hla.fa -a $W/br_allele.tsv -1 $W/br_aligned_1.fa -2 $W/br_aligned_2.fa --barcode $W/br_aligned_bc.fa -s 0.97 -t 64 -o $W/bar 2>$W/bar.log)"
  say "   reference VCF: $(stat -c %s $W/bar_allele.vcf) bytes (0 = no variant called: the precondition of the comparison)"
  cmpall $W/bao $W/bar _barcode_expr.tsv _allele.vcf
  say "   _barcode_expr.tsv: $(wc -l < $W/bar_barcode_expr.tsv) lines"
  cp $W/br_genotype.tsv gpurun_out/r03_barcode_1M_ref_genotype.tsv
  B_MD5="\"barcode_1M_100k\": {\"pairs\": 1000000, \"barcodes\": 100000, \"seed\": 5, \"sub\": 0.0, \"flags\": \"-s 0.97\", $(for s in _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa _aligned_bc.fa; do printf '"%s": "%s", ' $s $(md5sum < $W/br$s | cut -c1-32); done) \"analyzer_barcode_expr.tsv\": \"$(md5sum < $W/bar_barcode_expr.tsv | cut -c1-32)\", \"analyzer_vcf_bytes\": $(stat -c %s $W/bar_allele.vcf)}"
  echo "{$B_MD5}" > $J
  rm -f $W/bo_aligned* $W/b_1.fq $W/b_2.fq
fi

if [[ $PARTS == *H* ]]; then
  say "H. 10 M pairs, HLA-like rna, -s 0.97 (the bench.py input)"
  tools/t1k_synth reads --ref $W/hla.fa --pairs 10000000 --len 150 --seed 2 --out $W/h
  say "   this build: $(timed $OURG -f $W/hla.fa -1 $W/h_1.fq -2 $W/h_2.fq -s 0.97 -o $W/ho 2>$W/ho.log)"
  for s in _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa; do say "   ours $s md5 $(md5sum < $W/ho$s | cut -c1-32)"; done
  # the reference keeps every distinct read-end's overlap list in memory until pairing (Genotyper.cpp:455-560): ~8x the 1 M-pair footprint
  LIM=$((MEMKB / 100 * 85))
  say "   reference -t ${BT:-64} under ulimit -v ${LIM} kB (the 1 M-pair run above peaked at ${RSS1M:-?} GB)"
  if [ -n "$RSS1M" ] && awk -v r=$RSS1M -v m=$MEMKB 'BEGIN{exit !(r * 9 * 1048576 > m * 0.8)}'; then
    say "   SKIPPED: 9 x ${RSS1M} GB would not fit this box's memory"; exit 0
  fi
  say "   reference -t ${BT:-64}: $( ( ulimit -v $LIM; timed $REFG -f $W/hla.fa -1 $W/h_1.fq -2 $W/h_2.fq -s 0.97 -t ${BT:-64} -o $W/hr 2>$W/hr.log ) )"
  cmpall $W/ho $W/hr _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa
  say "   EM iterations: ours $(grep -o 'in [0-9]* EM' $W/ho.log) / reference $(grep -o 'in [0-9]* EM' $W/hr.log)"
  if [ -s $W/hr_genotype.tsv ]; then
    cp $W/hr_genotype.tsv gpurun_out/r03_hla_10M_ref_genotype.tsv
    cp $W/hr_allele.tsv gpurun_out/r03_hla_10M_ref_allele.tsv
    H_MD5="\"hla_10M\": {\"pairs\": 10000000, \"seed\": 2, \"flags\": \"-s 0.97\", $(for s in _genotype.tsv _allele.tsv _aligned_1.fa; do printf '"%s": "%s", ' $s $(md5sum < $W/hr$s | cut -c1-32); done) \"_aligned_2.fa\": \"$(md5sum < $W/hr_aligned_2.fa | cut -c1-32)\"}"
    echo "{${B_MD5:+$B_MD5, }$H_MD5}" > $J
  fi
fi
cat $J 2>/dev/null
