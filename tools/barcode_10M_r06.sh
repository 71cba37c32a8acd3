#!/bin/bash
# Round 6: BASELINE configs[4] AT ITS SIZE on one GPU -- 10 M 2x150 bp pairs carrying 100 k 10x-style barcodes, -s 0.97,
# genotyper -> analyzer (default --varMaxGroup 8) -- against the REFERENCE binaries (oracle/_ref, built by oracle/Makefile from /root/reference)
# run on the host cores of the same box.  The input is bench.py's own (`python bench.py --pairs 10000000 --barcodes 100000`: seed 2), so that the
# bench line of that command can check its files against the hashes this script produces (tests/golden/full_size_md5.json: barcode_10M_100k).
# The reference run (~40 min at -t 32, ~190 GB resident) goes on in the background; the GPU is used meanwhile by the parts under "while waiting".
#   -> gpurun_out/r06_barcode_10M.log, gpurun_out/r06_barcode_10M_md5.json, gpurun_out/r06_barcode_10M_ref_genotype.tsv / _allele.tsv
cd "$(dirname "$0")/.."
W=/tmp/t1k_bench; P=${P:-10000000}; B=${B:-100000}; mkdir -p $W gpurun_out
L=gpurun_out/r06_barcode_10M.log; : > $L
say() { echo "$@" | tee -a $L; }
REFG=oracle/_ref/genotyper; REFA=oracle/_ref/analyzer
OURG=t1k_amd/bin/genotyper; OURA=t1k_amd/bin/analyzer
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
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2, barcodes=$B)" || exit 1
REF=$W/hla_g24_s1.0.fa; X=$W/reads_g24_s1.0_p${P}_seed2_bc$B
say "input: $P pairs, $(grep -v '>' ${X}_bc.fa | sort -u | wc -l) distinct barcodes of $B"
( R=$(timed $REFG -f $REF -1 ${X}_1.fq -2 ${X}_2.fq --barcode ${X}_bc.fa -s 0.97 -t 32 -o $W/br 2>$W/br.log); echo "$R" > $W/br.time ) &
REFPID=$!

say "== while waiting: this build on the same input"
for i in 1 2; do say "   this build genotyper (cold process, run $i): $(timed $OURG -f $REF -1 ${X}_1.fq -2 ${X}_2.fq --barcode ${X}_bc.fa -s 0.97 -o $W/bo 2>$W/bo.log)"; done
say "   this build analyzer, default --varMaxGroup 8 (on this build's own genotyper output): $(timed timeout 1500 $OURA -f $REF -a $W/bo_allele.tsv -1 $W/bo_aligned_1.fa -2 $W/bo_aligned_2.fa --barcode $W/bo_aligned_bc.fa -s 0.97 -o $W/bao 2>$W/bao.log)"
tail -3 $W/bao.log | cut -c1-200 >> $L
say "   this build analyzer, --varMaxGroup 0: $(timed timeout 1500 $OURA -f $REF -a $W/bo_allele.tsv -1 $W/bo_aligned_1.fa -2 $W/bo_aligned_2.fa --barcode $W/bo_aligned_bc.fa -s 0.97 --varMaxGroup 0 -o $W/bao0 2>$W/bao0.log)"
if [ -n "$WHILE_WAITING" ]; then say "== while waiting: $WHILE_WAITING"; bash -c "$WHILE_WAITING" >> $L 2>&1; fi

wait $REFPID
say "== reference genotyper -t 32: $(cat $W/br.time)"
cmpall $W/bo $W/br _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa _aligned_bc.fa
say "   EM iterations: ours $(grep -o 'in [0-9]* EM' $W/bo.log) / reference $(grep -o 'in [0-9]* EM' $W/br.log)"
say "== reference analyzer -t 64 (on the reference genotyper's output): $(timed $REFA -f $REF -a $W/br_allele.tsv -1 $W/br_aligned_1.fa -2 $W/br_aligned_2.fa --barcode $W/br_aligned_bc.fa -s 0.97 -t 64 -o $W/bar 2>$W/bar.log)"
say "   reference VCF: $(stat -c %s $W/bar_allele.vcf) bytes, _barcode_expr.tsv: $(wc -l < $W/bar_barcode_expr.tsv) lines"
cmpall $W/bao $W/bar _barcode_expr.tsv _allele.vcf
say "   (--varMaxGroup 0 run of this build:)"; cmpall $W/bao0 $W/bar _barcode_expr.tsv
if [ -s $W/br_genotype.tsv ]; then
  cp $W/br_genotype.tsv gpurun_out/r06_barcode_10M_ref_genotype.tsv; cp $W/br_allele.tsv gpurun_out/r06_barcode_10M_ref_allele.tsv
  cp $W/bar_allele.vcf gpurun_out/r06_barcode_10M_ref_allele.vcf
  RT=$(cut -d' ' -f1 $W/br.time)
  echo "{\"barcode_10M_100k\": {\"pairs\": $P, \"barcodes\": $B, \"seed\": 2, \"flags\": \"-s 0.97\", $(for s in _genotype.tsv _allele.tsv _aligned_1.fa _aligned_2.fa _aligned_bc.fa; do printf '"%s": "%s", ' $s $(md5sum < $W/br$s | cut -c1-32); done) \"analyzer_barcode_expr.tsv\": \"$(md5sum < $W/bar_barcode_expr.tsv | cut -c1-32)\", \"analyzer_allele.vcf\": \"$(md5sum < $W/bar_allele.vcf | cut -c1-32)\", \"analyzer_vcf_bytes\": $(stat -c %s $W/bar_allele.vcf), \"reference_run\": {\"threads\": 32, \"wall_s\": $RT, \"log\": \"profiles/r06_barcode_10M.log\"}}}" > gpurun_out/r06_barcode_10M_md5.json
  cat gpurun_out/r06_barcode_10M_md5.json
fi
