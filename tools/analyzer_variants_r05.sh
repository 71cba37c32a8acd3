#!/bin/bash
# What the analyzer's variant pass costs beside the reference's analyzer (round 5): N pairs drawn from a reference with unknown SNPs,
# this build's genotyper, then both analyzers in their default mode.  usage: tools/analyzer_variants_r05.sh [pairs] > log
set -e
N=${1:-100000}
D=$(mktemp -d)
python - "$D" "$N" <<'PY'
import sys
sys.path.insert(0, "tests")
import util
util.several_snps_sample(sys.argv[1], 3, genes=8, pairs=int(sys.argv[2]), scale=0.1)
PY
G=$D/g
t1k_amd/bin/genotyper -f $D/ref.fa -1 $D/r_1.fq -2 $D/r_2.fq --barcode $D/r_bc.fa -o $G 2>/dev/null
echo "pairs $N, selected alleles $(wc -l < ${G}_allele.tsv), aligned fragments $(grep -c '>' ${G}_aligned_1.fa)"
C="-f $D/ref.fa -a ${G}_allele.tsv -1 ${G}_aligned_1.fa -2 ${G}_aligned_2.fa --barcode ${G}_aligned_bc.fa"
for mode in "" "--varMaxGroup 0"; do
  echo "== this build, mode '$mode'"
  ( time T1K_DEBUG_PHASES=1 t1k_amd/bin/analyzer $C -o $D/gpu$( echo $mode | tr -d ' -' ) $mode 2>&1 | grep -E "variant pass|Post analysis" ) 2>&1 | grep -E "variant pass|real"
done
if [ -x oracle/_ref/analyzer ]; then
  echo "== reference analyzer -t 16"
  ( time oracle/_ref/analyzer $C -t 16 -o $D/ref 2>/dev/null ) 2>&1 | grep real
  cmp $D/ref_allele.vcf $D/gpu_allele.vcf && cmp $D/ref_barcode_expr.tsv $D/gpu_barcode_expr.tsv && echo "files identical: $(wc -l < $D/ref_allele.vcf) VCF lines"
fi
rm -rf $D
