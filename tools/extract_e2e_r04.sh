#!/bin/bash
# Round 4: fastq-extractor end to end on 4 M page-cached plain 2x150 bp pairs (97 % background) against the HLA-like reference: the mapped
# input path, the streaming loop (T1K_EXTRACT_STREAM=1) and the reference's own binary at -t 32; outputs compared.  Run on the GPU box.
W=/tmp/t1k_bench; P=${1:-4000000}; mkdir -p $W gpurun_out
python -c "import bench; bench.ensure_inputs('$W', 1000, 24, 1.0, seed=2)" || exit 1
REF=$W/hla_g24_s1.0.fa
[ -f $W/x_${P}_2.fq ] || tools/t1k_synth reads --ref $REF --out $W/x_$P --seed 77 --pairs $P --len 150 --bg 0.97
cat $W/x_${P}_1.fq $W/x_${P}_2.fq > /dev/null
LOG=gpurun_out/r04_extract_e2e.log; : > $LOG
now() { date +%s%N; }
for mode in mapped mapped stream; do
  E=""; [ $mode = stream ] && E="T1K_EXTRACT_STREAM=1"
  t0=$(now); env $E T1K_DEBUG_PHASES=1 t1k_amd/bin/fastq-extractor -f $REF -1 $W/x_${P}_1.fq -2 $W/x_${P}_2.fq -t 8 -o $W/xo_$mode 2> $W/xo_$mode.err; rc=$?
  ms=$(( ($(now) - t0) / 1000000 ))
  echo "$mode: rc $rc, $ms ms wall = $(( P * 1000 / ms )) pairs/s, kept $(grep -c '^@' $W/xo_${mode}_1.fq), md5 $(md5sum < $W/xo_${mode}_1.fq | cut -c1-8) $(md5sum < $W/xo_${mode}_2.fq | cut -c1-8)" >> $LOG
  grep "^\[t1k\] extractor" $W/xo_$mode.err >> $LOG
done
if [ -x oracle/_ref/fastq-extractor ]; then
  t0=$(now); oracle/_ref/fastq-extractor -f $REF -1 $W/x_${P}_1.fq -2 $W/x_${P}_2.fq -t 32 -o $W/xo_ref 2> /dev/null
  ms=$(( ($(now) - t0) / 1000000 ))
  echo "reference fastq-extractor -t 32: $ms ms wall = $(( P * 1000 / ms )) pairs/s, md5 $(md5sum < $W/xo_ref_1.fq | cut -c1-8) $(md5sum < $W/xo_ref_2.fq | cut -c1-8)" >> $LOG
fi
cat $LOG
