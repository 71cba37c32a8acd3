#!/bin/bash
# Round 5, SURVEY 8f row 3: the genotyper executable on gzip-compressed read files (what sequencers deliver) next to `zcat` of the same
# files and next to the plain files.  10 M pairs by default; run on the GPU box.   tools/gz_r04.sh [pairs]  -> gpurun_out/r05_gz.log
W=/tmp/t1k_bench; P=${1:-10000000}
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || exit 1
F1=$W/reads_g24_s1.0_p${P}_seed2_1.fq; F2=$W/reads_g24_s1.0_p${P}_seed2_2.fq
LOG=gpurun_out/r05_gz.log; mkdir -p gpurun_out; : > $LOG
now() { date +%s%N; }
t0=$(now); ( gzip -6 -c $F1 > $F1.gz & gzip -6 -c $F2 > $F2.gz & wait ); echo "gzip -6 of both mates side by side: $(( ($(now) - t0) / 1000000 )) ms; $(stat -c %s $F1) -> $(stat -c %s $F1.gz) bytes per mate" >> $LOG
for i in 1 2; do t0=$(now); ( zcat $F1.gz > /dev/null & zcat $F2.gz > /dev/null & wait ); echo "zcat of both mates side by side -> /dev/null: $(( ($(now) - t0) / 1000000 )) ms" >> $LOG; done
run() {  # label, files, env...
  local label=$1 a=$2 b=$3; shift 3
  for i in 1 2; do
    t0=$(now)
    env "$@" T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $a -2 $b -s 0.97 -o $W/gz_$label 2> $W/gz_$label.err
    echo "$label run $i: rc $?, $(( ($(now) - t0) / 1000000 )) ms wall, genotype md5 $(md5sum < $W/gz_${label}_genotype.tsv | cut -c1-8) aligned_1 md5 $(md5sum < $W/gz_${label}_aligned_1.fa | cut -c1-8); $(grep 'read files mapped' $W/gz_$label.err | cut -c1-120) $(grep "main:" $W/gz_$label.err | cut -c11-80) $(grep "address space" $W/gz_$label.err | cut -c11-300)" >> $LOG
    sleep 15
  done
}
run plain $F1 $F2
run gz_streamed $F1.gz $F2.gz
run gz_whole_libdeflate $F1.gz $F2.gz T1K_STREAM_GZ=0
grep -h "gzip read files streamed\|windows, " $W/gz_gz_streamed.err | cut -c1-260 >> $LOG
cat $LOG
