#!/bin/bash
# every phase line (T1K_DEBUG_PHASES) of the genotyper executable on the bench input; $1 = pairs (default 10 M), run on the GPU box
W=/tmp/t1k_bench; P=${1:-10000000}
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || exit 1
for i in 1 2; do
  t0=$(date +%s%N)
  T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/exe_ph 2> $W/ph_$i.txt
  echo "run $i: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall, md5 $(md5sum < $W/exe_ph_genotype.tsv | cut -c1-8)"
  sleep 20
done
grep -v "^\[t1k\] \(fullalign\|range\)" $W/ph_2.txt | cut -c1-300
