#!/bin/bash
# the genotyper executable on the bench workload with T1K_DEBUG_PHASES (host phase lines): tools/phases_10M.sh [pairs] [extra env ...] -> gpurun_out/phases_<pairs>.log
W=/tmp/t1k_bench; P=${1:-10000000}; shift
mkdir -p gpurun_out $W
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || exit 1
for i in 1 2; do
  env "$@" T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/ph 2> $W/ph.err
  echo "== run $i ($*) rc $? genotype md5 $(md5sum < $W/ph_genotype.tsv | cut -c1-8)"
  grep "^\[t1k" $W/ph.err | grep -v "^\[t1k\] \(fullalign\|range\)" | cut -c1-400
done > gpurun_out/phases_$P.log
cat gpurun_out/phases_$P.log
