#!/bin/bash
# Round 6: three, four or five pipelines -- the warm bench step (four rounds of three steps) and the cold executable (three runs each)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callT.log; : > $L
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
tools/ab_r06.sh "|T1K_PIPELINES=4|T1K_PIPELINES=5||T1K_PIPELINES=4|T1K_PIPELINES=5||T1K_PIPELINES=4|T1K_PIPELINES=5||T1K_PIPELINES=4|T1K_PIPELINES=5" 10000000 3 2>&1 | cut -c1-130 | tee -a $L
W=/tmp/t1k_bench; P=10000000
ms() { echo $(( ($(date +%s%N) - $1) / 1000000 )); }
for rep in 1 2 3; do for v in "" "T1K_PIPELINES=4" "T1K_PIPELINES=5"; do
  sleep 12; t0=$(date +%s%N)
  env $v T1K_DEBUG_MEM=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/exe_ph 2> $W/ph.txt
  echo "cold ${v:-default}: $(ms $t0) ms; md5 $(md5sum < $W/exe_ph_genotype.tsv | cut -c1-8); $(grep 'contexts .* GB; device' $W/ph.txt | cut -c1-120)" | tee -a $L
done; done
