#!/bin/bash
# usage: shard_input_probe.sh <pairs> -- the executable with 1 rank, and with 2 / 3 ranks on one GPU that each index their own reads
mkdir -p /tmp/t1k_bench /root/repo/gpurun_out
cd /root/repo
n=$1
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $n, 24, 1.0, seed=2)"
cd /tmp/t1k_bench
A="-f hla_g24_s1.0.fa -1 reads_g24_s1.0_p${n}_seed2_1.fq -2 reads_g24_s1.0_p${n}_seed2_2.fq -s 0.97"
( time /root/repo/t1k_amd/bin/genotyper $A -o one 2> /root/repo/gpurun_out/shard_one.err ) 2>&1 | grep real
for g in 0,0 0,0,0; do
  ( time env T1K_GPUS=$g T1K_SHARD_INPUT=1 T1K_DEBUG_PHASES=1 /root/repo/t1k_amd/bin/genotyper $A -o own 2> /root/repo/gpurun_out/shard_own_$g.err ) 2>&1 | grep real
  grep "mapped + indexed" /root/repo/gpurun_out/shard_own_$g.err
  for s in genotype.tsv allele.tsv aligned_1.fa aligned_2.fa; do cmp one_$s own_$s && echo "$g $s identical"; done
done
