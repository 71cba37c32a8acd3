#!/bin/bash
# usage: exe_probe.sh "<pair counts>"  -- cold runs of the genotyper executable on the bench inputs, phases to gpurun_out/
mkdir -p /tmp/t1k_bench /root/repo/gpurun_out
cd /root/repo
for n in $1; do python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $n, 24, 1.0, seed=2)"; done
cd /tmp/t1k_bench
for n in $1; do
  ( time env T1K_DEBUG_PHASES=1 T1K_DEBUG_TASKS=1 $2 /root/repo/t1k_amd/bin/genotyper -f hla_g24_s1.0.fa -1 reads_g24_s1.0_p${n}_seed2_1.fq -2 reads_g24_s1.0_p${n}_seed2_2.fq -s 0.97 -o exe$n 2> /root/repo/gpurun_out/exe_$n.err ) 2>&1 | grep real
  grep "t1k job\|again with" /root/repo/gpurun_out/exe_$n.err | tail -12
  md5sum exe${n}_genotype.tsv
done
