#!/bin/bash
# Round 6: GPU-side experiments run beside the reference's 40-minute CPU run of tools/barcode_10M_r06.sh (the host is busy with 32 of its 256
# threads: timings here are A/B material, not headline numbers).
#   1. cold `genotyper` on the 10 M-pair headline input under fewer hardware queues (the experiment that was listed and not run)
#   2. two in-process ranks with the EM collective north_star names (T1K_EM_COLLECTIVE=allreduce) at 10 M pairs: calls / abundances against the exact run
cd "$(dirname "$0")/.."
W=/tmp/t1k_bench; P=10000000
tools/cold_r05.sh "|GPU_MAX_HW_QUEUES=4|GPU_MAX_HW_QUEUES=2|GPU_MAX_HW_QUEUES=1||GPU_MAX_HW_QUEUES=4|GPU_MAX_HW_QUEUES=2"
cp gpurun_out/r05_cold.log gpurun_out/r06_cold_queues.log; rm -f gpurun_out/r05_cold.log
cat gpurun_out/r06_cold_queues.log
REF=$W/hla_g24_s1.0.fa; X=$W/reads_g24_s1.0_p${P}_seed2
t1k_amd/bin/genotyper -f $REF -1 ${X}_1.fq -2 ${X}_2.fq -s 0.97 -o $W/ex 2> /dev/null
for mode in gather allreduce; do
  t0=$(date +%s%N)
  T1K_GPUS=0,0 T1K_SHARD_INPUT=1 T1K_EM_COLLECTIVE=$mode T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $REF -1 ${X}_1.fq -2 ${X}_2.fq -s 0.97 -o $W/ar_$mode 2> $W/ar_$mode.log
  echo "two in-process ranks, EM collective $mode: rc $?, $(( ($(date +%s%N) - t0) / 1000000 )) ms; $(grep -o 'in [0-9]* EM' $W/ar_$mode.log | head -1); genotype.tsv $(cmp -s $W/ex_genotype.tsv $W/ar_${mode}_genotype.tsv && echo IDENTICAL to the one-rank file || echo differs from the one-rank file)"
done
cp $W/ex_genotype.tsv gpurun_out/r06_hla_10M_genotype_exact.tsv; cp $W/ar_allreduce_genotype.tsv gpurun_out/r06_hla_10M_genotype_allreduce.tsv
cp $W/ex_allele.tsv gpurun_out/r06_hla_10M_allele_exact.tsv; cp $W/ar_allreduce_allele.tsv gpurun_out/r06_hla_10M_allele_allreduce.tsv
echo "md5 of the one-rank _genotype.tsv: $(md5sum < $W/ex_genotype.tsv | cut -c1-32) (committed reference hash: $(python -c "import json; print(json.load(open('tests/golden/full_size_md5.json'))['hla_10M']['_genotype.tsv'])"))"
diff $W/ex_genotype.tsv $W/ar_allreduce_genotype.tsv | head -20
rm -f $W/ar_*_aligned_* $W/ex_aligned_*
