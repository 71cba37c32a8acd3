set -e
cd /root/repo
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/fp_bench.log 2>&1
cp /tmp/t1k_bench/last_genotype.tsv gpurun_out/fp_ours_genotype.tsv
REF=/tmp/t1k_bench/hla_g24_s1.0.fa
R1=/tmp/t1k_bench/reads_g24_s1.0_p1000000_seed2_1.fq
R2=/tmp/t1k_bench/reads_g24_s1.0_p1000000_seed2_2.fq
SECONDS=0; oracle/_ref/genotyper -f $REF -1 $R1 -2 $R2 -s 0.97 -t 64 -o /tmp/t1k_bench/ref1m > gpurun_out/fp_ref.log 2>&1 || true; echo "reference genotyper -t 64 on 1M pairs: $SECONDS s"
cp /tmp/t1k_bench/ref1m_genotype.tsv gpurun_out/fp_ref_genotype.tsv
cmp gpurun_out/fp_ours_genotype.tsv gpurun_out/fp_ref_genotype.tsv && echo "IDENTICAL genotype.tsv at 1M pairs" || (diff gpurun_out/fp_ours_genotype.tsv gpurun_out/fp_ref_genotype.tsv | head -10)
t1k_amd/bin/genotyper -f $REF -1 $R1 -2 $R2 -s 0.97 -o /tmp/t1k_bench/ours1m > gpurun_out/fp_ours_exe.log 2>&1
cmp /tmp/t1k_bench/ours1m_genotype.tsv /tmp/t1k_bench/ref1m_genotype.tsv && echo "executable genotype.tsv IDENTICAL"
cmp /tmp/t1k_bench/ours1m_allele.tsv /tmp/t1k_bench/ref1m_allele.tsv && echo "executable allele.tsv IDENTICAL"
echo "reference wall seconds: $SECONDS (includes ours)"
nproc
