#!/bin/bash
# A/B of the mapped output writer: bench step at 10 M pairs, and the aligned files of the executable at 1 M pairs in both modes
cd /root/repo
bash tools/scratch/bench_ab.sh "T1K_NO_MMAP_OUTPUT=1" "T1K_X=0" "T1K_NO_MMAP_OUTPUT=1" "T1K_X=0"
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
R=/tmp/t1k_bench/reads_g24_s1.0_p1000000_seed2
T1K_NO_MMAP_OUTPUT=1 t1k_amd/bin/genotyper -f /tmp/t1k_bench/hla_g24_s1.0.fa -1 ${R}_1.fq -2 ${R}_2.fq -s 0.97 -o /tmp/t1k_bench/mA > /dev/null 2>&1
t1k_amd/bin/genotyper -f /tmp/t1k_bench/hla_g24_s1.0.fa -1 ${R}_1.fq -2 ${R}_2.fq -s 0.97 -o /tmp/t1k_bench/mB > /dev/null 2>&1
T1K_NO_STREAM_OUTPUT=1 t1k_amd/bin/genotyper -f /tmp/t1k_bench/hla_g24_s1.0.fa -1 ${R}_1.fq -2 ${R}_2.fq -s 0.97 -o /tmp/t1k_bench/mC > /dev/null 2>&1
md5sum /tmp/t1k_bench/m[ABC]_aligned_1.fa /tmp/t1k_bench/m[ABC]_aligned_2.fa /tmp/t1k_bench/m[ABC]_genotype.tsv
df -T /tmp | tail -1
