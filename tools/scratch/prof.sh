#!/bin/bash
# rocprofv3 kernel stats of one cold run of the executable; $1 = pairs, $2 = pipelines, $3 = output name
mkdir -p /tmp/t1k_bench /root/repo/gpurun_out
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $1, 24, 1.0, seed=2)"
cd /tmp && export TMPDIR=/tmp
T1K_PIPELINES=$2 rocprofv3 --kernel-trace --stats -d /tmp/prof_$3 -o $3 -- /root/repo/t1k_amd/bin/genotyper -f /tmp/t1k_bench/hla_g24_s1.0.fa -1 /tmp/t1k_bench/reads_g24_s1.0_p$1_seed2_1.fq -2 /tmp/t1k_bench/reads_g24_s1.0_p$1_seed2_2.fq -s 0.97 -o /tmp/t1k_bench/prof_out > /dev/null 2>&1
python /root/repo/tools/rocpd_stats.py $(ls /tmp/prof_$3/*.db | head -1) > /root/repo/gpurun_out/$3_kernel_stats.csv
ls /tmp/prof_$3/* | head
