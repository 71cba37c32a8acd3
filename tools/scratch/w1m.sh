#!/bin/bash
mkdir -p /tmp/t1k_bench gpurun_out
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
cd /tmp/t1k_bench
T1K_WINDOW=1048576 T1K_DEBUG_PHASES=1 T1K_DEBUG_ALLOC=1 T1K_DEBUG_TASKS=1 /root/repo/t1k_amd/bin/genotyper -f hla_g24_s1.0.fa -1 reads_g24_s1.0_p10000000_seed2_1.fq -2 reads_g24_s1.0_p10000000_seed2_2.fq -s 0.97 -o sw 2> /root/repo/gpurun_out/w1m.err
grep -c "t1k alloc" /root/repo/gpurun_out/w1m.err; grep "t1k alloc" /root/repo/gpurun_out/w1m.err | sort | uniq -c | sort -k1,1nr | head -20; grep "t1k job" /root/repo/gpurun_out/w1m.err | tail -4
