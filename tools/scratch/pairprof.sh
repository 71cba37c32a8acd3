#!/bin/bash
# phases of k_pair (s_memtime ticks of thread 0 per workgroup): rebuild with -DT1K_PAIR_PROFILE, run 1M pairs with one pipeline
cd /root/repo/t1k_amd/csrc && touch t1k_pair.hip && make -j8 all EXTRA=-DT1K_PAIR_PROFILE > /dev/null 2>&1
mkdir -p /tmp/t1k_bench /root/repo/gpurun_out
cd /root/repo && python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
cd /tmp/t1k_bench
T1K_PIPELINES=1 /root/repo/t1k_amd/bin/genotyper -f hla_g24_s1.0.fa -1 reads_g24_s1.0_p1000000_seed2_1.fq -2 reads_g24_s1.0_p1000000_seed2_2.fq -s 0.97 -o sp 2>&1 | grep "pair phases" | tail -2
