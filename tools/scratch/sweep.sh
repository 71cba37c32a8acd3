#!/bin/bash
mkdir -p /tmp/t1k_bench gpurun_out
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
cd /tmp/t1k_bench
run() { ( time env T1K_DEBUG_PHASES=1 "$@" /root/repo/t1k_amd/bin/genotyper -f hla_g24_s1.0.fa -1 reads_g24_s1.0_p10000000_seed2_1.fq -2 reads_g24_s1.0_p10000000_seed2_2.fq -s 0.97 -o sw 2> sw.err ) 2>&1 | grep real | tr '\n' ' '; grep -o "device loop [0-9.]* ms" sw.err | tr '\n' ' '; grep -o "allocated by the contexts in [0-9.]* ms" sw.err; }
echo "default"; run
echo "default again"; run
echo "pipes 3"; run T1K_PIPELINES=3
echo "pipes 4"; run T1K_PIPELINES=4
echo "window 4M"; run T1K_WINDOW=4194304
echo "window 1M"; run T1K_WINDOW=1048576
echo "batch 8192 frags"; run T1K_BATCH=8192
echo "pipes 3 + batch 8192"; run T1K_PIPELINES=3 T1K_BATCH=8192
