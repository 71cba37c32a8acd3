#!/bin/bash
# Round 6, second GPU call: k_pair with the next fragment's descriptor prefetched and the kept fragments in LDS -- parity subset, per-kernel A/B
# (one pipeline, 1 M pairs), the bench step A/B at 10 M pairs; the cold-process experiment with fewer hardware queues; two in-process ranks
# with the all-reduce EM collective at 10 M pairs; the contexts' memory by block name; the analyzer's variant pass by phase at 1 M pairs.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callB.log; : > $L
say() { echo "$@" | tee -a $L; }
say "== parity subset on the new library"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "golden_reference_outputs or live_reference_binary or adversarial or many_small_windows or baseline_configs or batching_and_rerun or analyzer_live or degenerate" 2>&1 | tail -5 | tee -a $L
say "== k_pair alone (1 M pairs, one pipeline): main = prefetch + LDS kept fragments; pairnopf = LDS kept fragments only; pairpf1 = main with one record per lane and round; pairold = round 5"
tools/kstats_r06.sh "main pairold pairnopf pairpf1 main pairold" 1 "k_pair" 2>&1 | tee -a $L
say "== bench step, 10 M pairs, three pipelines"
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
R=$(pwd)
tools/ab_r06.sh "T1K_GPU_LIB=$R/t1k_amd/lib/variants/libt1k_pairold.so||T1K_GPU_LIB=$R/t1k_amd/lib/variants/libt1k_pairold.so||T1K_GPU_LIB=$R/t1k_amd/lib/variants/libt1k_pairnopf.so" 10000000 2 2>&1 | tee -a $L
say "== memory by block (10 M pairs, executable)"
W=/tmp/t1k_bench; REF=$W/hla_g24_s1.0.fa; X=$W/reads_g24_s1.0_p10000000_seed2
T1K_DEBUG_MEM=1 T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $REF -1 ${X}_1.fq -2 ${X}_2.fq -s 0.97 -o $W/ex 2>&1 | grep -E "t1k mem|device memory|kept" | cut -c1-900 | tee -a $L
say "== cold process under fewer hardware queues; two in-process ranks with the all-reduce EM"
tools/while_waiting_r06.sh 2>&1 | tee -a $L
say "== analyzer variant pass by phase, 1 M pairs"
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
X1=$W/reads_g24_s1.0_p1000000_seed2
t1k_amd/bin/genotyper -f $REF -1 ${X1}_1.fq -2 ${X1}_2.fq -s 0.97 -o $W/g1 2> /dev/null
( time T1K_DEBUG_PHASES=1 t1k_amd/bin/analyzer -f $REF -a $W/g1_allele.tsv -1 $W/g1_aligned_1.fa -2 $W/g1_aligned_2.fa -s 0.97 -o $W/a1 ) 2>&1 | grep -E "variant pass|real|windows" | cut -c1-600 | tee -a $L
