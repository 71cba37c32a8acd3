#!/bin/bash
# Round 6, sixth GPU call: what each of k_pair's sweeps over a fragment's two lists costs (timing-only ablations: results are WRONG by construction)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callF.log; : > $L
echo "== k_pair alone (1 M pairs, one pipeline): pairabl1 = no keep sweep, pairabl2 = list 1 not entered into the join table, pairabl3 = no sweep of the truncated-reference rule (genotype md5 differs: expected)" | tee -a $L
tools/kstats_r06.sh "main pairabl1 pairabl2 pairabl3 main pairabl1 pairabl2 pairabl3" 1 "k_pair|k_seed_groups" 2>&1 | tee -a $L
