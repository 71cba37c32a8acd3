#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callV.log; : > $L
echo "== k_collect alone (1 M pairs, one pipeline): read-ends by index (T1K_COLLECT_ORDER=0) against heaviest first" | tee -a $L
for i in 1 2; do
  T1K_COLLECT_ORDER=0 tools/kstats_r06.sh "main" 1 "k_collect|k_csort_(count|scatter)<CostKey" 2>&1 | sed 's/^main /index /' | tee -a $L
  tools/kstats_r06.sh "main" 1 "k_collect|k_csort_(count|scatter)<CostKey" 2>&1 | sed 's/^main /heavy1st /' | tee -a $L
done
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
echo "== bench step, 10 M pairs, three pipelines" | tee -a $L
tools/ab_r06.sh "T1K_COLLECT_ORDER=0||T1K_COLLECT_ORDER=0||T1K_COLLECT_ORDER=0|" 10000000 3 2>&1 | cut -c1-140 | tee -a $L
