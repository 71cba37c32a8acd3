#!/bin/bash
# Round 6: range size (T1K_BATCH, fragments: x 2 read-ends) and pipelines, three rounds (10 M pairs, 3 steps each)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
tools/ab_r06.sh "|T1K_BATCH=24576|T1K_BATCH=20480|T1K_PIPELINES=5|T1K_BATCH=24576 T1K_PIPELINES=4||T1K_BATCH=24576|T1K_BATCH=20480|T1K_PIPELINES=5|T1K_BATCH=24576 T1K_PIPELINES=4||T1K_BATCH=24576|T1K_BATCH=20480|T1K_PIPELINES=5|T1K_BATCH=24576 T1K_PIPELINES=4" 10000000 3 2>&1 | tee gpurun_out/r06_callO.log | cut -c1-150
