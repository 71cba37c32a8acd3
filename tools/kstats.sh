#!/bin/bash
# per-kernel time of one bench step at 1 M pairs, one pipeline: tools/kstats.sh "ENV=..." -> gpurun_out/kstats_<tag>.csv (top 25 lines printed)
TAG=${2:-x}
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$TAG
env $1 T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -o p -- python $R/bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check > /dev/null 2>&1
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/kstats_$TAG.csv
head -22 "$f" | cut -c1-150
