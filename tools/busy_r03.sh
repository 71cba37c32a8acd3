#!/bin/bash
# how busy the GPU is over bench steps: rocprofv3 kernel trace of `bench.py --pairs N --steps 2 --warmup 1`, then tools/rocpd_stats.py --busy
# (union of the kernel intervals against the traced span, and the idle gaps of more than 1 ms with the kernels around them)
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
R=$(pwd); PAIRS=${1:-10000000}
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $PAIRS, 24, 1.0, seed=2)"
rm -rf /tmp/prof_busy
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_busy -o b -- python $R/bench.py --pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline --no-executable-check > /dev/null 2>&1 )
python tools/rocpd_stats.py $(ls /tmp/prof_busy/*.db | head -1) --busy 2> gpurun_out/r03_gpu_busy.txt > /dev/null
cat gpurun_out/r03_gpu_busy.txt
