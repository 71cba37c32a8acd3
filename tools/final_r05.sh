# Final validation of round 5 on the last code: -m gpu suite, smoke, kernel stats (one pipeline and three) at 10 M pairs, the bench line
mkdir -p gpurun_out /tmp/t1k_bench
export TMPDIR=/tmp
R=$(pwd); LOG=gpurun_out/r05_final.log; : > $LOG
timeout 1500 python -m pytest tests -m gpu -x -q > /tmp/pytest_gpu.out 2>&1; echo "pytest -m gpu: rc $?" >> $LOG
grep -E "passed|failed|error" /tmp/pytest_gpu.out | tail -3 >> $LOG
( grep -E "passed|failed|error" /tmp/pytest_gpu.out | tail -3; tail -25 /tmp/pytest_gpu.out | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" ) > gpurun_out/r05_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $LOG
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
( cd /tmp && T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f1 -o bench -- python $R/bench.py --pairs 10000000 --steps 1 --warmup 1 --no-cpu-baseline --no-executable-check > $R/gpurun_out/r05_bench_10M_1pipeline.json 2> /dev/null )
f="$(find /tmp/prof_f1 -name '*kernel_stats.csv' | head -1)"; [ -n "$f" ] && cp "$f" gpurun_out/r05_kernel_stats_10M_1pipeline.csv && cp "$f" profiles/r05_kernel_stats_10M_1pipeline.csv
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f3 -o bench -- python $R/bench.py --pairs 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-executable-check > $R/gpurun_out/r05_bench_under_profiler.json 2> /dev/null )
f="$(find /tmp/prof_f3 -name '*kernel_stats.csv' | head -1)"; [ -n "$f" ] && cp "$f" gpurun_out/r05_kernel_stats.csv
python bench.py --steps 3 --warmup 1 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
cut -c1-300 gpurun_out/r05_bench.json >> $LOG
head -12 gpurun_out/r05_kernel_stats_10M_1pipeline.csv | cut -c1-140 >> $LOG
