mkdir -p gpurun_out /tmp/t1k_bench
export TMPDIR=/tmp
R=$(pwd); LOG=gpurun_out/r05_c41.log; : > $LOG
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
( cd /tmp && T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c41 -o bench -- python $R/bench.py --pairs 10000000 --steps 1 --warmup 1 --no-cpu-baseline --no-executable-check > $R/gpurun_out/r05_c41_bench_1p.json 2> /dev/null )
cp "$(find /tmp/prof_c41 -name '*kernel_stats.csv' | head -1)" gpurun_out/r05_c41_kernel_stats_10M_1pipeline.csv
head -14 gpurun_out/r05_c41_kernel_stats_10M_1pipeline.csv | cut -c1-150 >> $LOG
python - >> $LOG <<'PY'
import json
d = json.load(open("gpurun_out/r05_c41_bench_1p.json"))
print("1 pipeline:", d["ms_per_step"], d["config"]["reference_output_check"])
PY
python bench.py --pairs 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-executable-check 2> /dev/null | cut -c1-220 >> $LOG
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "oracle or fuzz or golden" 2>&1 | grep "passed\|failed\|error" | tail -3 >> $LOG
