#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun; results land in gpurun_out/, the summaries are then copied to profiles/).
#   1. rocprofv3 --kernel-trace --stats of the bench command itself (default workload: 10 M pairs), per-kernel summary
#   2. the same at 1 M pairs with ONE pipeline (kernels alone: no overlap inflation)
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domain, as MI355X_MICROARCH.md prescribes) on the 1 M-pair workload with one pipeline
#   4. FETCH_SIZE calibration: a kernel that streams a known byte count with 8-byte and with 16-byte loads per lane
mkdir -p gpurun_out /tmp/t1k_bench
export TMPDIR=/tmp
PAIRS=${1:-10000000}
R=$(pwd)
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $PAIRS, 24, 1.0, seed=2); bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python $R/bench.py --pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline --no-executable-check > $R/gpurun_out/r03_bench_under_profiler.json 2> $R/gpurun_out/r03_bench_under_profiler.err )
python tools/rocpd_stats.py $(ls /tmp/prof_bench/*.db | head -1) > gpurun_out/r03_kernel_stats.csv
( cd /tmp && T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_1m -o bench -- python $R/bench.py --pairs 1000000 --steps 1 --warmup 1 --no-cpu-baseline --no-executable-check > $R/gpurun_out/r03_bench_1M_1pipeline.json 2>/dev/null )
python tools/rocpd_stats.py $(ls /tmp/prof_1m/*.db | head -1) > gpurun_out/r03_kernel_stats_1M_1pipeline.csv
summarise() {  # counter csv, counter name
python - "$1" $2 <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]: continue
    tot[r["Kernel_Name"]] += float(r["Counter_Value"]); calls[r["Kernel_Name"]] += 1
print("Kernel,Calls,%s_sum,%s_per_call" % (sys.argv[2], sys.argv[2]))
for k in sorted(tot, key=lambda k: -tot[k])[:40]:
    print('"%s",%d,%.6g,%.6g' % (k[:100], calls[k], tot[k], tot[k] / calls[k]))
PY
}
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && T1K_PIPELINES=1 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check > /dev/null 2>&1 )
  summarise "$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)" $c > gpurun_out/r03_pmc_$c.csv
  ( cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/cal_$c -o p -- $R/tools/fetch_calib 4096 > $R/gpurun_out/r03_fetch_calib_$c.json 2>/dev/null )
  summarise "$(find /tmp/cal_$c -name "*counter_collection.csv" | head -1)" $c > gpurun_out/r03_calib_$c.csv
done
head -6 gpurun_out/r03_pmc_FETCH_SIZE.csv gpurun_out/r03_pmc_WRITE_SIZE.csv gpurun_out/r03_calib_FETCH_SIZE.csv gpurun_out/r03_calib_WRITE_SIZE.csv
cat gpurun_out/r03_fetch_calib_FETCH_SIZE.json
head -30 gpurun_out/r03_kernel_stats.csv
tail -c 3000 gpurun_out/r03_bench_under_profiler.json
