#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun; results land in gpurun_out/, the summaries are then copied to profiles/).
#   1. rocprofv3 --kernel-trace --stats of the bench command itself (default workload: 10 M pairs), per-kernel summary
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domain, as MI355X_MICROARCH.md prescribes) on the 1 M-pair workload with one pipeline
mkdir -p gpurun_out /tmp/t1k_bench
export TMPDIR=/tmp
PAIRS=${1:-10000000}
cd /root/repo && python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $PAIRS, 24, 1.0, seed=2); bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python /root/repo/bench.py --pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r02_bench_under_profiler.json 2> /root/repo/gpurun_out/r02_bench_under_profiler.err )
python tools/rocpd_stats.py $(ls /tmp/prof_bench/*.db | head -1) > gpurun_out/r02_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && T1K_PIPELINES=1 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python /root/repo/bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 )
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY' > gpurun_out/r02_pmc_$c.csv
import csv, sys, collections
tot = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]: continue
    tot[r["Kernel_Name"]] += float(r["Counter_Value"]); calls[r["Kernel_Name"]] += 1
print("Kernel,Calls,%s_sum,%s_per_call" % (sys.argv[2], sys.argv[2]))
for k in sorted(tot, key=lambda k: -tot[k])[:40]:
    print('"%s",%d,%.6g,%.6g' % (k[:100], calls[k], tot[k], tot[k] / calls[k]))
PY
done
head -5 gpurun_out/r02_pmc_FETCH_SIZE.csv gpurun_out/r02_pmc_WRITE_SIZE.csv
tail -c 2500 gpurun_out/r02_bench_under_profiler.json
