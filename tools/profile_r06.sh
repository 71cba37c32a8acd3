#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun; results land in gpurun_out/, the summaries are then copied to profiles/).
#   1. rocprofv3 --kernel-trace --stats of the bench command itself (10 M pairs, three pipelines), per-kernel summary
#   2. the same at 1 M pairs with ONE pipeline (kernels alone: no overlap inflation)
#   3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, counters only, as MI355X_MICROARCH.md prescribes) on THE BENCH'S OWN COMMAND:
#      10 M pairs, three pipelines, one step -> bytes per step per kernel family (FETCH x 2: the calibration of round 3 for 8- and 16-byte
#      loads, profiles/r03_fetch_calib.json) -> gpurun_out/r06_traffic.json, which bench.py reads for roofline.traffic
mkdir -p gpurun_out /tmp/t1k_bench
export TMPDIR=/tmp
PAIRS=${1:-10000000}
R=$(pwd)
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $PAIRS, 24, 1.0, seed=2); bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline --no-executable-check --no-roofline-step > $R/gpurun_out/r06_bench_under_profiler.json 2> /dev/null )
cp "$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)" gpurun_out/r06_kernel_stats.csv
( cd /tmp && T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_1m -o bench -- python $R/bench.py --pairs 1000000 --steps 1 --warmup 1 --no-cpu-baseline --no-executable-check --no-roofline-step > /dev/null 2>&1 )
cp "$(find /tmp/prof_1m -name '*kernel_stats.csv' | head -1)" gpurun_out/r06_kernel_stats_1M_1pipeline.csv
# 2b. ONE pipeline at the BENCH size (kernels alone at 10 M pairs: what roofline.frac_alone is priced with); passes = warm-up + steps profiled
( cd /tmp && T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_10m1 -o bench -- python $R/bench.py --pairs $PAIRS --steps 1 --warmup 1 --no-cpu-baseline --no-executable-check --no-roofline-step > $R/gpurun_out/r06_bench_10M_1pipeline.json 2> /dev/null )
cp "$(find /tmp/prof_10m1 -name '*kernel_stats.csv' | head -1)" gpurun_out/r06_kernel_stats_10M_1pipeline.csv
echo "{\"pairs\": $PAIRS, \"pipelines\": 1, \"passes\": 2, \"csv\": \"r06_kernel_stats_10M_1pipeline.csv\", \"command\": \"T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats -- python bench.py --pairs $PAIRS --steps 1 --warmup 1 (tools/profile_r05.sh)\"}" > gpurun_out/r06_kernel_stats_10M_1pipeline.json
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --pairs $PAIRS --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check --no-roofline-step > /dev/null 2>&1 )
  python - "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" $c > gpurun_out/r06_pmc_$c.csv <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != sys.argv[2]: continue
    tot[r["Kernel_Name"]] += float(r["Counter_Value"]); calls[r["Kernel_Name"]] += 1
print("Kernel,Calls,%s_sum,%s_per_call" % (sys.argv[2], sys.argv[2]))
for k in sorted(tot, key=lambda k: -tot[k]):
    print('"%s",%d,%.6g,%.6g' % (k[:110], calls[k], tot[k], tot[k] / calls[k]))
PY
done
python - $PAIRS > gpurun_out/r06_traffic.json <<'PY'
import csv, json, sys, re
sys.path.insert(0, ".")
import bench
fam = bench.FAMILIES
out = {n: 0.0 for n, _ in fam}; parts = {n: {"fetch_kb": 0.0, "write_kb": 0.0} for n, _ in fam}; other = 0.0
for c, key, mul in (("FETCH_SIZE", "fetch_kb", 2.0), ("WRITE_SIZE", "write_kb", 1.0)):
    for r in csv.DictReader(open("gpurun_out/r06_pmc_%s.csv" % c)):
        v = float(r["%s_sum" % c])
        for n, rx in fam:
            if re.search(rx, r["Kernel"]):
                out[n] += v * 1024 * mul; parts[n][key] += v; break
        else:
            other += v * 1024 * mul
print(json.dumps({"pairs": int(sys.argv[1]), "pipelines": 3, "bytes_per_step": out, "counter_kb_per_step": parts, "other_kernels_bytes_per_step": other,
                  "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --pairs %s --steps 1 --warmup 0` (three pipelines; tools/profile_r05.sh); "
                          "bytes = FETCH_SIZE x 2 (calibrated in round 3 for 8- and 16-byte loads: profiles/r03_fetch_calib.json) + WRITE_SIZE, KB = 1024 B, summed over "
                          "the step's launches of the family's kernels; k_dp_dense (chain and extension phases) is counted with the chain family" % sys.argv[1]}, indent=1))
PY
head -25 gpurun_out/r06_kernel_stats.csv | cut -c1-160
cat gpurun_out/r06_traffic.json
tail -c 2500 gpurun_out/r06_bench_under_profiler.json
