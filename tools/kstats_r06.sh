#!/bin/bash
# (round 6) per-kernel time of one bench step at ${PAIRS:-1000000} pairs for a list of library variants: tools/kstats_lib.sh "name1 name2 ..." [pipelines] [kernel regex]
# ("main" = the production library) -> one line per variant and kernel matching the regex, appended to gpurun_out/r06_kstats_lib.log
R=$(pwd); mkdir -p gpurun_out; PIPES=${2:-1}; RE=${3:-k_seed_groups}
export TMPDIR=/tmp
for v in $1; do
  LIB=$R/t1k_amd/lib/variants/libt1k_$v.so; [ "$v" = "main" ] && LIB=$R/t1k_amd/lib/libt1k_gpu.so
  rm -rf /tmp/ksl_$v
  ( cd /tmp && T1K_GPU_LIB=$LIB T1K_PIPELINES=$PIPES rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksl_$v -o p -- python $R/bench.py --pairs ${PAIRS:-1000000} --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check --no-roofline-step > /tmp/ksl_$v.json 2>/dev/null )
  f=$(find /tmp/ksl_$v -name "*kernel_stats.csv" | head -1)
  python - "$v" "$f" "$RE" /tmp/ksl_$v.json <<'PY' | tee -a $R/gpurun_out/r06_kstats_lib.log
import csv, sys, re, json
v, f, rx, j = sys.argv[1:5]
tot = 0.0; rows = []
for r in csv.DictReader(open(f)):
    tot += float(r["TotalDurationNs"])
    if re.search(rx, r["Name"]): rows.append(r)
try:
    d = json.load(open(j)); step = d["ms_per_step"]; ok = not d["config"].get("reference_output_check", {}).get("FAILED")
except Exception: step = -1; ok = False
import hashlib
print("%-10s sum of kernels %.1f ms, step %.0f ms %s genotype md5 %s" % (v, tot / 1e6, step, "" if ok else "CHECK FAILED", hashlib.md5(open("/tmp/t1k_bench/last_genotype.tsv","rb").read()).hexdigest()[:8]))
for r in rows: print("    %-70s calls %s avg %.3f ms total %.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
