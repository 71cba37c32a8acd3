#!/bin/bash
# usage: kexp.sh <pattern> -- 1M pairs, one pipeline, under rocprofv3: kernels matching the pattern + total, and the result's md5
cd /root/repo; tools/scratch/prof.sh 1000000 1 kexp > /dev/null 2>&1
python3 - "$1" <<'PY'
import csv, sys, re
rows=list(csv.DictReader(open("gpurun_out/kexp_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if re.search(sys.argv[1], r["Name"]): print("%-56s calls %3s avg %.3f ms" % (r["Name"][:56], r["Calls"], float(r["AverageNs"])/1e6))
print("total kernel ms %.1f"%(tot/1e6))
PY
md5sum /tmp/t1k_bench/prof_out_genotype.tsv
