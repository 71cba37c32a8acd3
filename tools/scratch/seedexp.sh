#!/bin/bash
# $1 = label ; runs the 1M-pair executable under rocprofv3 with one pipeline and prints k_seed_groups / total
tools/scratch/prof.sh 1000000 1 $1 > /dev/null 2>&1
python3 - $1 <<'PY'
import csv,sys
rows=list(csv.DictReader(open("gpurun_out/%s_kernel_stats.csv"%sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if "k_seed_groups" in r["Name"] or "k_select<8192" in r["Name"] or "k_extract" in r["Name"]:
        print(sys.argv[1], r["Name"][:40], "calls", r["Calls"], "avg %.3f ms"%(float(r["AverageNs"])/1e6))
print(sys.argv[1], "total kernel ms %.1f"%(tot/1e6))
PY
md5sum /tmp/t1k_bench/prof_out_genotype.tsv
