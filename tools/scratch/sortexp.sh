#!/bin/bash
# upper bound of what a faster sort can give: k_select / k_truncate with the sort call compiled out (results are wrong)
cd /root/repo/t1k_amd/csrc && touch t1k_assign.hip && make -j8 all EXTRA="$1" > /dev/null 2>&1; cd /root/repo
tools/scratch/prof.sh 1000000 1 sortexp > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/sortexp_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if "k_select" in r["Name"] or "k_truncate" in r["Name"] or "k_collect" in r["Name"]:
        print(r["Name"][:44], "calls", r["Calls"], "avg %.3f ms"%(float(r["AverageNs"])/1e6))
print("total kernel ms %.1f"%(tot/1e6))
PY
