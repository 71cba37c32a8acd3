#!/bin/bash
# one PMC pass (SQ block) over a 1M-pair run of the executable with one pipeline; per-kernel sums -> gpurun_out/pmc_$1.csv
# usage: pmc_sq.sh <label> <counter> [<counter> ...]
label=$1; shift
mkdir -p /tmp/t1k_bench /root/repo/gpurun_out
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
cd /tmp && export TMPDIR=/tmp
T1K_PIPELINES=1 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$label -o p -- /root/repo/t1k_amd/bin/genotyper -f /tmp/t1k_bench/hla_g24_s1.0.fa -1 /tmp/t1k_bench/reads_g24_s1.0_p1000000_seed2_1.fq -2 /tmp/t1k_bench/reads_g24_s1.0_p1000000_seed2_2.fq -s 0.97 -o /tmp/t1k_bench/pmc_out > /dev/null 2>&1
f=$(find /tmp/pmc_$label -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' > /root/repo/gpurun_out/pmc_$label.csv
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]; c = r["Counter_Name"]
    if c not in names: names.append(c)
    tot[k][c] += float(r["Counter_Value"])
    if c == names[0]: calls[k] += 1
print("Kernel,Calls," + ",".join(names))
for k in sorted(tot, key=lambda k: -tot[k][names[0]])[:28]:
    print('"%s",%d,' % (k, calls[k]) + ",".join("%.4g" % tot[k][c] for c in names))
PY
