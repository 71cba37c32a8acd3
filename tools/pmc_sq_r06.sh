#!/bin/bash
# Round 6 (same passes as round 5): where the wave cycles of each kernel go (1 M pairs, ONE pipeline, one step; counters only -- no trace domains).
# Three passes (the SQ block has eight slots, TCC four): wave-cycle split, instruction mix, L2 hit rate.
# -> gpurun_out/r06_pmc_sq.csv: one row per kernel, counters summed over its launches; summary by tools/pmc_sq_r05.sh's python tail
mkdir -p gpurun_out /tmp/t1k_bench
export TMPDIR=/tmp
R=$(pwd)
PAIRS=${1:-1000000}
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', $PAIRS, 24, 1.0, seed=2)"
rocprofv3 -L 2> /dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Za-z_0-9]*\|GRBM_[A-Z_0-9]*\|TCP_[A-Za-z_0-9]*" | sort -u > gpurun_out/r06_pmc_counter_names.txt
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
i=0
for pass in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  ( cd /tmp && T1K_PIPELINES=1 timeout 600 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_sq_$i -o p -- python $R/bench.py --pairs $PAIRS --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check --no-roofline-step > /dev/null 2> /tmp/pmc_sq_$i.err )
  f="$(find /tmp/pmc_sq_$i -name '*counter_collection.csv' | head -1)"
  [ -z "$f" ] && { echo "pass $i: no counter file"; tail -5 /tmp/pmc_sq_$i.err; continue; }
  cp "$f" /tmp/pmc_sq_pass$i.csv
done
python - > gpurun_out/r06_pmc_sq.csv <<'PY'
import csv, collections, glob
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); names = []
for i, f in enumerate(sorted(glob.glob("/tmp/pmc_sq_pass*.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]; c = r["Counter_Name"]
        tot[k][c] += float(r["Counter_Value"])
        if c not in names: names.append(c)
        if i == 0 and c == "SQ_WAVES": calls[k] += 1
print("Kernel,Launches," + ",".join(names))
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
    print('"%s",%d,' % (k, calls[k]) + ",".join("%.6g" % tot[k].get(c, 0) for c in names))
PY
head -30 gpurun_out/r06_pmc_sq.csv | cut -c1-300
