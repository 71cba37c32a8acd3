#!/bin/bash
# Round 6, fourth GPU call: the whole -m gpu suite on the round's code; closed-form pass without state writes for candidate-less groups (kernel time,
# traffic, bench step); configs[4] at size: bench line + analyzer by phase; the analyzer's variant pass at 100 k pairs beside the reference's.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callD.log; : > $L
say() { echo "$@" | tee -a $L; }
R=$(pwd)
say "== pytest -m gpu (whole suite)"
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r06_pytest_gpu.log 2>&1
tail -22 gpurun_out/r06_pytest_gpu.log | tee -a $L
say "== kernels alone (1 M pairs, one pipeline)"
tools/kstats_r06.sh "main main" 1 "k_chain_fast<5, [01]|k_collect|k_seed_groups" 2>&1 | tee -a $L
say "== fabric traffic (FETCH_SIZE / WRITE_SIZE, KB per 1 M-pair step, one pipeline)"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( cd /tmp && T1K_PIPELINES=1 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check --no-roofline-step > /dev/null 2>&1 )
  python - "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" $c <<'PY' | tee -a $L
import csv, sys, collections
tot = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == sys.argv[2]: tot[r["Kernel_Name"][:40]] += float(r["Counter_Value"])
print("%s: " % sys.argv[2] + "; ".join("%s %.3g" % (k, v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:9]) + "; ALL %.4g" % sum(tot.values()))
PY
done
say "== bench step, 10 M pairs, three pipelines"
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
tools/ab_r06.sh "||" 10000000 2 2>&1 | tee -a $L
say "== configs[4] at size: bench line (1 timed step, reference hashes checked) and the analyzer by phase"
python bench.py --pairs 10000000 --barcodes 100000 --steps 1 --warmup 1 --no-cpu-baseline --cold-runs 2 > gpurun_out/r06_bench_barcodes_10M_100k.json 2> /tmp/bb.err; echo "bench --barcodes rc $?" | tee -a $L
python - <<'PY' | tee -a $L
import json
d = json.loads(open("gpurun_out/r06_bench_barcodes_10M_100k.json").read().strip().splitlines()[-1])
print("   value %.0f pairs/s, %.0f ms per step, cold %s, peak device %.1f GB, reference_md5_ok %s, check %s" % (d["value"], d["ms_per_step"], d["config"].get("executable_cold_run", {}).get("wall_s_all_runs"), d.get("peak_device_gb", -1), d["reference_md5_ok"], d["config"].get("reference_output_check")))
PY
W=/tmp/t1k_bench; REF=$W/hla_g24_s1.0.fa; X=$W/reads_g24_s1.0_p10000000_seed2_bc100000
t1k_amd/bin/genotyper -f $REF -1 ${X}_1.fq -2 ${X}_2.fq --barcode ${X}_bc.fa -s 0.97 -o $W/bo 2> /dev/null
( time T1K_DEBUG_PHASES=1 t1k_amd/bin/analyzer -f $REF -a $W/bo_allele.tsv -1 $W/bo_aligned_1.fa -2 $W/bo_aligned_2.fa --barcode $W/bo_aligned_bc.fa -s 0.97 -o $W/bao ) 2>&1 | grep -E "variant pass|t1k variants|real|windows," | cut -c1-700 | tee -a $L
say "   analyzer files: _barcode_expr.tsv md5 $(md5sum < $W/bao_barcode_expr.tsv | cut -c1-32), _allele.vcf $(stat -c %s $W/bao_allele.vcf) bytes (committed: $(python -c "import json; d=json.load(open('tests/golden/full_size_md5.json'))['barcode_10M_100k']; print(d['analyzer_barcode_expr.tsv'], d['analyzer_vcf_bytes'])"))"
rm -f $W/bo_aligned* $W/bao_*
say "== analyzer variant pass beside the reference's, 100 k pairs with unknown SNPs"
tools/analyzer_variants_r05.sh 100000 2>&1 | tee -a $L
