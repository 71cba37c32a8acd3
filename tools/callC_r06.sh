#!/bin/bash
# Round 6, third GPU call: the transposed copy of the reference under the closed-form pass (T1kRefDev::basesT) -- parity subset, kernel time and
# fabric traffic of k_chain_fast with and without it, the bench step; fragments in flight in k_pair; the analyzer's host fast path.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callC.log; : > $L
say() { echo "$@" | tee -a $L; }
R=$(pwd)
say "== parity subset on the new library (transposed reference windows, analyzer fast path)"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "golden_reference_outputs or live_reference_binary or adversarial or many_small_windows or baseline_configs or assign_stage or analyzer or long_reads_end_to_end or match_count" 2>&1 | tail -5 | tee -a $L
say "== k_chain_fast alone (1 M pairs, one pipeline): linear windows (T1K_REF_TRANSPOSE=0) against the transposed copy"
for i in 1 2; do
  T1K_REF_TRANSPOSE=0 tools/kstats_r06.sh "main" 1 "k_chain_fast<5, [01]|k_seed_groups" 2>&1 | sed 's/^main /linear /' | tee -a $L
  tools/kstats_r06.sh "main" 1 "k_chain_fast<5, [01]|k_seed_groups" 2>&1 | sed 's/^main /transp /' | tee -a $L
done
say "== fabric traffic of the chain kernels (FETCH_SIZE, KB per 1 M-pair step, one pipeline)"
for mode in 0 1; do
  rm -rf /tmp/pmc_t$mode
  ( cd /tmp && T1K_REF_TRANSPOSE=$mode T1K_PIPELINES=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_t$mode -o p -- python $R/bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check --no-roofline-step > /dev/null 2>&1 )
  python - "$(find /tmp/pmc_t$mode -name '*counter_collection.csv' | head -1)" $mode <<'PY' | tee -a $L
import csv, sys, collections
tot = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == "FETCH_SIZE": tot[r["Kernel_Name"][:40]] += float(r["Counter_Value"])
print("T1K_REF_TRANSPOSE=%s: " % sys.argv[2] + "; ".join("%s %.3g" % (k, v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]))
PY
done
say "== bench step, 10 M pairs, three pipelines"
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
tools/ab_r06.sh "T1K_REF_TRANSPOSE=0||T1K_REF_TRANSPOSE=0||T1K_PAIR_WGS=512|T1K_PAIR_WGS=768|T1K_PAIR_WGS=1536|T1K_PAIR_WGS=2048" 10000000 2 2>&1 | tee -a $L
say "== analyzer, 1 M pairs: host fast path against everything through the device"
W=/tmp/t1k_bench; REF=$W/hla_g24_s1.0.fa
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
X1=$W/reads_g24_s1.0_p1000000_seed2
t1k_amd/bin/genotyper -f $REF -1 ${X1}_1.fq -2 ${X1}_2.fq -s 0.97 -o $W/g1 2> /dev/null
for mode in "" "T1K_ANALYZER_NO_FAST=1"; do
  ( time env $mode T1K_DEBUG_PHASES=1 t1k_amd/bin/analyzer -f $REF -a $W/g1_allele.tsv -1 $W/g1_aligned_1.fa -2 $W/g1_aligned_2.fa -s 0.97 -o $W/a1_${mode:+nofast} ) 2>&1 | grep -E "variant pass|t1k variants|real" | cut -c1-700 | tee -a $L
done
say "   _allele.vcf $(cmp -s $W/a1__allele.vcf $W/a1_nofast_allele.vcf && echo identical || echo DIFFERENT) ($(stat -c %s $W/a1__allele.vcf) bytes)"
