# k_pair with PU records in flight per lane: PU = 3 (as built) against PU = 2 (rebuilt on the box), one pipeline, 10 M pairs, kernel stats
mkdir -p gpurun_out /tmp/t1k_bench
export TMPDIR=/tmp
R=$(pwd); LOG=gpurun_out/r05_c42.log; : > $LOG
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 10000000, 24, 1.0, seed=2)"
for pu in 3 2; do
  if [ $pu != 3 ]; then ( cd t1k_amd/csrc && touch t1k_pair.hip && make -j8 EXTRA=-DT1K_PAIR_UNROLL=$pu > /tmp/make_$pu.log 2>&1 ) || { echo "rebuild failed" >> $LOG; tail -5 /tmp/make_$pu.log >> $LOG; exit 1; }; fi
  rm -rf /tmp/prof_c42
  ( cd /tmp && T1K_PIPELINES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c42 -o bench -- python $R/bench.py --pairs 10000000 --steps 1 --warmup 1 --no-cpu-baseline --no-executable-check > $R/gpurun_out/r05_c42_bench_pu$pu.json 2> /dev/null )
  cp "$(find /tmp/prof_c42 -name '*kernel_stats.csv' | head -1)" gpurun_out/r05_c42_kstats_pu$pu.csv
  echo "== PU=$pu" >> $LOG
  grep "k_pair\|k_collect" gpurun_out/r05_c42_kstats_pu$pu.csv | cut -c1-120 >> $LOG
  python - $pu >> $LOG <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05_c42_bench_pu%s.json" % sys.argv[1]))
c = d["config"]["reference_output_check"]
print("1 pipeline:", round(d["ms_per_step"], 1), "ms; md5 identical:", c["genotype_and_allele_tsv_identical_every_step"], c["aligned_1_fa_identical"], c["aligned_2_fa_identical"])
PY
  python bench.py --pairs 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-executable-check 2> /dev/null | cut -c1-200 >> $LOG
done
