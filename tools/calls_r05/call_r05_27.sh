mkdir -p gpurun_out
W=/tmp/t1k_bench; P=30000000; LOG=gpurun_out/r05_c28_30M.log; : > $LOG
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)"
for v in "T1K_DEBUG_ARCHIVE=1" "" "" ""; do
  t0=$(date +%s%N)
  env $v T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/sz 2> $W/sz.err; rc=$?
  ms=$(( ($(date +%s%N) - t0) / 1000000 ))
  echo "== [$v] $P pairs: rc $rc, $ms ms wall, genotype md5 $(md5sum < $W/sz_genotype.tsv | cut -c1-8)" >> $LOG
  grep "windows,\|read sets of\|read sets are not kept\|device memory\|row entries in the chunks" $W/sz.err | cut -c1-330 >> $LOG
  sleep 10
done
