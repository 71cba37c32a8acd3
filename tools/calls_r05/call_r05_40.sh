# 30 M pairs on one GPU after the memory query counts the pool's cached blocks as free: three cold runs (exit code, wall, device memory, kept windows)
W=/tmp/t1k_bench; mkdir -p $W gpurun_out
LOG=gpurun_out/r05_30M_after_pool_fix.log; : > $LOG
P=30000000
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || { echo "input generation failed" >> $LOG; exit 1; }
for run in 1 2 3; do
  t0=$(date +%s%N)
  T1K_DEBUG_PHASES=1 T1K_DEBUG_ARCHIVE=1 timeout 300 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/sz 2> $W/sz.err; rc=$?
  ms=$(( ($(date +%s%N) - t0) / 1000000 ))
  echo "== run $run, $P pairs: rc $rc, $ms ms wall, genotype md5 $(md5sum < $W/sz_genotype.tsv | cut -c1-8)" >> $LOG
  grep "windows,\|read sets of\|read sets are not kept\|device memory\|-> kept\|-> not kept\|^genotyper:" $W/sz.err | cut -c1-300 >> $LOG
  rm -f $W/sz_aligned_1.fa $W/sz_aligned_2.fa
  sleep 15
done
