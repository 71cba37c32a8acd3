W=/tmp/t1k_bench; P=10000000
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)" || exit 1
for i in 1 2 3; do t0=$(date +%s%N); t1k_amd/bin/genotyper 2>/dev/null; echo "usage run: $(( ($(date +%s%N) - t0) / 1000000 )) ms"; done
for v in 0 1 0 1; do
  t0=$(date +%s%N)
  env $( [ $v = 1 ] && echo T1K_FAST_EXIT=1 ) T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/exe_ph 2> $W/ph.txt
  echo "fast_exit=$v: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall; $(grep main: $W/ph.txt | cut -c1-120)"
  sleep 20
done
