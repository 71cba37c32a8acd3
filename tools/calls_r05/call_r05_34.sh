mkdir -p gpurun_out
W=/tmp/t1k_bench; P=10000000
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)"
for v in "T1K_FIRST_WINDOW=131072" ""; do
  env $v T1K_DEBUG_PHASES=1 T1K_DEBUG_TASKS=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/fw 2> $W/fw.err
  echo "== [$v]" >> gpurun_out/r05_c34_windows.log
  grep "prep window\|windows, \|again with capacities" $W/fw.err | cut -c1-220 >> gpurun_out/r05_c34_windows.log
  grep -c "assign" $W/fw.err >> gpurun_out/r05_c34_windows.log
done
