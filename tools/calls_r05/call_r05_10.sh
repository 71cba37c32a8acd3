mkdir -p gpurun_out
W=/tmp/t1k_bench; P=10000000
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)"
T1K_DEBUG_PHASES=1 T1K_DEBUG_MAPS=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p${P}_seed2_1.fq -2 $W/reads_g24_s1.0_p${P}_seed2_2.fq -s 0.97 -o $W/exe_ph 2> $W/ph.txt
grep -E "address space|rss |Threads|Vm" $W/ph.txt | cut -c1-260 > gpurun_out/r05_c10_maps.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -15 > gpurun_out/r05_c10_pytest.log
