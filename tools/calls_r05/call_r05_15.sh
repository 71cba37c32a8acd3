mkdir -p gpurun_out
timeout 1500 bash tools/profile_r05.sh > gpurun_out/r05_c15_profile.txt 2>&1
timeout 600 python bench.py > gpurun_out/r05_c15_bench.json 2> gpurun_out/r05_c15_bench.err
