mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_c3_pytest.log
timeout 1500 bash tools/profile_r05.sh > gpurun_out/r05_c3_profile.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
