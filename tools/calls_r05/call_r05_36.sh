mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -12 > gpurun_out/r05_c36_pytest.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r05_c36_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r05_c36_bench.json 2> gpurun_out/r05_c36_bench.err
timeout 1200 bash tools/gz_r05.sh > gpurun_out/r05_c36_gz.txt 2>&1
