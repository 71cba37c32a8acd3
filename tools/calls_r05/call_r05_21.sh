mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -12 > gpurun_out/r05_c21_pytest.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r05_c21_smoke.log 2>&1
timeout 1500 bash tools/profile_r05.sh > gpurun_out/r05_c21_profile.txt 2>&1
