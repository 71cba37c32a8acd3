mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -15 > gpurun_out/r05_c16_pytest.log
timeout 900 bash tools/cold_r05.sh "|T1K_SERIAL_CONTEXTS=1||T1K_SERIAL_CONTEXTS=1" > gpurun_out/r05_c16_cold.txt 2>&1
cp gpurun_out/r05_cold.log gpurun_out/r05_c16_cold.log
