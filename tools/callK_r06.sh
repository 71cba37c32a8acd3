#!/bin/bash
# Round 6: the tests added or changed after the final suite run (call J), on the GPU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "linear_reference or analyzer_variant_fixtures or streamed or gzip or analyzer_live or golden_reference_outputs" ) > gpurun_out/r06_pytest_gpu_late_tests.log 2>&1
tail -8 gpurun_out/r06_pytest_gpu_late_tests.log
