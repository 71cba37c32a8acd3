#!/bin/bash
# Round 6, final measurements: the whole -m gpu suite, smoke(), bench.py as the driver runs it, the profiles behind the roofline record
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callJ.log; : > $L
say() { echo "$@" | tee -a $L; }
say "== pytest -m gpu"
( time timeout 1700 python -m pytest tests -m gpu -q --durations=8 ) > gpurun_out/r06_pytest_gpu.log 2>&1
tail -16 gpurun_out/r06_pytest_gpu.log | tee -a $L
say "== smoke()"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "rc $?" | tee -a $L; tail -3 gpurun_out/r06_smoke.log | cut -c1-300 | tee -a $L
say "== bench.py (defaults)"
( time python bench.py ) > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "rc $?" | tee -a $L
tail -c 1500 gpurun_out/r06_bench.json | tee -a $L
say "== profiles"
tools/profile_r06.sh > gpurun_out/r06_profile.out 2>&1; tail -40 gpurun_out/r06_profile.out | cut -c1-300 | tee -a $L
