#!/bin/bash
# Round 6, eighth GPU call: which of the register-budget variants of call G breaks the result (1 M pairs; expected genotype md5 3d0929a8)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callH.log; : > $L
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
R=$(pwd)
for rep in 1 2 3; do
for v in main vcol7 vcf07 vext7 vtrs6 vtrl8 vsel7 occA; do
  LIB=$R/t1k_amd/lib/variants/libt1k_$v.so; [ "$v" = "main" ] && LIB=$R/t1k_amd/lib/libt1k_gpu.so
  T1K_GPU_LIB=$LIB python bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check --no-roofline-step > /tmp/h.json 2>/dev/null
  echo "$v: rc $? genotype md5 $(md5sum < /tmp/t1k_bench/last_genotype.tsv | cut -c1-8) $(python -c "import json; d=json.loads(open('/tmp/h.json').read().strip().splitlines()[-1]); print('md5_ok', d.get('reference_md5_ok'), 'step %.0f ms' % d['ms_per_step'])" 2>/dev/null)" | tee -a $L
done
done
