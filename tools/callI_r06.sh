#!/bin/bash
# Round 6, ninth GPU call: k_chain_fast<5, 0> built for seven wavefronts per SIMD (72 VGPRs, spills) gives wrong, varying results (call H) -- with one pipeline? host-driven chain? eager coverage?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/t1k_bench; export TMPDIR=/tmp
L=gpurun_out/r06_callI.log; : > $L
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
R=$(pwd)
for rep in 1 2; do
for env in "" "T1K_PIPELINES=1" "T1K_HOST_CHAIN=1" "T1K_PIPELINES=1 T1K_HOST_CHAIN=1" "T1K_REF_TRANSPOSE=0" "AMD_SERIALIZE_KERNEL=3"; do
  env $env T1K_GPU_LIB=$R/t1k_amd/lib/variants/libt1k_vcf07.so python bench.py --pairs 1000000 --steps 1 --warmup 0 --no-cpu-baseline --no-executable-check --no-roofline-step > /tmp/h.json 2>/dev/null
  echo "vcf07 [$env]: rc $? genotype md5 $(md5sum < /tmp/t1k_bench/last_genotype.tsv | cut -c1-8)" | tee -a $L
done
done
