#!/bin/bash
# A/B of an environment switch on the 1 M-pair bench step (kernels alone with one pipeline, and three pipelines): tools/ab_env.sh VAR=value
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import bench; bench.ensure_inputs('/tmp/t1k_bench', 1000000, 24, 1.0, seed=2)"
for rep in 1 2; do
for v in base "$1"; do
  for pl in 1 3; do
    env T1K_PIPELINES=$pl $( [ "$v" = base ] || echo "$v" ) python bench.py --pairs 1000000 --steps 3 --warmup 1 --no-cpu-baseline --no-executable-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-24s pipelines=$pl rep=$rep: step %.1f ms, loop %.1f ms, seed launch %.3f ms, kernels %s' % ('$v', d['ms_per_step'], d['config']['phases_ms']['device_loop'], r['avg_launch_ms'], {k:round(v,1) for k,v in r['all_kernels_ms_per_step'].items()}))"
  done
done
done
