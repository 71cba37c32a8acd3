#!/bin/bash
# usage: bench_ab.sh "ENV1=.. ENV2=.." ...  -- one bench.py run (10M pairs, 2 timed steps) per argument, with that environment
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], "| pairs/s %.0f | ms/step %.0f |" % (d["value"], d["ms_per_step"]), {k: round(v) for k, v in d["config"]["phases_ms"].items()}, "distinct", d["config"]["distinct_read_ends"])
PY
done
