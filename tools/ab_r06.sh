#!/bin/bash
# A/B runs of round 5 on the GPU box (same shape as tools/ab_r03.sh): the bench step under a list of environment variants
#   tools/ab_r04.sh "NAME=VALUE ...|NAME=VALUE ...|..."  [pairs] [steps]     -> appended to gpurun_out/r06_ab.log
mkdir -p gpurun_out /tmp/t1k_bench
PAIRS=${2:-10000000}; STEPS=${3:-2}
IFS='|' read -ra VARS <<< "${1:-}"
[ ${#VARS[@]} -eq 0 ] && VARS=("")
for v in "${VARS[@]}"; do
  env $v python bench.py --pairs $PAIRS --steps $STEPS --warmup 1 --no-cpu-baseline --no-executable-check --no-roofline-step 2> /tmp/ab_err.log | tail -1 > /tmp/ab_line.json
  python - "$v" <<'PY'
import json, sys, hashlib
try:
    d = json.load(open("/tmp/ab_line.json"))
except Exception as e:
    print("%-44s FAILED: %s" % (sys.argv[1] or "(default)", open("/tmp/ab_err.log").read()[-600:])); sys.exit(0)
r, c = d["roofline"], d["config"]
ms = r["all_kernels_ms_per_step"]
chk = c.get("reference_output_check", {})
cm = c["calls_ms"]
print("%-44s step %.0f ms | create %.0f load %.0f run %.0f (loop %.0f coalesce %.0f em %.0f) | seed %.0f chain %.0f extend %.0f select %.0f trunc %.0f pair %.0f | md5 %s %s" % (
    sys.argv[1] or "(default)", d["ms_per_step"], cm["job_create_reference"], cm["load_reads"], cm["run"], c["phases_ms"]["device_loop"], c["phases_ms"]["coalesce"], c["phases_ms"]["em"],
    ms["k_seed_groups"], ms["chain kernels"], ms["k_extend"], ms["k_select"], ms["fullalign kernels"], ms["k_pair"],
    hashlib.md5(open("/tmp/t1k_bench/last_genotype.tsv", "rb").read()).hexdigest()[:8], ("ref-check " + ("FAILED" if chk.get("FAILED") else "ok")) if chk else ""))
PY
done 2>&1 | tee -a gpurun_out/r06_ab.log
