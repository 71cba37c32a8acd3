#!/bin/bash
# A/B runs of round 3 on the GPU box: the bench step at 1 M (and optionally 10 M) pairs under a list of environment variants; one line per
# variant with the step time, the device-loop time and the per-kernel-family times (HIP events), plus the md5 of the genotype text.
#   tools/ab_r03.sh "NAME=VALUE ...|NAME=VALUE ...|..."  [pairs] [steps]
mkdir -p gpurun_out /tmp/t1k_bench
PAIRS=${2:-1000000}; STEPS=${3:-3}
IFS='|' read -ra VARS <<< "${1:-}"
[ ${#VARS[@]} -eq 0 ] && VARS=("")
for v in "${VARS[@]}"; do
  env $v python bench.py --pairs $PAIRS --steps $STEPS --warmup 1 --no-cpu-baseline --no-executable-check 2> /tmp/ab_err.log | tail -1 > /tmp/ab_line.json
  python - "$v" <<'PY'
import json, sys, hashlib
try:
    d = json.load(open("/tmp/ab_line.json"))
except Exception as e:
    print("%-48s FAILED: %s" % (sys.argv[1] or "(default)", open("/tmp/ab_err.log").read()[-600:])); sys.exit(0)
r, c = d["roofline"], d["config"]
ms = r["all_kernels_ms_per_step"]
chk = c.get("reference_output_check", {})
print("%-48s step %.0f ms  loop %.0f  em %.0f | seed %.0f (%.2f/launch, frac %.3f) chain %.0f extend %.0f select %.0f fullalign %.0f pair %.0f | md5 %s %s" % (
    sys.argv[1] or "(default)", d["ms_per_step"], c["phases_ms"]["device_loop"], c["phases_ms"]["em"], ms["k_seed_groups"], r["avg_launch_ms"], r["frac"], ms["chain kernels"], ms["k_extend"],
    ms["k_select"], ms["fullalign kernels"], ms["k_pair"], hashlib.md5(open("/tmp/t1k_bench/last_genotype.tsv", "rb").read()).hexdigest()[:8],
    ("ref-check " + ("FAILED" if chk.get("FAILED") else "ok")) if chk else ""))
PY
done 2>&1 | tee -a gpurun_out/r03_ab.log
