#!/bin/bash
# the genotyper executable on the bench input, cold process, with the job's phase lines (run on the GPU box)
W=/tmp/t1k_bench; P=${1:-10000000}
python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=2)"
for i in 1 2 3; do
  python - "$W" "$P" "$i" <<'PY'
import subprocess, sys, time, hashlib, os
W, P, i = sys.argv[1], sys.argv[2], sys.argv[3]
t = time.time()
r = subprocess.run(["t1k_amd/bin/genotyper", "-f", W + "/hla_g24_s1.0.fa", "-1", "%s/reads_g24_s1.0_p%s_seed2_1.fq" % (W, P), "-2", "%s/reads_g24_s1.0_p%s_seed2_2.fq" % (W, P), "-s", "0.97", "-o", W + "/exe_cold"],
                   stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_DEBUG_PHASES="1"))
print("run %s: wall %.2f s rc %d md5 %s" % (i, time.time() - t, r.returncode, hashlib.md5(open(W + "/exe_cold_genotype.tsv", "rb").read()).hexdigest()[:8]))
print("\n".join(l[:250] for l in r.stderr.splitlines() if "t1k job" in l))
PY
  sleep ${2:-20}
done
