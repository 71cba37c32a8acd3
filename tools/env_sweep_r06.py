"""Round 6: the executable on the committed fixtures under RANDOM loop geometries (first window, window limit, assignment range, pairing range,
pipelines, ranks, coverage mode, cross-window table): every run must exit 0 and write the reference's committed files.  Meant to meet block sizes and
range counts nobody picked by hand (a clear one entry past its block had survived five rounds of fixed small-window settings).
usage: python tools/env_sweep_r06.py [runs] [seed]   (on a GPU box)"""
import hashlib, os, random, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import goldens, util
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
exe = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")
tmp = tempfile.mkdtemp(prefix="sweep_")
cases = {n: goldens.Case(n, tmp) for n in ("hla_synth_2x150", "cyp_dna_relax_2x150", "cyp_rna_single", "kir_synth_relax_2x150", "cyp_rna_2x100")}
bad = 0
for i in range(runs):
    name = rng.choice(sorted(cases))
    c = cases[name]
    env = {"T1K_FIRST_WINDOW": str(rng.randint(8, 300)), "T1K_WINDOW": str(rng.randint(16, 1500)), "T1K_BATCH": str(rng.randint(8, 200)),
           "T1K_PAIR_BATCH": str(rng.randint(8, 400)), "T1K_PIPELINES": str(rng.randint(1, 4))}
    if rng.random() < 0.25: env["T1K_GPUS"] = rng.choice(["0,0", "0,0,0"])
    if rng.random() < 0.2: env["T1K_COVERAGE"] = "eager"
    if rng.random() < 0.2: env["T1K_CROSS_WINDOW"] = "0"
    if rng.random() < 0.15: env["T1K_ARCHIVE_GB"] = "0.001"
    if rng.random() < 0.15: env["T1K_HOST_CHAIN"] = "1"
    out = os.path.join(tmp, "o")
    r = subprocess.run([exe] + c.args() + ["-o", out, "--outputReadAssignment"], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, env=dict(os.environ, **env))
    ok = r.returncode == 0
    why = ""
    if not ok: why = "rc %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "")
    else:
        for suf, key in (("_genotype.tsv", "genotype.tsv"), ("_allele.tsv", "allele.tsv"), ("_assign.tsv", "assign.tsv.gz")):
            if open(out + suf).read() != c.expected(key): ok = False; why = suf + " differs"; break
    if not ok:
        bad += 1
        print("FAIL %s %s -> %s" % (name, " ".join("%s=%s" % kv for kv in sorted(env.items())), why), flush=True)
print("%d runs, %d failed" % (runs, bad))
sys.exit(1 if bad else 0)
