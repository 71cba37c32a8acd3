"""Round 6: the analyzer (variant calling on) on the two committed novel-SNP samples under random settings of its pieces, the loop geometry, the
variant caller's threads and the alignment fast path: every run must write the reference analyzer's committed _allele.vcf and _barcode_expr.tsv
(tests/golden/analyzer_variants).  usage: python tools/env_sweep_analyzer_r06.py [runs] [seed]   (on a GPU box)"""
import os, random, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import util
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
geno = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")
ana = os.path.join(util.ROOT, "t1k_amd", "bin", "analyzer")
samples = {}
for het in (False, True):
    tmp = tempfile.mkdtemp(prefix="asweep_")
    ref, pfx = util.novel_snp_sample(tmp, het)
    g = os.path.join(tmp, "g")
    subprocess.run([geno, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "--barcode", pfx + "_bc.fa", "-o", g], check=True, stderr=subprocess.DEVNULL)
    samples[het] = (tmp, ref, g, os.path.join(util.GOLDEN, "analyzer_variants", "het" if het else "homo"))
bad = 0
for i in range(runs):
    het = rng.random() < 0.5
    tmp, ref, g, gold = samples[het]
    env = {"T1K_ANALYZER_PIECE": str(rng.choice([2, 3, 7, 16, 33, 64, 100, 257, 1000, 32768])), "T1K_VARIANTS_THREADS": str(rng.randint(1, 9)),
           "T1K_FIRST_WINDOW": str(rng.randint(8, 400)), "T1K_WINDOW": str(rng.randint(16, 3000)), "T1K_BATCH": str(rng.randint(8, 300)),
           "T1K_PAIR_BATCH": str(rng.randint(8, 500)), "T1K_PIPELINES": str(rng.randint(1, 3))}
    if rng.random() < 0.25: env["T1K_ANALYZER_NO_FAST"] = "1"
    a = os.path.join(tmp, "a")
    r = subprocess.run([ana, "-f", ref, "-a", g + "_allele.tsv", "-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa", "--barcode", g + "_aligned_bc.fa", "-o", a, "-t", str(rng.randint(1, 8))],
                       stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, env=dict(os.environ, **env))
    why = ""
    if r.returncode != 0: why = "rc %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "")
    elif open(a + "_allele.vcf").read() != open(gold + "_allele.vcf").read(): why = "_allele.vcf differs"
    elif open(a + "_barcode_expr.tsv").read() != open(gold + "_barcode_expr.tsv").read(): why = "_barcode_expr.tsv differs"
    if why:
        bad += 1
        print("FAIL %s %s -> %s" % ("het" if het else "homo", " ".join("%s=%s" % kv for kv in sorted(env.items())), why), flush=True)
print("%d analyzer runs, %d failed" % (runs, bad))
sys.exit(1 if bad else 0)
