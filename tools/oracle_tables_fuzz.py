"""Randomised CPU comparison of the oracle's pruning / selection / tables with the reference binary (oracle/_ref/genotyper):
python tools/oracle_tables_fuzz.py <seed> <cases>.  Test infrastructure only; needs /root/reference to have been built by oracle/Makefile."""
import os, subprocess, sys, tempfile, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import util
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0; three = 0
for it in range(n):
    tmp = tempfile.mkdtemp()
    kind = rnd.choice(["ref-rna", "ref-dna"])
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref(kind, ref, genes=rnd.randint(2, 6), scale=rnd.choice([0.01, 0.02, 0.04]), seed=rnd.randint(1, 10**6))
    parts = rnd.choice([1, 1, 2, 3])
    L = rnd.choice([75, 100, 150])
    for p in range(parts):
        util.synth_reads(ref, os.path.join(tmp, "p%d" % p), pairs=rnd.randint(60, 300), len=L, seed=rnd.randint(1, 10**6), sub=rnd.choice([0.002, 0.01]), bg=rnd.choice([0.01, 0.2]))
    for m in ("1", "2"):
        with open(os.path.join(tmp, "r_%s.fq" % m), "w") as o:
            for p in range(parts): o.write(open(os.path.join(tmp, "p%d_%s.fq" % (p, m))).read())
    flags = ["-s", rnd.choice(["0.8", "0.9", "0.95", "0.97"])]
    if kind == "ref-dna" and rnd.random() < 0.5: flags += ["--relaxIntronAlign"]
    if rnd.random() < 0.3: flags += ["--frac", rnd.choice(["0.05", "0.3", "0.5"])]
    if rnd.random() < 0.3: flags += ["--cov", rnd.choice(["0.5", "3", "20"])]
    if rnd.random() < 0.3: flags += ["--crossGeneRate", rnd.choice(["0", "0.2", "1.0"])]
    if rnd.random() < 0.2: flags += ["--squaremMinAlpha", "-2"]
    single = rnd.random() < 0.2
    args = ["-f", ref] + (["-u", os.path.join(tmp, "r_1.fq")] if single else ["-1", os.path.join(tmp, "r_1.fq"), "-2", os.path.join(tmp, "r_2.fq")]) + flags
    ra = subprocess.run([util.REF_BIN] + args + ["-o", os.path.join(tmp, "a"), "-t", "1"], stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    rb = subprocess.run([util.ORACLE_CLI] + args + ["-o", os.path.join(tmp, "b")], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    if ra.returncode or rb.returncode: print(it, "rc", ra.returncode, rb.returncode, rb.stderr[-200:]); bad += 1; continue
    ga, gb = open(os.path.join(tmp, "a_genotype.tsv")).read(), open(os.path.join(tmp, "b_genotype.tsv")).read()
    aa, ab = open(os.path.join(tmp, "a_allele.tsv")).read(), open(os.path.join(tmp, "b_allele.tsv")).read()
    t3 = sum(1 for l in ga.splitlines() if l.split("\t")[-1] != "")
    three += t3 > 0
    ok = ga == gb and aa == ab
    print(it, kind, parts, flags, "third-column genes", t3, "OK" if ok else "DIFF")
    if not ok:
        bad += 1
        print(ga); print("---"); print(gb); print(aa); print("---"); print(ab)
print("bad", bad, "cases with >2 types", three)
