"""Randomised CPU comparison of the PRODUCT's host tables (refset.cpp + genotype.cpp through tests/harness/host_tables_harness.cpp) with the
oracle CLI on mixed samples (genes with three and more allele types, the type-pair search): python tools/host_tables_fuzz.py <seed> <cases>.
Test infrastructure only."""
import os, subprocess, sys, tempfile, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import util
import test_host_tables_cpu as tht
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
exe = os.path.join(tempfile.mkdtemp(), "host_tables_harness")
HOST = tht.HOST
subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(util.ROOT, "tests", "harness", "host_tables_harness.cpp"),
                os.path.join(HOST, "refset.cpp"), os.path.join(HOST, "genotype.cpp"), "-lz", "-lpthread"], check=True)
bad = 0; three = 0
for it in range(n):
    tmp = tempfile.mkdtemp()
    kind = rnd.choice(["ref-rna", "ref-rna", "ref-dna"])
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref(kind, ref, genes=rnd.randint(2, 6), scale=rnd.choice([0.01, 0.02, 0.04]), seed=rnd.randint(1, 10**6))
    parts = rnd.choice([2, 3, 4, 6])
    L = rnd.choice([75, 100, 150])
    for p in range(parts):
        util.synth_reads(ref, os.path.join(tmp, "p%d" % p), pairs=rnd.randint(100, 400), len=L, seed=rnd.randint(1, 10**6), sub=rnd.choice([0.002, 0.01]))
    for m in ("1", "2"):
        with open(os.path.join(tmp, "r_%s.fq" % m), "w") as o:
            for p in range(parts): o.write(open(os.path.join(tmp, "p%d_%s.fq" % (p, m))).read())
    flags = ["-s", rnd.choice(["0.9", "0.95", "0.97"])]
    if rnd.random() < 0.3: flags += ["--frac", rnd.choice(["0.05", "0.3"])]
    if rnd.random() < 0.3: flags += ["--cov", rnd.choice(["0.5", "3"])]
    if rnd.random() < 0.3: flags += ["--crossGeneRate", rnd.choice(["0", "0.2"])]
    r1, r2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
    orc = os.path.join(tmp, "orc")
    rb = subprocess.run([util.ORACLE_CLI, "-f", ref, "-1", r1, "-2", r2] + flags + ["-o", orc], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    if rb.returncode: print(it, "oracle rc", rb.returncode); bad += 1; continue
    out = os.path.join(tmp, "host")
    tht.run_host(exe, ref, orc, tht.longest_read(r1, r2), flags, out)
    ga = open(orc + "_genotype.tsv").read()
    t3 = sum(1 for l in ga.splitlines() if l.split("\t")[-1] != "")
    three += t3 > 0
    ok = all(open(out + w).read() == open(orc + w).read() for w in ("_genotype.tsv", "_allele.tsv"))
    print(it, kind, parts, flags, "third-column genes", t3, "OK" if ok else "DIFF", flush=True)
    if not ok: bad += 1
print("bad", bad, "cases with >2 types", three)
