# barcode edge cases: every record "missing_barcode"; only the first / last records kept -- this build against the reference binary
import os, subprocess, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = "/tmp/t1k_bc"; os.makedirs(W, exist_ok=True)
open(W + "/ref.fa", "wb").write(gzip.open(ROOT + "/tests/golden/cyp2d6_rna_seq.fa.gz").read())
for m in (1, 2):
    open(W + "/r%d.fq" % m, "wb").write(gzip.open(ROOT + "/tests/golden/cyp_rna_2x100/reads_%d.fq.gz" % m).read())
n = sum(1 for _ in open(W + "/r1.fq")) // 4
names = [l[1:].split()[0].rsplit("/", 1)[0] for i, l in enumerate(open(W + "/r1.fq")) if i % 4 == 0]
res = 0
for tag, keep in (("all_missing", lambda i: False), ("first_last", lambda i: i in (0, n - 1)), ("every_7th_missing", lambda i: i % 7 != 0)):
    with open(W + "/bc.fa", "w") as f:
        for i in range(n):
            f.write(">%s\n%s\n" % (names[i], ("ACGTACGTAC" + "ACGT"[i % 4] * 2) if keep(i) else "missing_barcode"))
    args = ["-f", W + "/ref.fa", "-1", W + "/r1.fq", "-2", W + "/r2.fq", "--barcode", W + "/bc.fa", "--alleleDigitUnits", "1", "--alleleDelimiter", "."]
    a = subprocess.run([ROOT + "/oracle/_ref/genotyper"] + args + ["-o", W + "/ref"], stderr=subprocess.PIPE, text=True)
    for env_tag, env in (("1gpu", {}), ("2ranks", {"T1K_GPUS": "0,0"}), ("tiny-windows", {"T1K_FIRST_WINDOW": "8", "T1K_WINDOW": "40", "T1K_BATCH": "8", "T1K_PAIR_BATCH": "8"})):
        b = subprocess.run([ROOT + "/t1k_amd/bin/genotyper"] + args + ["-o", W + "/gpu"], stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        print(tag, env_tag, "rc", a.returncode, b.returncode, end=" ")
        for s in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa", "_aligned_bc.fa"):
            same = os.path.exists(W + "/ref" + s) and os.path.exists(W + "/gpu" + s) and open(W + "/ref" + s, "rb").read() == open(W + "/gpu" + s, "rb").read()
            print(s, "ok" if same else "DIFF", end=" ")
            res |= 0 if same else 1
        print()
        if b.returncode: print(b.stderr[-300:])
sys.exit(res)
