# odd but legal read records mixed into a fixture (empty sequences, very short reads, all-N reads, tabs in headers, CRLF, no final
# newline): this build against the reference binary, every file
import os, subprocess, sys, gzip, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = "/tmp/t1k_odd"; os.makedirs(W, exist_ok=True)
open(W + "/ref.fa", "wb").write(gzip.open(ROOT + "/tests/golden/cyp2d6_rna_seq.fa.gz").read())
def recs(p):
    l = gzip.open(p, "rt").read().split("\n")
    return [l[i:i + 4] for i in range(0, len(l) - 1, 4)]
r1, r2 = recs(ROOT + "/tests/golden/cyp_rna_2x100/reads_1.fq.gz"), recs(ROOT + "/tests/golden/cyp_rna_2x100/reads_2.fq.gz")
rng = random.Random(3)
o1, o2 = [], []
for i, (a, b) in enumerate(zip(r1, r2)):
    a, b = list(a), list(b)
    k = i % 9
    if k == 1: a[1] = ""; a[3] = ""
    if k == 2: b[1] = b[1][:5]; b[3] = b[3][:5]
    if k == 3: a[1] = "N" * len(a[1])
    if k == 4: a[0] += "\tcomment with tab"; b[0] += " comment"
    if k == 5: a[1] = a[1][:37]; a[3] = a[3][:37]; b[1] = b[1][:11]; b[3] = b[3][:11]
    if k == 6: b[1] = ""; b[3] = ""; a[1] = ""; a[3] = ""
    o1.append(a); o2.append(b)
def write(path, rs, crlf=False, final_nl=True):
    t = "".join("\n".join(r) + "\n" for r in rs)
    if crlf: t = t.replace("\n", "\r\n")
    if not final_nl: t = t.rstrip("\r\n")
    open(path, "w", newline="").write(t)
res = 0
for name, kw in (("plain", {}), ("crlf", {"crlf": True}), ("nofinal", {"final_nl": False})):
    write(W + "/o1.fq", o1, **kw); write(W + "/o2.fq", o2, **kw)
    args = ["-f", W + "/ref.fa", "-1", W + "/o1.fq", "-2", W + "/o2.fq", "--alleleDigitUnits", "1", "--alleleDelimiter", "."]
    a = subprocess.run([ROOT + "/oracle/_ref/genotyper"] + args + ["-o", W + "/ref"], stderr=subprocess.PIPE, text=True)
    b = subprocess.run([ROOT + "/t1k_amd/bin/genotyper"] + args + ["-o", W + "/gpu"], stderr=subprocess.PIPE, text=True)
    print(name, "rc", a.returncode, b.returncode, end=" ")
    for s in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        same = os.path.exists(W + "/ref" + s) and os.path.exists(W + "/gpu" + s) and open(W + "/ref" + s, "rb").read() == open(W + "/gpu" + s, "rb").read()
        print(s, "ok" if same else "DIFF", end=" ")
        res |= 0 if same else 1
    print()
    if b.returncode: print(b.stderr[-300:])
sys.exit(res)
