# odd but legal records through fastq-extractor: this build against the reference binary
import os, subprocess, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = "/tmp/t1k_oddx"; os.makedirs(W, exist_ok=True)
open(W + "/ref.fa", "wb").write(gzip.open(ROOT + "/tests/golden/cyp2d6_rna_seq.fa.gz").read())
def recs(p):
    l = gzip.open(p, "rt").read().split("\n")
    return [l[i:i + 4] for i in range(0, len(l) - 1, 4)]
r1, r2 = recs(ROOT + "/tests/golden/cyp_rna_2x100/reads_1.fq.gz"), recs(ROOT + "/tests/golden/cyp_rna_2x100/reads_2.fq.gz")
for i, (a, b) in enumerate(zip(r1, r2)):
    k = i % 9
    if k == 1: a[1] = a[3] = ""
    if k == 2: b[1], b[3] = b[1][:5], b[3][:5]
    if k == 3: a[1] = "N" * len(a[1])
    if k == 4: a[0] += "\tcomment with tab"; b[0] += " comment"
    if k == 5: a[1], a[3], b[1], b[3] = a[1][:37], a[3][:37], b[1][:11], b[3][:11]
    if k == 6: a[1] = a[3] = b[1] = b[3] = ""
    if k == 7: a[1] = "ACGT" * 25; b[1] = "A" * 100
res = 0
for name, crlf in (("plain", False), ("crlf", True)):
    for m, rs in ((1, r1), (2, r2)):
        t = "".join("\n".join(r) + "\n" for r in rs)
        if crlf: t = t.replace("\n", "\r\n")
        open(W + "/o%d.fq" % m, "w", newline="").write(t)
    for mode, args in (("paired", ["-1", W + "/o1.fq", "-2", W + "/o2.fq"]), ("single", ["-u", W + "/o1.fq"])):
        outs = {}
        for who, binary in (("ref", ROOT + "/oracle/_ref/fastq-extractor"), ("gpu", ROOT + "/t1k_amd/bin/fastq-extractor")):
            for f in os.listdir(W):
                if f.startswith(who + "_"): os.remove(W + "/" + f)
            r = subprocess.run([binary, "-f", W + "/ref.fa"] + args + ["-o", W + "/" + who], stderr=subprocess.PIPE, text=True)
            outs[who] = (r.returncode, {f[len(who):]: open(W + "/" + f, "rb").read() for f in os.listdir(W) if f.startswith(who + "_")})
        same = outs["ref"] == outs["gpu"]
        print(name, mode, "rc", outs["ref"][0], outs["gpu"][0], {k: len(v) for k, v in outs["gpu"][1].items()}, "ok" if same else "DIFF")
        res |= 0 if same else 1
sys.exit(res)
