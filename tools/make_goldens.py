#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by running the REFERENCE itself (oracle/_ref/*, built by
oracle/Makefile from /root/reference).  Runs only where those binaries exist (this container).  Fixtures are data:
input reads / references (gz) and the reference's outputs for them; no reference source is stored.

  python tools/make_goldens.py
"""
import gzip
import json
import os
import random
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref")
SYNTH = os.path.join(ROOT, "tools", "t1k_synth")
CYP = "/root/reference/vcf_database/cyp2d6_idx"


def gz_write(path, text):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(text.encode())


def gz_copy(src, dst):
    gz_write(dst, open(src).read())


def run_case(name, ref_fa, reads_args, flags, tmp, ref_gz=None, paired=True, barcodes=0, premade=None):
    out = os.path.join(GOLD, name)
    os.makedirs(out, exist_ok=True)
    pfx = os.path.join(tmp, name)
    if premade:
        for i, src in enumerate(premade):
            shutil.copy(src, "%s_%d.fq" % (pfx, i + 1))
    else:
        subprocess.run([SYNTH, "reads", "--ref", ref_fa, "--out", pfx] + [str(x) for x in reads_args], check=True)
    args = ["-f", ref_fa] + (["-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] if paired else ["-u", pfx + "_1.fq"]) + flags
    if barcodes:
        args += ["--barcode", pfx + "_bc.fa"]
    o = os.path.join(tmp, name + "_out")
    p = subprocess.run([os.path.join(REF, "genotyper")] + args + ["-t", "1", "-o", o, "--outputReadAssignment"], stderr=subprocess.PIPE, text=True, check=True)
    log = p.stderr
    d = subprocess.run([os.path.join(REF, "genotyper_dbg")] + args + ["-t", "1", "-o", o + "_dbg"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       check=True)
    # last EM iteration of the -DDEBUG dump (Genotyper.hpp:1284-1286): "ec names size: readCount length. abundance"
    em = [l for l in d.stdout.splitlines() if re.match(r"^\d+ \S+ \d+: [-\d.eE+naninf]+ \d+\. [-\d.eE+naninf]+$", l)]
    nec = 0
    for l in em:
        i = int(l.split()[0])
        if i == 0 and nec:
            break
        nec = max(nec, i + 1)
    last = em[-nec:] if nec else []
    gz_copy(pfx + "_1.fq", os.path.join(out, "reads_1.fq.gz"))
    if paired:
        gz_copy(pfx + "_2.fq", os.path.join(out, "reads_2.fq.gz"))
    if barcodes:
        gz_copy(pfx + "_bc.fa", os.path.join(out, "barcodes.fa.gz"))
        shutil.copy(o + "_aligned_bc.fa", os.path.join(out, "aligned_bc.fa"))
    if ref_gz:
        gz_copy(ref_fa, os.path.join(out, ref_gz))
    shutil.copy(o + "_genotype.tsv", os.path.join(out, "genotype.tsv"))
    shutil.copy(o + "_allele.tsv", os.path.join(out, "allele.tsv"))
    gz_copy(o + "_assign.tsv", os.path.join(out, "assign.tsv.gz"))
    ids = [l[1:].strip() for l in open(o + ("_aligned_1.fa" if paired else "_aligned.fa")) if l.startswith(">")]
    gz_write(os.path.join(out, "aligned_ids.txt.gz"), "\n".join(ids) + "\n")
    gz_write(os.path.join(out, "em_last_iteration.txt.gz"), "\n".join(last) + "\n")
    m = re.search(r"in (\d+) EM iterations", log)
    m2 = re.search(r"(\d+) read fragments can be assigned \(average ([-\d.naninf]+) alleles/read\)", log)
    meta = dict(flags=flags, paired=paired, em_iterations=int(m.group(1)) if m else None, assigned_fragments=int(m2.group(1)), avg_alleles=m2.group(2),
                reads_args=[str(x) for x in reads_args], reference=os.path.basename(ref_gz) if ref_gz else None, barcodes=bool(barcodes),
                reference_version="run-t1k v1.0.9-r239 (mourisl/T1K @ 2025-06-14)")
    json.dump(meta, open(os.path.join(out, "meta.json"), "w"), indent=1, sort_keys=True)
    print(name, meta["em_iterations"], meta["assigned_fragments"], open(o + "_genotype.tsv").read().strip().replace("\n", " | ")[:200])


def ga_vectors(tmp):
    rnd = random.Random(20250614)
    pairs = []

    def rs(n, alpha="ACGT"):
        return "".join(rnd.choice(alpha) for _ in range(n))

    def mutate(s, k):
        s = list(s)
        for _ in range(k):
            i = rnd.randrange(len(s))
            s[i] = rnd.choice([c for c in "ACGT" if c != s[i]])
        return "".join(s)
    for L in list(range(1, 40)) + [64, 100, 101, 127, 128, 129, 150, 151, 200]:
        for k in [0, 1, 2, 3, 4, 5, 8]:
            for alpha in ("ACGT", "AC"):
                t = rs(L, alpha)
                pairs.append((t, mutate(t, min(k, L))))
    for _ in range(600):  # indels, unequal lengths
        L = rnd.randrange(5, 160)
        t = rs(L)
        p = list(mutate(t, rnd.randrange(0, 4)))
        for _ in range(rnd.randrange(1, 3)):
            i = rnd.randrange(len(p))
            if rnd.random() < 0.5:
                del p[i:i + rnd.randrange(1, 4)]
            else:
                p[i:i] = list(rs(rnd.randrange(1, 4)))
        if p:
            pairs.append((t, "".join(p)))
    for _ in range(300):  # N's and periodic sequences
        L = rnd.randrange(4, 150)
        unit = rs(rnd.randrange(1, 5))
        t = (unit * L)[:L]
        p = list(mutate(t, rnd.randrange(0, 6)))
        for _ in range(rnd.randrange(0, 3)):
            p[rnd.randrange(len(p))] = "N"
        t = list(t)
        if rnd.random() < 0.3:
            t[rnd.randrange(len(t))] = "N"
        pairs.append(("".join(t), "".join(p)))
    for _ in range(150):  # very different lengths (boundary quirks of the traceback)
        pairs.append((rs(rnd.randrange(1, 60)), rs(rnd.randrange(1, 12))))
        pairs.append((rs(rnd.randrange(1, 12)), rs(rnd.randrange(1, 60))))
    inp = "".join("%s %s\n" % tp for tp in pairs)
    r = subprocess.run([os.path.join(REF, "ga_harness")], input=inp, stdout=subprocess.PIPE, text=True, check=True)
    res = r.stdout.splitlines()
    assert len(res) == len(pairs)
    gz_write(os.path.join(GOLD, "ga_vectors.tsv.gz"), "".join("%s\t%s\t%s\t%s\n" % (t, p, l.split()[0], l.split()[1]) for (t, p), l in zip(pairs, res)))
    print("ga vectors:", len(pairs))


def synth_ref(kind, path, **kw):
    args = [SYNTH, kind]
    for k, v in kw.items():
        args += ["--" + k, str(v)]
    with open(path, "w") as f:
        subprocess.run(args, check=True, stdout=f)


def main():
    if not os.path.exists(os.path.join(REF, "genotyper")):
        sys.exit("oracle/_ref/genotyper missing: run `make -C oracle ref` where /root/reference exists")
    tmp = tempfile.mkdtemp(prefix="t1k_gold_")
    cyp = ["--alleleDigitUnits", "1", "--alleleDelimiter", "."]
    ga_vectors(tmp)
    run_case("cyp_rna_2x100", CYP + "/cyp2d6_rna_seq.fa", ["--pairs", 300, "--len", 100, "--seed", 11, "--sub", 0.005], cyp, tmp)
    run_case("cyp_rna_single", CYP + "/cyp2d6_rna_seq.fa", ["--pairs", 200, "--len", 100, "--seed", 21, "--sub", 0.01], cyp, tmp, paired=False)
    run_case("cyp_dna_relax_2x150", CYP + "/cyp2d6_dna_seq.fa", ["--pairs", 250, "--len", 150, "--seed", 12, "--sub", 0.005, "--fragmean", 420],
             cyp + ["-s", "0.9", "--relaxIntronAlign"], tmp)
    hla = os.path.join(tmp, "hla.fa")
    synth_ref("ref-rna", hla, genes=4, scale=0.03, seed=5)
    run_case("hla_synth_2x150", hla, ["--pairs", 400, "--len", 150, "--seed", 3, "--barcodes", 20], ["-s", "0.97"], tmp, ref_gz="ref.fa.gz", barcodes=20)
    kir = os.path.join(tmp, "kir.fa")
    synth_ref("ref-dna", kir, genes=4, scale=0.2, seed=6)
    run_case("kir_synth_relax_2x150", kir, ["--pairs", 300, "--len", 150, "--seed", 5], ["-s", "0.9", "--relaxIntronAlign"], tmp, ref_gz="ref.fa.gz")
    # the reference's own worked example (KIR reads) against CYP2D6: the empty-call path (BASELINE config 1 plumbing)
    ex = "/root/reference/example"
    sub = []
    for m in (1, 2):
        lines = open("%s/example_%d.fq" % (ex, m)).read().splitlines()[:4 * 150]
        pth = os.path.join(tmp, "ex_%d.fq" % m)
        open(pth, "w").write("\n".join(lines) + "\n")
        sub.append(pth)
    run_case("example_kir_vs_cyp", CYP + "/cyp2d6_rna_seq.fa", [], cyp + ["-s", "0.97"], tmp, premade=sub)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
