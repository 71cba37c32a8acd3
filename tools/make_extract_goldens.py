#!/usr/bin/env python3
"""Golden fixtures of the candidate-read extractor (tests/golden/extract_*): inputs plus what the REFERENCE's own fastq-extractor
(oracle/_ref/fastq-extractor, built by oracle/Makefile from /root/reference/FastqExtractor.cpp) keeps for them.  Runs only where that
binary exists (this container).  Fixtures are data: reads (gz), the ids the reference kept, its parameters; no reference source.

  python tools/make_extract_goldens.py
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REFBIN = os.path.join(ROOT, "oracle", "_ref", "fastq-extractor")
ORC = os.path.join(ROOT, "oracle", "t1k_oracle_extract")
SYNTH = os.path.join(ROOT, "tools", "t1k_synth")


def gunzip_to(src, dst):
    with gzip.open(src, "rb") as a, open(dst, "wb") as b:
        shutil.copyfileobj(a, b)
    return dst


def gz_copy(src, dst):
    with open(src, "rb") as a, gzip.GzipFile(dst, "wb", mtime=0) as b:
        shutil.copyfileobj(a, b)


def ids_of(path):
    return [l[1:].rstrip("\n") for i, l in enumerate(open(path)) if l[0] in "@>" and (i % 4 == 0 or l[0] == ">")]


def case(name, ref_gz, tmp, synth_args=None, reads_from=None, paired=True, flags=()):
    out = os.path.join(GOLD, "extract_" + name)
    os.makedirs(out, exist_ok=True)
    ref = gunzip_to(os.path.join(GOLD, ref_gz), os.path.join(tmp, name + "_ref.fa"))
    pfx = os.path.join(tmp, name)
    if reads_from:
        for i in (1, 2):
            gunzip_to(os.path.join(GOLD, reads_from, "reads_%d.fq.gz" % i), "%s_%d.fq" % (pfx, i))
    else:
        subprocess.run([SYNTH, "reads", "--ref", ref, "--out", pfx] + [str(x) for x in synth_args], check=True)
        gz_copy(pfx + "_1.fq", os.path.join(out, "reads_1.fq.gz"))
        if paired:
            gz_copy(pfx + "_2.fq", os.path.join(out, "reads_2.fq.gz"))
    args = ["-f", ref] + (["-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] if paired else ["-u", pfx + "_1.fq"]) + list(flags)
    o = os.path.join(tmp, name + "_out")
    subprocess.run([REFBIN] + args + ["-o", o], check=True, stderr=subprocess.DEVNULL)
    kept = ids_of(o + ("_1.fq" if paired else ".fq"))
    # parameters the reference derived (printed by the restatement's driver, which must agree with the reference on the kept set)
    p = subprocess.run([ORC] + args + ["-o", o + "_orc"], check=True, stderr=subprocess.PIPE, text=True)
    m = re.search(r"k=(\d+) hitLenRequired=(\d+)", p.stderr)
    assert ids_of(o + "_orc" + ("_1.fq" if paired else ".fq")) == kept
    with gzip.GzipFile(os.path.join(out, "kept_ids.txt.gz"), "wb", mtime=0) as f:
        f.write(("\n".join(kept) + "\n").encode() if kept else b"")
    total = len(ids_of(pfx + "_1.fq"))
    json.dump({"reference": ref_gz, "reads_from": reads_from, "paired": paired, "flags": list(flags), "kmer_length": int(m.group(1)),
               "hit_len_required": int(m.group(2)), "fragments": total, "kept": len(kept)}, open(os.path.join(out, "meta.json"), "w"), indent=1)
    print(name, "kept", len(kept), "of", total, "k", m.group(1), "hitLen", m.group(2))


def main():
    tmp = tempfile.mkdtemp(prefix="t1k_xgold_")
    case("cyp_rna_2x100", "cyp2d6_rna_seq.fa.gz", tmp, ["--seed", 101, "--pairs", 1500, "--len", 100, "--bg", 0.4, "--sub", 0.06, "--indel", 0.004, "--nrate", 0.01])
    case("cyp_dna_2x150_noisy", "cyp2d6_dna_seq.fa.gz", tmp, ["--seed", 102, "--pairs", 1200, "--len", 150, "--bg", 0.3, "--sub", 0.15, "--indel", 0.01, "--nrate", 0.03],
         flags=["-t", "4"])
    case("cyp_rna_single_s95", "cyp2d6_rna_seq.fa.gz", tmp, ["--seed", 103, "--pairs", 1500, "--len", 125, "--bg", 0.3, "--sub", 0.03], paired=False, flags=["-s", "0.95"])
    case("example_reads_vs_cyp_dna", "cyp2d6_dna_seq.fa.gz", tmp, reads_from="example_kir_vs_cyp")
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
