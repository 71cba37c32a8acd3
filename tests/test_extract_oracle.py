"""CPU tests of the candidate-read extraction row: the restatement under oracle/ against the golden fixtures
the reference's own fastq-extractor produced (tests/golden/extract_*, tools/make_extract_goldens.py) and, where the reference
binary is present, against it live on fresh seeded data."""
import gzip
import json
import os
import subprocess

import pytest

import util

CASES = ["cyp_rna_2x100", "cyp_dna_2x150_noisy", "cyp_rna_single_s95", "example_reads_vs_cyp_dna"]


class XCase:
    def __init__(self, name, tmp):
        self.dir = os.path.join(util.GOLDEN, "extract_" + name)
        self.meta = json.load(open(os.path.join(self.dir, "meta.json")))
        os.makedirs(tmp, exist_ok=True)
        src = os.path.join(util.GOLDEN, self.meta["reads_from"]) if self.meta["reads_from"] else self.dir
        self.paired = self.meta["paired"]
        self.r1 = util.gunzip_to(os.path.join(src, "reads_1.fq.gz"), os.path.join(tmp, name + "_1.fq"))
        self.r2 = util.gunzip_to(os.path.join(src, "reads_2.fq.gz"), os.path.join(tmp, name + "_2.fq")) if self.paired else None
        self.ref = util.gunzip_to(os.path.join(util.GOLDEN, self.meta["reference"]), os.path.join(tmp, name + "_ref.fa"))
        with gzip.open(os.path.join(self.dir, "kept_ids.txt.gz"), "rt") as f:
            self.kept = [l for l in f.read().split("\n") if l]

    def args(self):
        return ["-f", self.ref] + (["-1", self.r1, "-2", self.r2] if self.paired else ["-u", self.r1]) + list(self.meta["flags"])

    def similarity(self):
        f = self.meta["flags"]
        return float(f[f.index("-s") + 1]) if "-s" in f else 0.8

    def strips_mate_suffix(self):
        f = self.meta["flags"]
        return not ("-t" in f and int(f[f.index("-t") + 1]) > 1)


def kept_ids(path):
    return [n for n, _ in util.fastx_records(path)]


@pytest.mark.parametrize("name", CASES)
def test_restatement_keeps_what_the_reference_kept(built, tmp_path, name):
    c = XCase(name, str(tmp_path))
    o = str(tmp_path / "orc")
    p = subprocess.run([util.ORACLE_EXTRACT] + c.args() + ["-o", o], stderr=subprocess.PIPE, text=True, check=True)
    assert "k=%d hitLenRequired=%d" % (c.meta["kmer_length"], c.meta["hit_len_required"]) in p.stderr
    assert kept_ids(o + ("_1.fq" if c.paired else ".fq")) == c.kept
    assert len(c.kept) == c.meta["kept"]
    if c.paired:
        assert kept_ids(o + "_2.fq") == c.kept


def test_low_complexity_rule(built):
    """IsLowComplexity (FastqExtractor.cpp:89-111) on hand-checked strings."""
    orc = util.ExtractOracle(util.gunzip_to(util.CYP_RNA, "/tmp/t1k_xo_ref.fa"))
    assert orc.low_complexity("A" * 50 + "CGT" * 16 + "CG")          # one base >= len / 2
    assert not orc.low_complexity("A" * 49 + "CGT" * 17)             # 49 < 50
    assert orc.low_complexity("ACGT" * 20 + "N" * 10)                # N >= len / 10 (90 / 10 = 9)
    assert not orc.low_complexity("ACGT" * 20 + "N" * 7)             # 7 < 87 / 10 = 8
    assert orc.low_complexity("AC" * 24 + "GG" + "TT")               # two bases with <= 2 occurrences
    assert not orc.low_complexity("ACG" * 30 + "TT")                 # only T is rare
    assert orc.low_complexity("")                                    # 0 >= 0
    orc.close()


def test_inferred_kmer_length(built, tmp_path):
    """SeqSet::InferKmerLength (SeqSet.hpp:2830-2845): digits of the total length in base 4, plus one."""
    fa = tmp_path / "r.fa"
    for total, want in ((255, 5), (256, 6), (4 ** 9 - 1, 10), (4 ** 9, 11)):
        fa.write_text(">a\n" + "ACGT" * (total // 4) + "ACG"[: total % 4] + "\n")
        orc = util.ExtractOracle(str(fa))
        assert orc.inferred_k == want
        orc.close()


@pytest.mark.parametrize("seed,length,sub,sim,single", [(11, 76, 0.05, 0.8, False), (12, 150, 0.15, 0.9, False), (13, 250, 0.08, 0.95, True)])
def test_restatement_vs_reference_binary_live(built, tmp_path, seed, length, sub, sim, single):
    util.need(util.REF_EXTRACT)  # decided when the test runs, after the `built` fixture had its chance to build oracle/_ref
    ref = str(tmp_path / "ref.fa")
    util.synth_ref("ref-dna", ref, seed=seed, genes=3, scale=0.05)
    pfx = str(tmp_path / "r")
    util.synth_reads(ref, pfx, seed=seed + 100, pairs=1500, len=length, bg=0.4, sub=sub, indel=0.005, nrate=0.01)
    reads = ["-u", pfx + "_2.fq"] if single else ["-1", pfx + "_1.fq", "-2", pfx + "_2.fq"]
    args = ["-f", ref] + reads + ["-s", str(sim)]
    subprocess.run([util.REF_EXTRACT] + args + ["-o", str(tmp_path / "a")], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([util.ORACLE_EXTRACT] + args + ["-o", str(tmp_path / "b")], check=True, stderr=subprocess.DEVNULL)
    for suffix in ([".fq"] if single else ["_1.fq", "_2.fq"]):
        assert open(str(tmp_path / "a") + suffix).read() == open(str(tmp_path / "b") + suffix).read()


def homopolymer_reads(tmp_path, k):
    """reference sequences with A / T runs longer than k next to an N, and reads that put an N before / after / inside such runs: the code of
    a k-mer holding an N (this program maps N to 0, FastqExtractor.cpp:51-54) decides whether its neighbour repeats the previous k-mer,
    both when the index is built (KmerIndex.hpp:121) and when the read is looked up (SeqSet.hpp:1104)"""
    import random
    rng = random.Random(9)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    body = rnd(200) + "N" + "A" * (k + 4) + rnd(150) + "T" * (k + 3) + "N" + rnd(180) + "A" * (k + 6) + rnd(120) + "N" + "T" * (k + 2) + rnd(150)
    ref = tmp_path / "homo.fa"
    ref.write_text(">s0\n%s\n>s1\n%s\n" % (body, rnd(900)))
    reads = []
    runs = [i for i in range(1, len(body)) if body[i] in "AT" and body[i - 1] != body[i] and body[i:i + k] == body[i] * k]
    for start in runs:
        ln = len(body[start:]) - len(body[start:].lstrip(body[start]))
        for npos in (None, start - 2, start - 1, start, start + 1, start + ln - 1, start + ln, start + ln + 1):
            for w0 in (start - 60, start - 25, start - 95):
                w0 = max(0, w0)
                s = list(body[w0:w0 + 120].replace("N", "G"))
                if npos is not None and 0 <= npos - w0 < len(s):
                    s[npos - w0] = "N"
                reads.append("".join(s))
                reads.append("".join(comp[c] for c in reversed(s)))
    return str(ref), reads


def test_n_next_to_homopolymers_vs_reference_binary(built, tmp_path):
    util.need(util.REF_EXTRACT)  # decided when the test runs, after the `built` fixture had its chance to build oracle/_ref
    ref, reads = homopolymer_reads(tmp_path, 9)
    fq = tmp_path / "h.fq"
    fq.write_text("".join("@h%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads)))
    split = False
    for sim in ("0.8", "0.985", "0.992", "0.999"):
        args = ["-f", ref, "-u", str(fq), "-s", sim]
        subprocess.run([util.REF_EXTRACT] + args + ["-o", str(tmp_path / "a")], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([util.ORACLE_EXTRACT] + args + ["-o", str(tmp_path / "b")], check=True, stderr=subprocess.DEVNULL)
        a, b = open(str(tmp_path / "a.fq")).read(), open(str(tmp_path / "b.fq")).read()
        assert a == b
        split |= 0 < a.count("@h") < len(reads)
    assert split
