"""Shared helpers for the test-suite (test infrastructure: the only place, with bench.py's cpu_baseline leg and
__graft_entry__.smoke(), that touches oracle/)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
SYNTH = os.path.join(ROOT, "tools", "t1k_synth")
ORACLE_SO = os.path.join(ROOT, "oracle", "libt1k_oracle.so")
ORACLE_CLI = os.path.join(ROOT, "oracle", "t1k_oracle_cli")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "genotyper")
ORACLE_EXTRACT = os.path.join(ROOT, "oracle", "t1k_oracle_extract")
REF_EXTRACT = os.path.join(ROOT, "oracle", "_ref", "fastq-extractor")
REF_ANALYZER = os.path.join(ROOT, "oracle", "_ref", "analyzer")
REF_READS = os.path.join(ROOT, "oracle", "_ref", "reads_harness")
CYP_RNA = os.path.join(GOLDEN, "cyp2d6_rna_seq.fa.gz")
CYP_DNA = os.path.join(GOLDEN, "cyp2d6_dna_seq.fa.gz")
CYP_FLAGS = ["--alleleDigitUnits", "1", "--alleleDelimiter", "."]


def need(path):
    """skip the calling test (at run time, not at collection time) when a reference-built binary is absent"""
    import pytest
    if not os.path.exists(path):
        pytest.skip("%s not built (oracle/Makefile builds it from /root/reference where that exists)" % os.path.relpath(path, ROOT))


def synth(*args):
    subprocess.run([SYNTH] + [str(a) for a in args], check=True)


def synth_ref(kind, path, **kw):
    args = [SYNTH, kind]
    for k, v in kw.items():
        args += ["--" + k, str(v)]
    with open(path, "w") as f:
        subprocess.run(args, check=True, stdout=f)


def synth_reads(ref, prefix, **kw):
    args = [SYNTH, "reads", "--ref", ref, "--out", prefix]
    for k, v in kw.items():
        args += ["--" + k, str(v)]
    subprocess.run(args, check=True)


def novel_snp_sample(tmp, het):
    """reads drawn from a copy of the reference in which one exonic position of gene 0 carries a base the database does not know -- in every
    allele of the gene (a homozygous novel SNP) or in one of the two alleles the reads are drawn from (het) -- genotyped against
    the ORIGINAL reference: the input on which the reference's VariantCaller does call a variant"""
    ref = os.path.join(tmp, "ref.fa")
    synth_ref("ref-rna", ref, genes=4, scale=0.05, seed=31)
    recs = []
    for line in open(ref):
        if line.startswith(">"):
            recs.append([line.rstrip("\n"), ""])
        else:
            recs[-1][1] += line.strip()
    gene0 = sorted(set(r[0][1:].split("*")[0] for r in recs))[0]
    swap = {"A": "C", "C": "G", "G": "T", "T": "A"}
    mut = os.path.join(tmp, "ref_mut.fa")
    with open(mut, "w") as o:
        k = 0
        for n, sq in recs:
            if n[1:].split("*")[0] == gene0 and len(sq) > 400 and sq[400] in swap:
                if not het or k % 8 == 2:  # (one of the two alleles the read sampler draws for this gene with this seed: the reference calls the SNP on that allele only)
                    sq = sq[:400] + swap[sq[400]] + sq[401:]
                k += 1
            o.write(n + "\n" + sq + "\n")
    pfx = os.path.join(tmp, "r")
    synth_reads(mut, pfx, pairs=3000, len=150, seed=5, barcodes=50, sub=0.0)
    return ref, pfx


def read_fa(path):
    recs = []
    for line in open(path):
        if line.startswith(">"):
            recs.append([line[1:].split()[0], line.rstrip("\n"), ""])
        else:
            recs[-1][2] += line.strip()
    return recs


def several_snps_sample(tmp, seed, genes=5, every=3, positions=(150, 152, 400, 800), pairs=4000, sub=0.002, kind="ref-rna", scale=0.05, **reads_kw):
    """like util.novel_snp_sample with several unknown bases per gene -- two of them three bases apart, so that one read-end spans both and
    their candidates fall into one group -- carried by two alleles in three, and sequencing errors on top"""
    ref = os.path.join(tmp, "ref.fa")
    if os.path.exists(kind):  # a reference file of its own (a real database) instead of a synthetic one
        gunzip_to(kind, ref) if kind.endswith(".gz") else __import__("shutil").copy(kind, ref)
    else:
        synth_ref(kind, ref, genes=genes, scale=scale, seed=seed)
    swap = {"A": "C", "C": "G", "G": "T", "T": "A"}
    mut = os.path.join(tmp, "ref_mut.fa")
    k = 0
    with open(mut, "w") as o:
        for name, head, sq in read_fa(ref):
            if k % every != every - 1:
                s = list(sq)
                for p in positions:
                    if p < len(s) and s[p] in swap:
                        s[p] = swap[s[p]]
                sq = "".join(s)
            k += 1
            o.write(head + "\n" + sq + "\n")
    pfx = os.path.join(tmp, "r")
    synth_reads(mut, pfx, pairs=pairs, len=150, seed=seed + 1, barcodes=40, sub=sub, **reads_kw)
    return ref, pfx


def gunzip_to(src, dst):
    import gzip
    import shutil
    with gzip.open(src, "rb") as a, open(dst, "wb") as b:
        shutil.copyfileobj(a, b)
    return dst


class Oracle:
    """ctypes view of oracle/libt1k_oracle.so (CPU restatement) -- the checker, never the thing under test."""

    def __init__(self, fasta, similarity=0.8, relax=False, max_assign=2000, digit_units=-1, delimiter=b"\0"):
        L = C.CDLL(ORACLE_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_char]
        L.orc_load_reference.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_assign_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_coverage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_allele_count.argtypes = [C.c_void_p]
        L.orc_allele_name.argtypes = [C.c_void_p, C.c_int]
        L.orc_allele_name.restype = C.c_char_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_global_alignment.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        self.L = L
        self.h = L.orc_create(similarity, 1 if relax else 0, max_assign, digit_units, delimiter)
        if fasta is not None:
            n = L.orc_load_reference(self.h, fasta.encode())
            assert n > 0, "oracle could not load %s" % fasta
            self.n_alleles = n

    def assign_read(self, seq, weight=1, cap=65536):
        out = np.zeros((cap, 12), dtype=np.int32)
        sim = np.zeros(cap, dtype=np.float64)
        n = self.L.orc_assign_read(self.h, seq.encode(), weight, out.ctypes.data, sim.ctypes.data, cap)
        assert n <= cap
        return out[:n], sim[:n]

    def coverage(self, allele, length):
        out = np.zeros(length, dtype=np.int32)
        self.L.orc_coverage(self.h, allele, out.ctypes.data, length)
        return out

    def global_alignment(self, t, p):
        ops = np.zeros(len(t) + len(p) + 2, dtype=np.int8)
        n = C.c_int()
        s = self.L.orc_global_alignment(t.encode(), len(t), p.encode(), len(p), ops.ctypes.data, C.byref(n))
        return s, ops[:n.value].copy()


class ExtractOracle:
    """ctypes view of the extraction restatement (oracle/oracle_extract.cpp) -- the checker of t1k_extract_batch."""

    def __init__(self, fasta, similarity=0.8, k=None, hit_len_required=27):
        L = C.CDLL(ORACLE_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_char]
        for f in ("orc_load_reference_fa", "orc_has_hit_in_set", "orc_is_good_candidate"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_char_p]
        L.orc_infer_kmer_length.argtypes = [C.c_void_p]
        L.orc_set_extract_params.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_is_low_complexity.argtypes = [C.c_char_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        self.L = L
        self.h = L.orc_create(similarity, 0, 2000, -1, b"\0")
        assert L.orc_load_reference_fa(self.h, fasta.encode()) > 0
        self.inferred_k = L.orc_infer_kmer_length(self.h)
        self.k = max(9, self.inferred_k) if k is None else k  # FastqExtractor.cpp:409-416
        L.orc_set_extract_params(self.h, self.k, max(hit_len_required, self.k) if k is None else hit_len_required)

    def low_complexity(self, read):
        return bool(self.L.orc_is_low_complexity(read.encode()))

    def has_hit(self, read):
        return bool(self.L.orc_has_hit_in_set(self.h, read.encode()))

    def good(self, read):
        return bool(self.L.orc_is_good_candidate(self.h, read.encode()))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None


def fastx_records(path):
    """(name up to the first blank, sequence) of a FASTA/FASTQ file with one-line records."""
    out = []
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        l = lines[i]
        if l[:1] == "@":
            out.append((l[1:].split()[0], lines[i + 1]))
            i += 4
        elif l[:1] == ">":
            out.append((l[1:].split()[0], lines[i + 1]))
            i += 2
        else:
            i += 1
    return out
REF_BAM_EXTRACT = os.path.join(ROOT, "oracle", "_ref", "bam-extractor")
