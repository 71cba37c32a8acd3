"""GPU tests (pytest -m gpu) of the candidate-read extraction row: t1k_extract_batch through the C ABI against the oracle's
IsGoodCandidate read by read, and the fastq-extractor executable against the golden fixtures of the reference's own binary
(tests/golden/extract_*).  All results are 0/1 flags and byte strings: bit-exact."""
import os
import random
import subprocess

import numpy as np
import pytest

import util
import t1k_amd
from test_extract_oracle import CASES, XCase, homopolymer_reads, kept_ids

pytestmark = pytest.mark.gpu
XBIN = os.path.join(util.ROOT, "t1k_amd", "bin", "fastq-extractor")


def device_flags(ref, seqs, epf, k, hit_len, sim):
    c = t1k_amd.Context(kmer_length=k, hit_len_required=hit_len, ref_seq_similarity=sim, n_base_code=0)
    try:
        c.ref_upload(ref)
        c.reads_upload(seqs)
        return c.extract(epf)
    finally:
        c.close()


def ref_seqs(path):
    out, cur = [], []
    for l in open(path):
        if l[0] == ">":
            if cur:
                out.append("".join(cur))
            cur = []
        else:
            cur.append(l.strip())
    if cur:
        out.append("".join(cur))
    return out


@pytest.mark.parametrize("name", CASES)
def test_extract_batch_vs_oracle_and_golden(built, tmp_path, name):
    """every fragment's flag == oracle's IsGoodCandidate (first end, else the mate); the kept set == what the reference kept"""
    c = XCase(name, str(tmp_path))
    k, hl, sim = c.meta["kmer_length"], c.meta["hit_len_required"], c.similarity()
    r1 = util.fastx_records(c.r1)
    r2 = util.fastx_records(c.r2) if c.paired else None
    seqs = [s for pair in zip((s for _, s in r1), (s for _, s in r2)) for s in pair] if c.paired else [s for _, s in r1]
    good, st = device_flags(ref_seqs(c.ref), seqs, 2 if c.paired else 1, k, hl, sim)
    orc = util.ExtractOracle(c.ref, similarity=sim, k=k, hit_len_required=hl)
    want = np.array([1 if (orc.good(r1[i][1]) or (c.paired and orc.good(r2[i][1]))) else 0 for i in range(len(r1))], dtype=np.uint8)
    orc.close()
    assert np.array_equal(good, want)
    ids = [n[:-2] if (c.strips_mate_suffix() and n[-2:] in ("/1", "/2")) else n for n, _ in r1]
    assert [ids[i] for i in range(len(ids)) if good[i]] == c.kept
    assert st["read_ends"] >= len(r1) and st["read_ends_chained"] <= st["read_ends_with_hits"] <= st["read_ends"]


@pytest.mark.parametrize("name", CASES)
def test_executable_vs_golden(built, tmp_path, name):
    c = XCase(name, str(tmp_path))
    o = str(tmp_path / "out")
    subprocess.run([XBIN] + c.args() + ["-o", o], check=True, stderr=subprocess.PIPE)
    assert kept_ids(o + ("_1.fq" if c.paired else ".fq")) == c.kept
    if c.paired:
        assert kept_ids(o + "_2.fq") == c.kept
    # chunking of the input must not matter
    subprocess.run([XBIN] + c.args() + ["-o", o + "_c"], check=True, stderr=subprocess.PIPE, env=dict(os.environ, T1K_EXTRACT_CHUNK="97"))
    for suffix in (["_1.fq", "_2.fq"] if c.paired else [".fq"]):
        assert open(o + suffix).read() == open(o + "_c" + suffix).read()


@pytest.mark.parametrize("layout", ["fastq", "fastq_crlf_no_final_newline", "fasta", "fastq_gz", "single_end"])
def test_mapped_input_path_equals_streaming_path(built, tmp_path, layout):
    """ordinary read files go through the mapped input (indexed in place by the host threads, batches alternating between two device
    contexts that share the index, kept records formatted from the mapping); the streaming loop (T1K_EXTRACT_STREAM=1: reader threads,
    owned chunks) is what every other input takes.  Same files byte for byte -- with -t 1 (ids lose /1 /2) and -t 4 (raw names), read
    windows, many small batches -- and, where the reference binary is here, the same as its own output."""
    ref = str(tmp_path / "ref.fa")
    util.synth_ref("ref-rna", ref, seed=51, genes=6, scale=0.2)
    pfx = str(tmp_path / "r")
    util.synth_reads(ref, pfx, seed=52, pairs=6000, len=150, bg=0.5, sub=0.04, nrate=0.01, **({"fasta": 1} if layout == "fasta" else {}))
    ext = "fa" if layout == "fasta" else "fq"
    f1, f2 = pfx + "_1." + ext, pfx + "_2." + ext
    if layout == "fastq_crlf_no_final_newline":
        for f in (f1, f2):
            text = open(f, "rb").read()
            open(f, "wb").write(text.replace(b"\n", b"\r\n").rstrip(b"\r\n"))
    if layout == "fastq_gz":
        import gzip
        for f in (f1, f2):
            open(f + ".gz", "wb").write(gzip.compress(open(f, "rb").read(), 1))
        f1, f2 = f1 + ".gz", f2 + ".gz"
    files = ["-u", f1] if layout == "single_end" else ["-1", f1, "-2", f2]
    for extra in (["-t", "1"], ["-t", "4", "--read1Start", "3", "--read1End", "120", "--read2Start", "0", "--read2End", "99"]):
        outs = {}
        for mode, env in (("mapped", {"T1K_EXTRACT_CHUNK": "700"}), ("mapped_one_batch", {}), ("stream", {"T1K_EXTRACT_STREAM": "1"})):
            o = str(tmp_path / (mode + "_".join(extra)))
            r = subprocess.run([XBIN, "-f", ref] + files + extra + ["-o", o], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_DEBUG_PHASES="1", **env))
            assert r.returncode == 0, r.stderr[-1500:]
            assert ("mapped input" in r.stderr) == (mode != "stream"), r.stderr[-1500:]
            outs[mode] = [open(o + sfx, "rb").read() for sfx in ([".fq"] if layout == "single_end" else ["_1.fq", "_2.fq"])]
            assert len(outs[mode][0]) > 1000
        assert outs["mapped"] == outs["stream"] and outs["mapped_one_batch"] == outs["stream"]
        if os.path.exists(util.REF_EXTRACT) and not (layout == "single_end" and "--read2End" in extra):
            o = str(tmp_path / ("ref" + "_".join(extra)))
            subprocess.run([util.REF_EXTRACT, "-f", ref] + files + extra + ["-o", o], check=True, stderr=subprocess.PIPE)
            assert [open(o + sfx, "rb").read() for sfx in ([".fq"] if layout == "single_end" else ["_1.fq", "_2.fq"])] == outs["mapped"]


def test_edge_reads_vs_oracle(built, tmp_path):
    """empty / shorter-than-k / all-N / homopolymer / repeat / exact-copy / reverse-complement / boundary-similarity reads"""
    ref_fa = util.gunzip_to(util.CYP_DNA, str(tmp_path / "ref.fa"))
    ref = ref_seqs(ref_fa)
    rng = random.Random(7)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    a0 = ref[0].replace("N", "A")
    reads = ["", "ACGTACG", "N" * 100, "A" * 100, "AC" * 50, "ACG" * 40, a0[100:250], "".join(comp[c] for c in reversed(a0[300:450])), a0[500:511], a0[500:512]]
    for _ in range(300):  # windows of alleles with a growing number of substitutions / N / one indel
        al = rng.choice(ref)
        L = rng.choice([36, 75, 100, 150, 250, 320])
        p = rng.randrange(0, max(1, len(al) - L))
        s = list(al[p:p + L])
        for _ in range(rng.choice([0, 1, 3, 8, 15, 30, 60])):
            q = rng.randrange(len(s))
            s[q] = rng.choice("ACGTN")
        if rng.random() < 0.3 and len(s) > 20:
            q = rng.randrange(5, len(s) - 5)
            s[q:q + rng.choice([1, 2, 7])] = []
        if rng.random() < 0.5:
            s = [comp[c] for c in reversed(s)]
        reads.append("".join(s))
    for k, hl, sim in ((12, 27, 0.8), (12, 30, 0.97), (9, 23, 0.8), (14, 40, 0.9)):
        good, _ = device_flags(ref, reads, 1, k, hl, sim)
        orc = util.ExtractOracle(ref_fa, similarity=sim, k=k, hit_len_required=hl)
        want = np.array([1 if orc.good(r) else 0 for r in reads], dtype=np.uint8)
        orc.close()
        assert np.array_equal(good, want), (k, hl, sim, np.nonzero(good != want)[0][:10])
        assert 0 < want.sum() < len(reads)


def test_full_size_properties(built, tmp_path):
    """200k synthetic pairs (too many for the pure-CPU oracle in a test): extraction is idempotent (running it on its own output keeps
    everything), keeps input order, keeps pairs together, and agrees with the oracle on a random sample of fragments."""
    ref = str(tmp_path / "ref.fa")
    util.synth_ref("ref-rna", ref, seed=41, genes=8, scale=0.3)
    pfx = str(tmp_path / "r")
    util.synth_reads(ref, pfx, seed=42, pairs=200000, len=150, bg=0.6, sub=0.05, indel=0.003, nrate=0.01)
    o = str(tmp_path / "x")
    p = subprocess.run([XBIN, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "-o", o], check=True, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, T1K_DEBUG_PHASES="1"))
    k1, k2 = kept_ids(o + "_1.fq"), kept_ids(o + "_2.fq")
    assert k1 == k2 and 50000 < len(k1) < 100000
    order = [int(n[1:]) for n in k1]
    assert order == sorted(order)
    subprocess.run([XBIN, "-f", ref, "-1", o + "_1.fq", "-2", o + "_2.fq", "-o", o + "2"], check=True, stderr=subprocess.PIPE)
    assert open(o + "_1.fq").read() == open(o + "2_1.fq").read() and open(o + "_2.fq").read() == open(o + "2_2.fq").read()
    import re
    m = re.search(r"k=(\d+) hitLenRequired=(\d+)", p.stderr)
    orc = util.ExtractOracle(ref, k=int(m.group(1)), hit_len_required=int(m.group(2)))
    r1, r2 = util.fastx_records(pfx + "_1.fq"), util.fastx_records(pfx + "_2.fq")
    kept = set(k1)
    rng = random.Random(3)
    for i in rng.sample(range(len(r1)), 600):
        assert (orc.good(r1[i][1]) or orc.good(r2[i][1])) == (r1[i][0][:-2] in kept)
    orc.close()


def test_large_bucket_takes_the_big_shape(built, tmp_path):
    """a trinucleotide repeat puts thousands of hits into one (strand, sequence) bucket: more than the production LDS shape holds (1024),
    so the batch is run again in the large shape (8192), and beyond that with the hit arrays in HBM (65 535); results still equal the oracle's."""
    rng = random.Random(11)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    ref = [rnd(400) + "CAG" * 300 + rnd(400), rnd(1500), rnd(300) + "CAG" * 40 + rnd(300)]
    fa = tmp_path / "rep.fa"
    fa.write_text("".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(ref)))
    reads = ["CAG" * 50, "AGC" * 50, "CTG" * 50, ref[0][330:480], ref[0][1250:1400], ref[2][280:430], rnd(150), ("CAG" * 50)[:100] + rnd(50)]
    for k, hl in ((12, 30), (9, 27)):
        orc = util.ExtractOracle(str(fa), k=k, hit_len_required=hl)
        want = np.array([1 if orc.good(r) else 0 for r in reads], dtype=np.uint8)
        orc.close()
        good, st = device_flags(ref, reads, 1, k, hl, 0.8)
        assert np.array_equal(good, want), (k, good, want)
        assert want[0] == 1 and st["big_shape"] == 1
    os.environ["T1K_EXTRACT_FORCE_BIG"] = "1"
    try:
        good2, _ = device_flags(ref, reads, 1, 9, 27, 0.8)
    finally:
        del os.environ["T1K_EXTRACT_FORCE_BIG"]
    assert np.array_equal(good2, want)
    # beyond the large LDS shape (8192 hits a bucket): third attempt with the hit arrays in HBM, still the oracle's answers
    huge = [rnd(200) + "CAG" * 900 + rnd(200), rnd(800)]
    fa2 = tmp_path / "huge.fa"
    fa2.write_text("".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(huge)))
    hreads = ["CAG" * 50, huge[0][150:300], huge[1][100:250], rnd(150), "CTG" * 50]
    orc = util.ExtractOracle(str(fa2), k=9, hit_len_required=27)
    hwant = np.array([1 if orc.good(r) else 0 for r in hreads], dtype=np.uint8)
    orc.close()
    hgood, hst = device_flags(huge, hreads, 1, 9, 27, 0.8)
    assert np.array_equal(hgood, hwant), (hgood, hwant)
    assert hst["big_shape"] == 2, hst
    os.environ["T1K_EXTRACT_FORCE_HUGE"] = "1"
    try:
        good3, st3 = device_flags(ref, reads, 1, 9, 27, 0.8)
    finally:
        del os.environ["T1K_EXTRACT_FORCE_HUGE"]
    assert np.array_equal(good3, want) and st3["big_shape"] == 2
    # 65 535 hits a bucket is the end (16-bit links of the chain): loudly
    with pytest.raises(t1k_amd.T1kError, match="more than 65 535 hits"):
        device_flags([rnd(200) + "CAG" * 12000 + rnd(200)], ["CAG" * 50], 1, 9, 27, 0.8)


def test_n_next_to_homopolymers_vs_oracle(built, tmp_path):
    """k-mer windows holding an N next to A / T runs (see test_extract_oracle.homopolymer_reads): index build and look-up rule with N -> 0"""
    ref_fa, reads = homopolymer_reads(tmp_path, 9)
    ref = ref_seqs(ref_fa)
    for k, hl, sim in ((9, 23, 0.8), (9, 27, 0.985), (9, 23, 0.992), (11, 23, 0.999), (11, 40, 0.8)):
        orc = util.ExtractOracle(ref_fa, similarity=sim, k=k, hit_len_required=hl)
        want = np.array([1 if orc.good(r) else 0 for r in reads], dtype=np.uint8)
        orc.close()
        good, _ = device_flags(ref, reads, 1, k, hl, sim)
        assert np.array_equal(good, want), (k, hl, sim, np.nonzero(good != want)[0][:10])


def test_reads_beyond_320_bases_vs_oracle_and_reference_binary(built, tmp_path):
    """a batch that holds reads of more than 320 bases takes the wide shapes of k_extract_screen / k_extract (16 positions a lane, 8 a
    thread, 4 lists a thread): flags against the oracle read by read -- windows of alleles of 100 .. 1000 bases with substitutions, N,
    indels, both strands, self-repeating reads, random reads -- and the executable on a 2 x 150 bp file with 2 x 500 / 2 x 900 bp pairs
    among them against the reference's own fastq-extractor, byte for byte"""
    ref_fa = util.gunzip_to(util.CYP_DNA, str(tmp_path / "ref.fa"))
    ref = ref_seqs(ref_fa)
    rng = random.Random(11)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    reads = [rnd(400), rnd(1000), "A" * 700, "ACGT" * 200]
    for _ in range(260):
        al = rng.choice(ref)
        L = rng.choice([100, 150, 321, 400, 512, 640, 777, 1000])
        p0 = rng.randrange(0, max(1, len(al) - L))
        sq = list(al[p0:p0 + L])
        for _ in range(rng.choice([0, 2, 8, 30, 80, 200])):
            q = rng.randrange(len(sq))
            sq[q] = rng.choice("ACGTN")
        if rng.random() < 0.3 and len(sq) > 40:
            q = rng.randrange(5, len(sq) - 20)
            sq[q:q + rng.choice([1, 2, 7, 11])] = []
        if rng.random() < 0.2:
            sq = sq[:len(sq) // 2] * 2
        if rng.random() < 0.5:
            sq = [comp[c] for c in reversed(sq)]
        reads.append("".join(sq)[:1000])
    for k, hl, sim in ((12, 27, 0.8), (12, 30, 0.97), (9, 23, 0.8), (14, 40, 0.9)):
        good, _ = device_flags(ref, reads, 1, k, hl, sim)
        orc = util.ExtractOracle(ref_fa, similarity=sim, k=k, hit_len_required=hl)
        want = np.array([1 if orc.good(r) else 0 for r in reads], dtype=np.uint8)
        orc.close()
        assert np.array_equal(good, want), (k, hl, sim, np.nonzero(good != want)[0][:10])
        assert 0 < int(want.sum()) < len(reads)
    good2, _ = device_flags(ref, reads[:len(reads) // 2 * 2], 2, 12, 27, 0.8)   # paired: the mate is asked when the first end fails
    orc = util.ExtractOracle(ref_fa, similarity=0.8, k=12, hit_len_required=27)
    want2 = np.array([1 if (orc.good(reads[2 * i]) or orc.good(reads[2 * i + 1])) else 0 for i in range(len(reads) // 2)], dtype=np.uint8)
    orc.close()
    assert np.array_equal(good2, want2)
    # the executable against the reference's
    util.need(util.REF_EXTRACT)
    rna = util.gunzip_to(util.CYP_RNA, str(tmp_path / "rna.fa"))
    util.synth_reads(rna, str(tmp_path / "s"), pairs=3000, len=150, seed=31, bg=0.5)
    util.synth_reads(rna, str(tmp_path / "l"), pairs=40, len=500, seed=32, bg=0.3, fragmean=1040)
    util.synth_reads(rna, str(tmp_path / "x"), pairs=10, len=900, seed=33, bg=0.3, fragmean=1400)
    for m in (1, 2):
        sh = open(str(tmp_path / ("s_%d.fq" % m))).read().split("\n")
        lo = open(str(tmp_path / ("l_%d.fq" % m))).read().split("\n")[:160]
        xl = open(str(tmp_path / ("x_%d.fq" % m))).read().split("\n")[:40]
        open(str(tmp_path / ("mix_%d.fq" % m)), "w").write("\n".join(sh[:4000] + lo + sh[4000:8000] + xl + sh[8000:]))
    args = ["-f", rna, "-1", str(tmp_path / "mix_1.fq"), "-2", str(tmp_path / "mix_2.fq"), "-t", "8"]  # (-t also decides whether /1 /2 stay in the ids)
    subprocess.run([util.REF_EXTRACT] + args + ["-o", str(tmp_path / "r")], check=True, stderr=subprocess.DEVNULL)
    for tag, env in (("o", {}), ("c", {"T1K_EXTRACT_CHUNK": "97"})):
        subprocess.run([XBIN] + args + ["-o", str(tmp_path / tag)], check=True, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        for suffix in ("_1.fq", "_2.fq"):
            assert open(str(tmp_path / "r") + suffix).read() == open(str(tmp_path / tag) + suffix).read(), (tag, suffix)
    assert len(kept_ids(str(tmp_path / "o_1.fq"))) > 100


def test_barcode_whitelist_vs_reference_binary(built, tmp_path):
    """--barcode / --barcodeWhitelist (BarcodeCorrector.hpp): exact, corrected, ambiguous (quality tie-break), uncorrectable and N barcodes,
    plain and with a reverse-complemented sub-range; the _bc.fa files against the reference's own fastq-extractor"""
    util.need(util.REF_EXTRACT)  # decided when the test runs, after the `built` fixture had its chance to build oracle/_ref
    c = XCase("cyp_rna_2x100", str(tmp_path))
    rng = random.Random(21)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    L = 16
    wl = [rnd(L) for _ in range(40)]
    wl += [b[:5] + ("A" if b[5] != "A" else "C") + b[6:] for b in wl[:15]]          # neighbours: corrections can be ambiguous
    (tmp_path / "wl.txt").write_text("\n".join(wl) + "\n")
    (tmp_path / "wl_rc.txt").write_text("\n".join("".join(comp[x] for x in reversed(b[2:14])) for b in wl) + "\n")
    n = len(util.fastx_records(c.r1))
    with open(str(tmp_path / "bc.fq"), "w") as f:
        for i in range(n):
            b = list(rng.choice(wl)) if rng.random() < 0.9 else list(rnd(L))
            u = rng.random()
            for _ in range(1 if u < 0.35 else 2 if u < 0.45 else 0):
                b[rng.randrange(L)] = rng.choice("ACGTN")
            f.write("@b%d\n%s\n+\n%s\n" % (i, "".join(b), "".join(chr(rng.randrange(35, 74)) for _ in range(L))))
    for name, extra in (("plain", ["--barcodeWhitelist", str(tmp_path / "wl.txt")]),
                        ("rc", ["--barcodeWhitelist", str(tmp_path / "wl_rc.txt"), "--barcodeStart", "2", "--barcodeEnd", "13", "--barcodeRevComp", "-t", "3"])):
        args = c.args() + ["--barcode", str(tmp_path / "bc.fq")] + extra
        subprocess.run([XBIN] + args + ["-o", str(tmp_path / ("o_" + name))], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([util.REF_EXTRACT] + args + ["-o", str(tmp_path / ("r_" + name))], check=True, stderr=subprocess.DEVNULL)
        a, b = open(str(tmp_path / ("r_" + name)) + "_bc.fa").read(), open(str(tmp_path / ("o_" + name)) + "_bc.fa").read()
        assert a == b
        assert 0 < a.count("missing_barcode") < a.count(">")
