"""bam-extractor (BamExtractor.cpp:463-949, run-t1k:350) against the reference's own bam-extractor
(oracle/_ref/bam-extractor: BamExtractor.cpp + the samtools-0.1.19 the reference vendors, built by oracle/Makefile) on synthetic
coordinate-sorted BAM files (tests/bamsynth.py).  Every output file must be identical byte for byte to the reference's -t 1 run.
CPU tests: inputs whose reads all align to the primary assembly need no HasHitInSet, so the host side -- the native BGZF / BAM reader,
the interval walk, the CIGAR spans, the two passes, the tag lookup -- is checked without a GPU.  GPU tests (-m gpu): alternative contigs
and unaligned reads, whose fate is decided by t1k_extract_batch."""
import os
import subprocess

import pytest

import bamsynth
import util

BAMX = os.path.join(util.ROOT, "t1k_amd", "bin", "bam-extractor")


def run_both(tmp, sc, args=(), block=0xff00, env=None):
    util.need(util.REF_BAM_EXTRACT)
    bam, fa = os.path.join(tmp, "in.bam"), os.path.join(tmp, "coord.fa")
    sc.write(bam, block)
    sc.write_fasta(fa)
    outs = []
    for tag, binary in (("ref", util.REF_BAM_EXTRACT), ("gpu", BAMX)):
        o = os.path.join(tmp, tag)
        r = subprocess.run([binary, "-b", bam, "-f", fa, "-o", o] + list(args) + (["-t", "1"] if tag == "ref" else ["-t", "8"]), stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                           env=dict(os.environ, **(env or {})) if tag == "gpu" else None)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        outs.append(o)
    return outs


def same_files(a, b, suffixes):
    total = 0
    for suf in suffixes:
        x, y = open(a + suf, "rb").read(), open(b + suf, "rb").read()
        assert x == y, suf
        total += len(x)
    return total


@pytest.mark.parametrize("seed,barcodes,suffix,block", [(1, False, False, 0xff00), (2, True, False, 3000), (3, False, True, 777)])
def test_paired_bam_of_primary_alignments_vs_reference(built, tmp_path, seed, barcodes, suffix, block):
    """pairs over, beside and far from the gene intervals, spliced and clipped alignments, low-complexity reads, a secondary record,
    half-aligned pairs; barcode / UMI tags behind other tags; read names with /1 /2 and --mateIdSuffixLen; BGZF blocks far smaller than a
    record batch (records straddle blocks)"""
    sc = bamsynth.paired_scenario(seed, with_unaligned=False, with_alt=False, barcodes=barcodes, suffix=suffix)
    args = (["--barcode", "CB", "--UMI", "UB"] if barcodes else []) + (["--mateIdSuffixLen", "2"] if suffix and seed % 2 else [])
    a, b = run_both(str(tmp_path), sc, args, block)
    n = same_files(a, b, ["_1.fq", "_2.fq"] + (["_bc.fa", "_umi.fa"] if barcodes else []))
    assert n > 5000


@pytest.mark.parametrize("seed,barcodes", [(11, False), (12, True)])
def test_single_end_bam_of_primary_alignments_vs_reference(built, tmp_path, seed, barcodes):
    sc = bamsynth.single_scenario(seed, with_unaligned=False, with_alt=False, barcodes=barcodes)
    a, b = run_both(str(tmp_path), sc, ["--barcode", "CB"] if barcodes else [])
    assert same_files(a, b, [".fq"] + (["_bc.fa"] if barcodes else [])) > 3000


def test_bam_extractor_exit_codes_and_bad_input(built, tmp_path):
    r = subprocess.run([BAMX], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "-b STRING" in r.stderr
    r = subprocess.run([BAMX, "-b", "x.bam"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Need to use -f" in r.stderr
    fa = str(tmp_path / "c.fa")
    open(fa, "w").write(">G chr6 10 200 +\nACGTACGTACGTAGCTAGCTAGCTAGCATCGATCGAT\n")
    r = subprocess.run([BAMX, "-f", fa], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Need to use -b" in r.stderr
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(b"this is not a BAM file, but it is long enough to be looked at")
    r = subprocess.run([BAMX, "-f", fa, "-b", bad], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "BGZF" in r.stderr
    r = subprocess.run([BAMX, "--noSuchFlag"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1


def test_damaged_bam_under_address_and_ub_sanitizers(tmp_path):
    """a BAM file is untrusted input: bam-extractor's host side (the BGZF / BAM reader, the tag lookup, the CIGAR spans, both passes) built
    with -fsanitize=address,undefined over 2 400 damaged variants of a paired and a single-end file -- length words, name / CIGAR / sequence
    lengths, contig numbers, flags, tag types, string terminators and random bytes changed inside the inflated stream, the stream cut or a
    piece removed and wrapped in BGZF again, block headers / BC / ISIZE of the container changed (tests/harness/bam_fuzz.cpp; the device
    stage is a stand-in there).  The program may refuse a file or write its outputs; it may not read outside its buffers.  The reader of
    the round before failed this within the first 600 variants of every seed (contig number beyond the header's table, a name or a tag
    string without its terminator, CIGAR lengths that overflow an int)."""
    exe = str(tmp_path / "bam_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-o", exe,
                        os.path.join(util.ROOT, "tests", "harness", "bam_fuzz.cpp")] + [os.path.join(util.ROOT, "t1k_amd", "csrc", "host", f) for f in ("reads.cpp", "refset.cpp", "inflate.cpp")] +
                       ["-lz", "-lpthread", "-ldl"], stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        pytest.skip("no address sanitizer runtime for g++ here: " + r.stderr[-200:])
    cases = [("p", bamsynth.paired_scenario(5, barcodes=True, n=40), 3000, ["--barcode", "CB", "--UMI", "UB"], 1),
             ("s", bamsynth.single_scenario(6, barcodes=True, n=40), 0xff00, ["--barcode", "CB"], 2),
             ("u", bamsynth.paired_scenario(7, suffix=True, n=40), 777, ["-u", "--mateIdSuffixLen", "2"], 3)]
    for tag, sc, block, args, seed in cases:
        bam, fa = str(tmp_path / (tag + ".bam")), str(tmp_path / (tag + ".fa"))
        sc.write(bam, block)
        sc.write_fasta(fa)
        r = subprocess.run([exe, bam, fa, str(tmp_path), "800", str(seed)] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0 and "ERROR" not in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (tag, r.stdout, r.stderr[-3000:])
        n, refused, accepted = (int(x) for x in r.stdout.split()[-3:])
        assert n == 800 and refused + accepted == 800 and refused > 300 and accepted > 40, (tag, r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,barcodes,suffix,env", [(21, False, False, {}), (22, True, False, {"T1K_EXTRACT_CHUNK": "7"}), (23, False, True, {"T1K_EXTRACT_CHUNK": "64"})])
def test_paired_bam_with_alt_contigs_and_unaligned_pairs_vs_reference(built, tmp_path, seed, barcodes, suffix, env):
    """reads on alternative contigs and unaligned pairs (either mate first, low-complexity mates, N-rich mates): kept through
    IsLowComplexity + HasHitInSet on the GPU, written in file order; small GPU batches so that events of several batches interleave"""
    sc = bamsynth.paired_scenario(seed, barcodes=barcodes, suffix=suffix)
    args = ["--barcode", "CB", "--UMI", "UB"] if barcodes else []
    a, b = run_both(str(tmp_path), sc, args, env=env)
    assert same_files(a, b, ["_1.fq", "_2.fq"] + (["_bc.fa", "_umi.fa"] if barcodes else [])) > 8000


@pytest.mark.gpu
def test_paired_bam_with_reads_beyond_320_bases_vs_reference(built, tmp_path):
    """2 x 450 bp records (alternative contigs, unaligned pairs): their candidate test runs in the wide kernel shapes"""
    sc = bamsynth.paired_scenario(51, read_len=450, gene_len=2400)
    a, b = run_both(str(tmp_path), sc, env={"T1K_EXTRACT_CHUNK": "16"})
    assert same_files(a, b, ["_1.fq", "_2.fq"]) > 8000


@pytest.mark.gpu
@pytest.mark.parametrize("seed,barcodes", [(31, False), (32, True)])
def test_single_end_bam_with_alt_contigs_and_unaligned_reads_vs_reference(built, tmp_path, seed, barcodes):
    sc = bamsynth.single_scenario(seed, barcodes=barcodes)
    a, b = run_both(str(tmp_path), sc, ["--barcode", "CB", "--UMI", "UB"] if barcodes else [], env={"T1K_EXTRACT_CHUNK": "16"})
    assert same_files(a, b, [".fq"] + (["_bc.fa", "_umi.fa"] if barcodes else [])) > 5000


@pytest.mark.gpu
def test_abnormal_unaligned_flag_vs_reference(built, tmp_path):
    """-u: unaligned templates are tested read by read and collected by name in the second pass"""
    sc = bamsynth.paired_scenario(41)
    a, b = run_both(str(tmp_path), sc, ["-u"])
    assert same_files(a, b, ["_1.fq", "_2.fq"]) > 8000
