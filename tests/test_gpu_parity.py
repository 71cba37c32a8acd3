"""GPU tests (pytest -m gpu): the HIP path, called through the C ABI (ctypes) and through the drop-in executable, against
(1) the committed golden fixtures produced by the reference itself, (2) the oracle, (3) the reference binary live.
Integer / byte / index results must be bit-exact; doubles (similarities, EM) are compared bit-for-bit as well because
the device arithmetic follows the reference's operation order (tolerance stated where it is not)."""
import gzip
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import goldens
import util
import t1k_amd

pytestmark = pytest.mark.gpu
GENO = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")


@pytest.fixture(scope="module")
def ctx(built):
    c = t1k_amd.Context()
    yield c
    c.close()


def ga_vectors():
    out = []
    with gzip.open(os.path.join(util.GOLDEN, "ga_vectors.tsv.gz"), "rt") as f:
        for line in f:
            t, p, score, ops = line.rstrip("\n").split("\t")
            out.append((t, p, int(score), "" if ops == "-" else ops))
    return out


def test_global_alignment_kernel_vs_reference_vectors(ctx):
    """t1k_align_batch (general banded DP + traceback on the device) vs the reference's own GlobalAlignment outputs."""
    vec = ga_vectors()
    score, nm, nx, ni, ops = ctx.align_batch([v[0] for v in vec], [v[1] for v in vec])
    for i, (t, p, s, o) in enumerate(vec):
        assert score[i] == s, (t, p)
        assert "".join(str(int(x)) for x in ops[i]) == o, (t, p)
        assert (nm[i], nx[i], ni[i]) == (o.count("0"), o.count("1"), o.count("2") + o.count("3"))


def test_production_match_count_vs_reference_vectors(ctx):
    """the production routine (<=3-mismatch popcount fast path, else register-resident banded forward sweep) must return
    the number of MATCH columns of the reference's traceback for every equal-length vector."""
    vec = [v for v in ga_vectors() if len(v[0]) == len(v[1])]
    assert len(vec) > 600
    got = ctx.align_count_batch([v[0] for v in vec], [v[1] for v in vec])
    for g, (t, p, s, o) in zip(got, vec):
        assert g == o.count("0"), (t, p)


def test_assign_stage_vs_oracle(built):
    import gpu_assign_check
    assert gpu_assign_check.main() == 0


@pytest.mark.parametrize("name", goldens.CASES)
def test_executable_vs_golden_reference_outputs(built, tmp_path, name):
    """the drop-in genotyper must reproduce the reference's files byte for byte on the committed inputs"""
    c = goldens.Case(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "gpu")
    r = subprocess.run([GENO] + c.args() + ["-o", out, "--outputReadAssignment"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out + "_genotype.tsv").read() == c.expected("genotype.tsv")
    assert open(out + "_allele.tsv").read() == c.expected("allele.tsv")
    assert open(out + "_assign.tsv").read() == c.expected("assign.tsv.gz")
    ids = [l[1:].strip() for l in open(out + ("_aligned_1.fa" if c.paired else "_aligned.fa")) if l.startswith(">")]
    assert ids == c.expected("aligned_ids.txt.gz").split()
    if c.paired:
        ids2 = [l[1:].strip() for l in open(out + "_aligned_2.fa") if l.startswith(">")]
        assert ids2 == ids
    if c.bc:
        assert open(out + "_aligned_bc.fa").read() == c.expected("aligned_bc.fa")
    m = re.search(r"in (\d+) EM iterations", r.stderr)
    assert int(m.group(1)) == c.meta["em_iterations"]
    m = re.search(r"(\d+) read fragments can be assigned \(average ([-\d.naninf]+) alleles/read\)", r.stderr)
    assert int(m.group(1)) == c.meta["assigned_fragments"] and m.group(2) == c.meta["avg_alleles"]


@pytest.mark.parametrize("name", ["hla_synth_2x150", "cyp_dna_relax_2x150", "cyp_rna_2x100"])
def test_linear_reference_windows_give_the_same_files(built, tmp_path, name):
    """round 6: the closed-form pass of the chain reads the allele windows from a transposed copy of the reference (T1kRefDev::basesT); with
    T1K_REF_TRANSPOSE=0 it reads the linear layout as every other kernel does -- both against the reference's committed files
    (SeqSet.hpp:1697-1848 through the assignment table)"""
    c = goldens.Case(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "lin")
    r = subprocess.run([GENO] + c.args() + ["-o", out, "--outputReadAssignment"], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_REF_TRANSPOSE="0"))
    assert r.returncode == 0, r.stderr
    assert open(out + "_genotype.tsv").read() == c.expected("genotype.tsv")
    assert open(out + "_allele.tsv").read() == c.expected("allele.tsv")
    assert open(out + "_assign.tsv").read() == c.expected("assign.tsv.gz")


def test_executable_vs_live_reference_binary(built):
    util.need(util.REF_BIN)  # decided when the test runs, after the `built` fixture had its chance to build oracle/_ref
    import gpu_e2e_check
    assert gpu_e2e_check.main() == 0


def test_job_api_batching_and_rerun_invariance(built, tmp_path):
    """size-independent properties: results do not depend on the device batch size, and a job can be re-run (coverage and
    group state are reset) with identical output."""
    c = goldens.Case("hla_synth_2x150", str(tmp_path))
    r1 = [s for _, _, s in t1k_amd.read_fastx(c.r1)]
    r2 = [s for _, _, s in t1k_amd.read_fastx(c.r2)]
    texts = []
    for bf in (0, 37, 128):
        job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97, batch_fragments=bf)
        job.set_reads(r1, r2)
        job.run()
        a = job.genotype_text()
        job.run()
        assert job.genotype_text() == a
        texts.append((a, job.counts()))
        job.close()
    os.environ["T1K_PIPELINES"] = "1"  # one pipeline per GPU instead of the default three: same absorption order, same result
    try:
        job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97, batch_fragments=37)
        job.set_reads(r1, r2)
        job.run()
        texts.append((job.genotype_text(), job.counts()))
        job.close()
    finally:
        del os.environ["T1K_PIPELINES"]
    assert texts[0] == texts[1] == texts[2] == texts[3]
    # without barcodes no fragment is dropped, so only the gene lines' abundances may differ from the golden (barcode) run;
    # the calls themselves must agree
    exp = [l.split("\t")[2] for l in c.expected("genotype.tsv").splitlines()]
    got = [l.split("\t")[2] for l in texts[0][0].splitlines()]
    assert exp == got


@pytest.mark.parametrize("G,E,long_classes", [(3000, 257, False), (20000, 37, True)])
def test_em_update_bit_exact(ctx, G, E, long_classes):
    """t1k_em_update vs a sequential numpy restatement of Genotyper::EMupdate (same order of double operations).  The second shape has
    classes of thousands of entries (the class pass takes them 512 at a time, 64 to an ordered piece), one of exactly 1024, and an empty one."""
    rng = np.random.default_rng(5)
    if long_classes:
        rows = [rng.choice(E - 2, size=rng.integers(1, 30), replace=False) for _ in range(G)]
        for g in range(100, 100 + 1024):
            rows[g] = np.append(rows[g], E - 1)  # class E-1: exactly 1024 entries; class E-2: none
    else:
        rows = [rng.choice(E, size=rng.integers(1, 40), replace=False) for _ in range(G)]
    row_ptr = np.zeros(G + 1, np.uint64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    ec_idx = np.concatenate(rows).astype(np.uint32)
    count = rng.choice([1.0, 0.5, 0.1, 37.5, 1200.25], size=G)
    ec_len = rng.integers(900, 1300, size=E).astype(np.int32)
    x0 = rng.random(E) * 3
    x0[rng.integers(0, E, 20)] = 0.0
    ctx.em_setup(row_ptr, ec_idx, count, ec_len)
    x1, n, diff = ctx.em_update(x0)
    n_ref = np.zeros(E)
    for g, r in enumerate(rows):
        psum = 0.0
        for e in r:
            psum += x0[e]
        if psum == 0:
            psum = 1
        for e in r:
            n_ref[e] += count[g] * (x0[e] / psum)
    norm = 0.0
    for i in range(E):
        norm += n_ref[i] / ec_len[i]
    x_ref = np.array([n_ref[i] / ec_len[i] / norm for i in range(E)])
    d_ref = 0.0
    for i in range(E):
        d_ref += abs(x_ref[i] - x0[i])
    assert np.array_equal(n, n_ref) and np.array_equal(x1, x_ref) and diff == d_ref


def test_ragged_and_degenerate_reads(ctx, tmp_path):
    """empty batch, reads shorter than k, all-N reads, reads with no hit: no overlaps, no crash; mixed lengths in one batch"""
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    names, seqs, masks, _ = t1k_amd.load_reference_fasta(ref)
    ctx.ref_upload(seqs, masks)
    ctx.reads_upload([])
    ctx.assign()
    counts, ovl = ctx.overlaps()
    assert len(counts) == 0 and len(ovl) == 0
    reads = ["ACGT", "N" * 80, "ACGTTGCA" * 12, seqs[0][100:175], seqs[3][10:310], "A" * 150, seqs[5][0:11]]
    ctx.reads_upload(reads)
    ctx.assign()
    counts, ovl = ctx.overlaps()
    orc = util.Oracle(ref)
    pos = 0
    for i, r in enumerate(reads):
        o, s = orc.assign_read(r)
        assert len(o) == counts[i]
        g = ovl[pos:pos + counts[i]]
        pos += counts[i]
        if len(o):
            assert np.array_equal(o[:, 0], g["seq_idx"]) and np.array_equal(o[:, 6], g["match_cnt"]) and np.array_equal(s, g["similarity"])
    assert counts[0] == 0 and counts[1] == 0 and counts[3] > 0 and counts[4] > 0


def test_missing_coverage_kernel_vs_host(ctx, tmp_path):
    """t1k_missing_coverage (prefix sum + radix-selection median on the device) against the same statistic computed from the
    downloaded per-base coverage the way SeqSet::GetSeqMissingBaseCoverage does (sort, element size/2, cutoff, count)"""
    ref = str(tmp_path / "hla.fa")
    util.synth_ref("ref-dna", ref, genes=3, scale=0.03)
    names, seqs, masks, _ = t1k_amd.load_reference_fasta(ref)
    ctx.ref_upload(seqs, masks)
    util.synth_reads(ref, str(tmp_path / "r"), pairs=400, len=150, seed=4)
    reads = [s for _, _, s in t1k_amd.read_fastx(str(tmp_path / "r_1.fq"))] + [s for _, _, s in t1k_amd.read_fastx(str(tmp_path / "r_2.fq"))]
    ctx.reads_upload(reads)
    ctx.coverage_reset()
    ctx.assign()
    cov = ctx.coverage()
    got = ctx.missing_coverage()
    off = 0
    exp = np.zeros(len(seqs), dtype=np.int32)
    for a, (sq, m) in enumerate(zip(seqs, masks)):
        ex = np.sort(cov[off:off + len(sq)][np.asarray(m, dtype=bool)])
        off += len(sq)
        if len(ex):
            cutoff = max(1.0, ex[len(ex) // 2] * 0.01)
            exp[a] = int(np.sum(~(ex >= cutoff)))
    assert cov.sum() > 0 and np.array_equal(got, exp)


def homopolymer_case(tmp_path, k):
    """a reference with A / T runs longer than k and reads that put an N right before, right after or inside them, both strands:
    the k-mer code of a window holding an N decides whether its neighbour counts as a repeat of the previous k-mer"""
    import random
    rng = random.Random(5)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    body = rnd(260) + "A" * (k + 6) + rnd(240) + "T" * (k + 5) + rnd(260) + "A" * (k + 1) + "C" + "T" * (k + 2) + rnd(200)
    alt = body[:100] + ("A" if body[100] != "A" else "C") + body[101:]
    fa = tmp_path / "homo.fa"
    fa.write_text(">X*01:01 1 0 %d\n%s\n>X*01:02 1 0 %d\n%s\n" % (len(body) - 1, body, len(alt) - 1, alt))
    reads = []
    runs = [(260, k + 6), (260 + k + 6 + 240, k + 5), (260 + k + 6 + 240 + k + 5 + 260, k + 1), (260 + k + 6 + 240 + k + 5 + 260 + k + 2, k + 2)]
    for start, ln in runs:
        for npos in (start - 2, start - 1, start, start + 1, start + ln - 1, start + ln, start + ln + 1):
            for w0 in (start - 70, start - 20, start - 120):
                s = list(body[w0:w0 + 150])
                if 0 <= npos - w0 < len(s):
                    s[npos - w0] = "N"
                reads.append("".join(s))
                reads.append("".join(comp[c] for c in reversed(s)))
    return str(fa), reads


def test_n_next_to_homopolymer_vs_oracle(built, tmp_path):
    import gpu_assign_check
    fa, reads = homopolymer_case(tmp_path, 11)
    assert gpu_assign_check.compare(fa, reads, 0.8, False, "N next to homopolymers") == 0


@pytest.mark.parametrize("name", ["hla_synth_2x150", "cyp_dna_relax_2x150", "kir_synth_relax_2x150", "cyp_rna_single"])
@pytest.mark.parametrize("gpus", ["0,0", "0,0,0", "0,0+own-input", "0,0,0+own-input"])
def test_sharded_job_equals_single_gpu_job(built, tmp_path, name, gpus):
    """The multi-GPU path with several ranks on ONE device (T1K_GPUS=0,0: one thread, job and context set per rank, in-process
    transport): fragments sharded by contiguous slices, coverage all-reduce, row exchange to the pattern owners + coalescing + group
    gather, EM with a sharded row pass.  Every output file must equal the reference's (= the single-GPU run's) byte for byte.
    "+own-input": every rank indexes only its own fragments of the read files (newline counts per MiB block exchanged, host/reads.cpp
    openSharded) and writes only its own part of the *_aligned*.fa files -- what ranks in separate processes do."""
    c = goldens.Case(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "sharded")
    env = dict(os.environ, T1K_GPUS=gpus.split("+")[0])
    if "+" in gpus:
        env["T1K_SHARD_INPUT"] = "1"
    r = subprocess.run([GENO] + c.args() + ["-o", out, "--outputReadAssignment"], stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert open(out + "_genotype.tsv").read() == c.expected("genotype.tsv")
    assert open(out + "_allele.tsv").read() == c.expected("allele.tsv")
    assert open(out + "_assign.tsv").read() == c.expected("assign.tsv.gz")   # the ranks' tables, gathered in rank order
    ids = [l[1:].strip() for l in open(out + ("_aligned_1.fa" if c.paired else "_aligned.fa")) if l.startswith(">")]
    assert ids == c.expected("aligned_ids.txt.gz").split()
    if c.paired:
        ids2 = [l[1:].strip() for l in open(out + "_aligned_2.fa") if l.startswith(">")]
        assert ids2 == ids
    if "+" in gpus:  # the read files written in parts by the ranks: byte for byte those of a single-GPU run
        one = os.path.join(str(tmp_path), "one")
        r1 = subprocess.run([GENO] + c.args() + ["-o", one], stderr=subprocess.PIPE, text=True)
        assert r1.returncode == 0, r1.stderr
        for suf in (("_aligned_1.fa", "_aligned_2.fa") if c.paired else ("_aligned.fa",)):
            assert open(out + suf, "rb").read() == open(one + suf, "rb").read(), suf
    if c.bc:
        assert open(out + "_aligned_bc.fa").read() == c.expected("aligned_bc.fa")
    m = re.search(r"in (\d+) EM iterations", r.stderr)
    assert int(m.group(1)) == c.meta["em_iterations"]
    m = re.search(r"(\d+) read fragments can be assigned \(average ([-\d.naninf]+) alleles/read\)", r.stderr)
    assert int(m.group(1)) == c.meta["assigned_fragments"] and m.group(2) == c.meta["avg_alleles"]


@pytest.mark.parametrize("name", ["hla_synth_2x150", "cyp_dna_relax_2x150", "kir_synth_relax_2x150"])
@pytest.mark.parametrize("gpus", ["0,0", "0,0,0"])
def test_em_allreduce_collective_within_tolerance_of_exact_mode(built, tmp_path, name, gpus):
    """T1K_EM_COLLECTIVE=allreduce (opt-in): every rank sums its own read groups' contributions per class and the E partial sums are
    all-reduced (the collective BASELINE.json's north_star names: E doubles per EM update, Genotyper.hpp:372-421) instead of the
    bit-exact gather of the nnz-sized contribution array.  The re-associated sums may move low bits, so the stated tolerance applies:
    identical allele calls and qualities, abundances within 1e-4 relative of the reference's (= the exact mode's) golden output,
    the same number of EM iterations."""
    c = goldens.Case(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "reduce")
    env = dict(os.environ, T1K_GPUS=gpus, T1K_EM_COLLECTIVE="allreduce")
    r = subprocess.run([GENO] + c.args() + ["-o", out], stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr

    def close(a, b):
        if len(a) != len(b):
            return False
        for x, y in zip(a, b):
            if x == y:
                continue
            try:
                fx, fy = float(x), float(y)
            except ValueError:
                return False
            if abs(fx - fy) > 1e-4 * max(abs(fx), abs(fy)):   # north_star: abundances within 1e-4 relative
                return False
        return True

    for suf, key in (("_genotype.tsv", "genotype.tsv"), ("_allele.tsv", "allele.tsv")):
        got, want = open(out + suf).read().splitlines(), c.expected(key).splitlines()
        assert len(got) == len(want), suf
        for g, w in zip(got, want):
            assert close(g.split("\t"), w.split("\t")), (suf, g, w)
    m = re.search(r"in (\d+) EM iterations", r.stderr)
    assert int(m.group(1)) == c.meta["em_iterations"]


def test_rccl_single_rank_communicator(built, tmp_path):
    """the RCCL transport itself (librccl bound lazily, ncclCommInitRank, all-reduce / send-recv / broadcast groups) with one rank:
    the job must run through every collective and give the single-GPU result"""
    c = goldens.Case("hla_synth_2x150", str(tmp_path))
    ref_text = None
    for use_comm in (False, True):
        job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97)
        job.load_reads(c.r1, c.r2, c.bc)
        comm = None
        if use_comm:
            comm = t1k_amd.Comm(job, 1, 0, unique_id=t1k_amd.comm_unique_id())
            assert comm.is_rccl()
            job.set_shard(0, 1, comm)
        job.run()
        text = job.genotype_text()
        job.close()
        if comm:
            comm.close()
        if ref_text is None:
            ref_text = text
        assert text == ref_text == c.expected("genotype.tsv")


def _golden_files_equal(c, out):
    assert open(out + "_genotype.tsv").read() == c.expected("genotype.tsv")
    assert open(out + "_allele.tsv").read() == c.expected("allele.tsv")
    ids = [l[1:].strip() for l in open(out + ("_aligned_1.fa" if c.paired else "_aligned.fa")) if l.startswith(">")]
    assert ids == c.expected("aligned_ids.txt.gz").split()


def test_gzip_read_files(built, tmp_path):
    """reads straight from .gz files (inflated once into memory, then indexed in place like a mapped file): the committed fixtures are
    stored gzipped, so they are the input as they lie"""
    c = goldens.Case("cyp_dna_relax_2x150", str(tmp_path))
    out = os.path.join(str(tmp_path), "gz")
    args = ["-f", c.ref, "-1", os.path.join(c.dir, "reads_1.fq.gz"), "-2", os.path.join(c.dir, "reads_2.fq.gz")] + c.flags
    r = subprocess.run([GENO] + args + ["-o", out], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    _golden_files_equal(c, out)


@pytest.mark.parametrize("case,env", [("cyp_dna_relax_2x150", {}), ("cyp_rna_2x100", {"T1K_FIRST_WINDOW": "64", "T1K_WINDOW": "256", "T1K_BATCH": "32", "T1K_PAIR_BATCH": "64"}),
                                      ("cyp_dna_relax_2x150", {"T1K_COVERAGE": "eager", "T1K_PIPELINES": "1"}),
                                      ("cyp_rna_single", {"T1K_FIRST_WINDOW": "32", "T1K_WINDOW": "128", "T1K_BATCH": "16"}),      # single-end (-u)
                                      ("kir_synth_relax_2x150", {"T1K_FIRST_WINDOW": "128", "T1K_WINDOW": "512", "T1K_BATCH": "64"}),
                                      ("hla_synth_2x150", {"T1K_FIRST_WINDOW": "64", "T1K_WINDOW": "256", "T1K_BATCH": "32"})])          # with a barcode file (.fa.gz)
def test_gzip_read_files_streamed_under_the_loop(built, tmp_path, case, env):
    """the .gz fixtures handed to the window loop while they are still being inflated (t1k_reads_open_stream: host/inflate.cpp publishes its
    progress, the records are indexed behind it, windows are cut from what has arrived; ReadFiles.hpp:13,95 / kseq.h:94-150 stream through
    zlib): every output file equals the reference's, the fragment count is logged behind the loop, and a damaged file ends the job"""
    c = goldens.Case(case, str(tmp_path))
    out = os.path.join(str(tmp_path), "gzs")
    r1, r2 = os.path.join(c.dir, "reads_1.fq.gz"), os.path.join(c.dir, "reads_2.fq.gz")
    args = ["-f", c.ref] + (["-1", r1, "-2", r2] if c.paired else ["-u", r1]) + c.flags
    if c.bc:
        args += ["--barcode", os.path.join(c.dir, "barcodes.fa.gz")]   # streamed beside the mates; fragments = records with a barcode
    e = dict(os.environ, T1K_STREAM_GZ_MIN_MB="0.0001", T1K_DEBUG_PHASES="1", **env)
    r = subprocess.run([GENO] + args + ["-o", out], stderr=subprocess.PIPE, text=True, env=e)
    assert r.returncode == 0, r.stderr
    assert "gzip read files streamed" in r.stderr, r.stderr      # the streaming reader did run
    _golden_files_equal(c, out)
    if c.bc:
        assert open(out + "_aligned_bc.fa").read() == c.expected("aligned_bc.fa")
    found = [l for l in r.stderr.splitlines() if "Found " in l and "read fragments" in l]
    whole = subprocess.run([GENO] + args + ["-o", out + "_w"], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_STREAM_GZ="0"))
    assert whole.returncode == 0 and "gzip read files streamed" not in whole.stderr
    assert [l.split("] ")[-1] for l in found] == [l.split("] ")[-1] for l in whole.stderr.splitlines() if "Found " in l and "read fragments" in l]
    if not env:
        blob = bytearray(open(r1, "rb").read())
        blob[-7] ^= 0x21  # the trailer's CRC
        bad = os.path.join(str(tmp_path), "bad_1.fq.gz")
        open(bad, "wb").write(blob)
        r = subprocess.run([GENO, "-f", c.ref, "-1", bad, "-2", r2] + c.flags + ["-o", out + "_bad"], stderr=subprocess.PIPE, text=True, env=e)
        assert r.returncode != 0 and "damaged" in r.stderr, r.stderr
        blob = open(r1, "rb").read()
        cut = os.path.join(str(tmp_path), "cut_1.fq.gz")
        open(cut, "wb").write(blob[:len(blob) * 3 // 5])   # a file that ends in the middle of its (only) block: nothing of it is ever published, the
        r = subprocess.run([GENO, "-f", c.ref, "-1", cut, "-2", r2] + c.flags + ["-o", out + "_cut"], stderr=subprocess.PIPE, text=True, env=e)  # whole-file reader reports it
        assert r.returncode != 0 and "genotyper:" in r.stderr, r.stderr   # (a file cut behind several blocks, i.e. under the running loop: tests/test_host_reads_cpu.py)
        assert not os.path.exists(out + "_cut_genotype.tsv")


def test_gzip_lanes_streamed_under_the_loop(built, tmp_path):
    """three .gz files per mate (lanes) streamed back to back: the first ends without a line end, the second with blank lines, the third is CRLF"""
    import gzip
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    args = ["-f", c.ref]
    for m, path in ((1, c.r1), (2, c.r2)):
        lines = open(path).read().split("\n")
        recs = ["\n".join(lines[i:i + 4]) + "\n" for i in range(0, len(lines) - 1, 4)]
        cut = [0, len(recs) // 5, len(recs) * 2 // 3, len(recs)]
        texts = ["".join(recs[cut[0]:cut[1]]).rstrip("\n"), "".join(recs[cut[1]:cut[2]]) + "\n\n", "".join(recs[cut[2]:cut[3]]).replace("\n", "\r\n")]
        for i, t in enumerate(texts):
            g = os.path.join(str(tmp_path), "lane%d_%d.fq.gz" % (i, m))
            with gzip.open(g, "wb", compresslevel=(1, 6, 9)[i]) as f:
                f.write(t.encode())
            args += ["-%d" % m, g]
    out = os.path.join(str(tmp_path), "lanes")
    r = subprocess.run([GENO] + args + c.flags + ["-o", out], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_STREAM_GZ_MIN_MB="0.0001", T1K_DEBUG_PHASES="1"))
    assert r.returncode == 0, r.stderr
    assert "gzip read files streamed" in r.stderr, r.stderr
    _golden_files_equal(c, out)


def test_streamed_reads_through_the_c_abi_and_a_second_run(built, tmp_path):
    """t1k_reads_open_stream + t1k_job_attach_reads + t1k_job_run from a library caller: the same tables as the input opened whole; the caller's
    text stays resident (only the executables drop it), so the job can run again; t1k_reads_fragments waits for the stream's end"""
    os.environ["T1K_STREAM_GZ_MIN_MB"] = "0.0001"   # (read once, at the first streamed open of this process)
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    r1, r2 = os.path.join(c.dir, "reads_1.fq.gz"), os.path.join(c.dir, "reads_2.fq.gz")
    texts = []
    job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97)
    job.load_reads(r1, r2)
    job.run()
    texts.append((job.genotype_text(), job.counts()))
    job.close()
    rd = t1k_amd.Reads(r1, r2, stream=True)
    assert rd.fragments() == texts[0][1]["fragments"]      # (waits for the end of the stream)
    rd.close()
    job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97)
    job.attach_reads(t1k_amd.Reads(r1, r2, stream=True))
    job.run()
    texts.append((job.genotype_text(), job.counts()))
    job.run()                                              # the text is still there
    texts.append((job.genotype_text(), job.counts()))
    job.close()
    assert texts[0] == texts[1] == texts[2]


def test_streamed_gzip_that_leaves_the_strict_layout_is_opened_whole(built, tmp_path):
    """behind the head the streaming reader checks, mate 1 holds a blank line between two records and mate 2's last record has no quality
    lines: text the whole-file reader takes as the reference's kseq does (kseq.h:94-150), and the streaming reader does not follow.  The
    run must not end there: t1k_job_run opens the files whole and starts over -- same files as the golden run, from the executable and
    from a library caller"""
    import gzip
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    l1, l2 = open(c.r1).read().split("\n"), open(c.r2).read().split("\n")
    n = (len(l1) - 1) // 4
    k = 4 * (n * 3 // 5)
    t1 = "\n".join(l1[:k]) + "\n\n" + "\n".join(l1[k:])
    t2 = "\n".join(l2[:4 * n - 2]) + "\n"
    g1, g2 = os.path.join(str(tmp_path), "odd_1.fq.gz"), os.path.join(str(tmp_path), "odd_2.fq.gz")
    for g, t in ((g1, t1), (g2, t2)):
        with gzip.open(g, "wb") as f:
            f.write(t.encode())
    env = dict(os.environ, T1K_STREAM_GZ_MIN_MB="0.0001", T1K_STREAM_HEAD_MB="0.01", T1K_DEBUG_PHASES="1")
    for tag, files in (("both", (g1, g2)), ("blank", (g1, c.r2 + ".gz")), ("tail", (c.r1 + ".gz", g2))):
        for plain in (c.r1, c.r2):
            if not os.path.exists(plain + ".gz"):
                with open(plain, "rb") as f, gzip.open(plain + ".gz", "wb") as g:
                    g.write(f.read())
        out = os.path.join(str(tmp_path), "odd_" + tag)
        r = subprocess.run([GENO, "-f", c.ref, "-1", files[0], "-2", files[1]] + c.flags + ["-o", out], stderr=subprocess.PIPE, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert "opened whole and the job starts over" in r.stderr, r.stderr
        _golden_files_equal(c, out)
    # a library caller (this process read T1K_STREAM_GZ_MIN_MB at its first streamed open; the head size is read here for the first time)
    os.environ["T1K_STREAM_GZ_MIN_MB"] = "0.0001"
    os.environ["T1K_STREAM_HEAD_MB"] = "0.01"
    job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97)
    job.load_reads(c.r1, c.r2)
    job.run()
    want = (job.genotype_text(), job.counts())
    job.close()
    job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97)
    job.attach_reads(t1k_amd.Reads(g1, g2, stream=True))
    job.run()
    assert (job.genotype_text(), job.counts()) == want
    job.close()


def test_reads_opened_beside_job_creation(built, tmp_path):
    """t1k_reads_open on a second thread while t1k_job_create runs, then t1k_job_attach_reads (what the executable and bench.py do) leaves
    the job as t1k_job_load_reads does; the executable's serial order (T1K_SERIAL_OPEN=1) writes the same files; a failed open arrives
    as the job's error"""
    import threading
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    texts = []
    job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97)
    job.load_reads(c.r1, c.r2)
    job.run()
    texts.append((job.genotype_text(), job.counts()))
    job.close()
    box = {}
    th = threading.Thread(target=lambda: box.update(reads=t1k_amd.Reads(c.r1, c.r2)))
    th.start()
    job = t1k_amd.Job(c.ref, ref_seq_similarity=0.97)
    th.join()
    assert box["reads"].fragments() == texts[0][1]["fragments"]
    job.attach_reads(box["reads"])
    assert box["reads"].h is None  # consumed
    job.run()
    texts.append((job.genotype_text(), job.counts()))
    job.close()
    assert texts[0] == texts[1]
    with pytest.raises(t1k_amd.T1kError):
        t1k_amd.Reads(os.path.join(str(tmp_path), "missing_1.fq"), c.r2)
    out = os.path.join(str(tmp_path), "serial")
    r = subprocess.run([GENO] + c.args() + ["-o", out], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_SERIAL_OPEN="1"))
    assert r.returncode == 0, r.stderr
    _golden_files_equal(c, out)
    r = subprocess.run([GENO, "-f", c.ref, "-1", os.path.join(str(tmp_path), "missing_1.fq"), "-2", c.r2, "-o", out], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "missing_1.fq" in r.stderr


def test_several_files_per_mate_and_wrapped_fasta(built, tmp_path):
    """every -1 / -2 counts and the files are read back to back (ReadFiles::AddReadFile); the second half is given as FASTA with the
    sequences wrapped at 60 columns and CRLF line ends, which the in-place indexer hands to the general (kseq-rule) reader"""
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    parts = {}
    for m, path in ((1, c.r1), (2, c.r2)):
        lines = open(path).read().split("\n")
        recs = [lines[i:i + 4] for i in range(0, len(lines) - 1, 4)]
        half = len(recs) // 2
        a, b = os.path.join(str(tmp_path), "a_%d.fq" % m), os.path.join(str(tmp_path), "b_%d.fa" % m)
        with open(a, "w") as f:
            f.write("".join("\n".join(r) + "\n" for r in recs[:half]))
        with open(b, "w", newline="") as f:
            for r in recs[half:]:
                f.write(">" + r[0][1:] + "\r\n")
                for k in range(0, len(r[1]), 60):
                    f.write(r[1][k:k + 60] + "\r\n")
        parts[m] = (a, b)
    out = os.path.join(str(tmp_path), "multi")
    args = ["-f", c.ref, "-1", parts[1][0], "-1", parts[1][1], "-2", parts[2][0], "-2", parts[2][1]] + c.flags
    r = subprocess.run([GENO] + args + ["-o", out], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    _golden_files_equal(c, out)


@pytest.mark.parametrize("gpus", ["0,0", "0,0,0"])
def test_own_input_sharding_over_several_files_per_mate(built, tmp_path, gpus):
    """ranks that index only their own reads, with the reads of each mate spread over three files of unequal size (the multi-GPU
    benchmark gives one file per rank): the slices of the ranks start and end inside files and span file boundaries; one file ends
    without a newline, one has CRLF line ends.  Every output file must equal the single-file golden run."""
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    args = ["-f", c.ref]
    for m, path in ((1, c.r1), (2, c.r2)):
        lines = open(path).read().split("\n")
        recs = [lines[i:i + 4] for i in range(0, len(lines) - 1, 4)]
        n = len(recs)
        cuts = [0, n // 7, n // 7 + n // 2, n]
        for k in range(3):
            part = os.path.join(str(tmp_path), "p%d_%d.fq" % (k, m))
            text = "".join("\n".join(r) + "\n" for r in recs[cuts[k]:cuts[k + 1]])
            if k == 0:
                text = text[:-1]                       # no newline at the end of the file
            if k == 1:
                text = text.replace("\n", "\r\n")
            with open(part, "w", newline="") as f:
                f.write(text)
            args += ["-%d" % m, part]
    out = os.path.join(str(tmp_path), "multi_own")
    r = subprocess.run([GENO] + args + c.flags + ["-o", out], stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, T1K_GPUS=gpus, T1K_SHARD_INPUT="1", T1K_DEBUG_PHASES="1"))
    assert r.returncode == 0, r.stderr
    parts = [(int(a), int(b)) for a, b in re.findall(r"mapped \+ indexed: (\d+) of (\d+) fragments", r.stderr)]
    assert len(parts) == len(gpus.split(",")) and all(0 < a < b for a, b in parts) and sum(a for a, _ in parts) == parts[0][1], parts
    _golden_files_equal(c, out)
    one = os.path.join(str(tmp_path), "one")
    r1 = subprocess.run([GENO] + c.args() + ["-o", one], stderr=subprocess.PIPE, text=True)
    assert r1.returncode == 0, r1.stderr
    for suf in ("_aligned_1.fa", "_aligned_2.fa"):
        assert open(out + suf, "rb").read() == open(one + suf, "rb").read(), suf


def test_random_loop_geometries(built):
    """tools/env_sweep_r06.py: the executable on the committed fixtures under 40 random loop geometries (first window, window limit, assignment and pairing
    ranges, 1-4 pipelines, 1-3 ranks, coverage mode, cross-window table, host-driven chain) -- every run exits 0 with the reference's committed files.  The
    tool exists because a clear one entry past its block survived five rounds of hand-picked small-window settings; 650 geometries ran clean at the end of
    round 6 (profiles/r06_env_sweep.log)."""
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "env_sweep_r06.py"), "40", "20260930"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


def test_later_range_one_entry_past_a_smaller_first_range(built, tmp_path):
    """round 6 (found by tests/test_gpu_fuzz.py under 96-read-end ranges): a context's per-read-end count blocks were asked for n entries and cleared for
    n + 1; a block keeps an eighth + 256 bytes of slack, so the clear stayed inside it unless a LATER range of the context had exactly the number of
    read-ends at which the slack ends -- 29 read-ends first, 96 later: 386 bytes held, 388 cleared, `hipMemsetAsync: invalid argument` and a failed job.
    Here: one pipeline, a first window of 15 fragments with 29 distinct read-ends (two of its read-ends are the same sequence), then ranges of 96; the
    files must equal the ones of an undisturbed run."""
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref("ref-rna", ref, genes=3, scale=0.05, seed=77)
    util.synth_reads(ref, os.path.join(tmp, "r"), pairs=900, len=150, seed=78)
    l1 = open(os.path.join(tmp, "r_1.fq")).read().split("\n")
    l1[4 * 2 + 1] = l1[1]      # fragment 2's first mate = fragment 0's: 29 distinct read-ends among the first 15 fragments
    open(os.path.join(tmp, "r_1.fq"), "w").write("\n".join(l1))
    args = [GENO, "-f", ref, "-1", os.path.join(tmp, "r_1.fq"), "-2", os.path.join(tmp, "r_2.fq"), "-s", "0.9", "--outputReadAssignment"]
    a, b = os.path.join(tmp, "plain"), os.path.join(tmp, "ranges")
    r = subprocess.run(args + ["-o", a], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    for first in ("15", "14", "16"):   # (14 / 16: neighbours of the size that hit the slack's end)
        env = dict(os.environ, T1K_PIPELINES="1", T1K_FIRST_WINDOW=first, T1K_WINDOW="512", T1K_BATCH="48", T1K_PAIR_BATCH="96", T1K_CROSS_WINDOW="0")
        r = subprocess.run(args + ["-o", b], stderr=subprocess.PIPE, text=True, env=env)
        assert r.returncode == 0, (first, r.stderr[-1500:])
        for suf in ("_genotype.tsv", "_allele.tsv", "_assign.tsv", "_aligned_1.fa", "_aligned_2.fa"):
            assert open(a + suf).read() == open(b + suf).read(), (first, suf)


@pytest.mark.parametrize("name", ["hla_synth_2x150", "cyp_dna_relax_2x150", "cyp_rna_single"])
@pytest.mark.parametrize("env", [{"T1K_FIRST_WINDOW": "8", "T1K_WINDOW": "96", "T1K_BATCH": "8", "T1K_PAIR_BATCH": "16"},
                                 {"T1K_FIRST_WINDOW": "24", "T1K_WINDOW": "48", "T1K_WINDOW_GROWTH": "1", "T1K_BATCH": "16", "T1K_PAIR_BATCH": "8", "T1K_PIPELINES": "2"},
                                 {"T1K_FIRST_WINDOW": "16", "T1K_WINDOW": "40", "T1K_BATCH": "8", "T1K_NO_STREAM_OUTPUT": "1"},
                                 {"T1K_FIRST_WINDOW": "8", "T1K_WINDOW": "32", "T1K_BATCH": "8", "T1K_PAIR_BATCH": "8", "T1K_GPUS": "0,0"},
                                 {"T1K_FIRST_WINDOW": "8", "T1K_WINDOW": "32", "T1K_BATCH": "8", "T1K_PAIR_BATCH": "8", "T1K_GPUS": "0,0,0", "T1K_SHARD_INPUT": "1"}])
def test_many_small_windows_and_ranges(built, tmp_path, name, env):
    """the job loop as it runs on millions of fragments, on a fixture: many windows cut at run time (a tiny first one, then sized from
    the measured rates or by a fixed factor), several ranges per window, pipelines moving on to the next window while the last ranges
    of the previous one are still out, the output writer following the pairing ranges and releasing the input behind it.  Every file
    must equal the golden one byte for byte."""
    c = goldens.Case(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "win")
    r = subprocess.run([GENO] + c.args() + ["-o", out], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_DEBUG_PHASES="1", **env))
    assert r.returncode == 0, r.stderr
    m = re.search(r"(\d+) windows,", r.stderr)
    assert m and int(m.group(1)) >= 3, r.stderr[-1500:]
    _golden_files_equal(c, out)
    if c.bc:
        assert open(out + "_aligned_bc.fa").read() == c.expected("aligned_bc.fa")
    one = os.path.join(str(tmp_path), "one")
    r1 = subprocess.run([GENO] + c.args() + ["-o", one], stderr=subprocess.PIPE, text=True)
    assert r1.returncode == 0, r1.stderr
    for suf in (("_aligned_1.fa", "_aligned_2.fa") if c.paired else ("_aligned.fa",)):
        assert open(out + suf, "rb").read() == open(one + suf, "rb").read(), suf
    # identical read-ends across windows are assigned once (t1k_xwin): with the table off every window assigns its own copies -- more
    # distinct read-ends, the same files
    off = os.path.join(str(tmp_path), "off")
    r0 = subprocess.run([GENO] + c.args() + ["-o", off], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_DEBUG_PHASES="1", T1K_CROSS_WINDOW="0", **env))
    assert r0.returncode == 0, r0.stderr
    _golden_files_equal(c, off)
    if "T1K_GPUS" not in env:  # (several ranks print a line each, in any order)
        d_on, d_off = (int(re.search(r"read-ends -> (\d+) distinct", x.stderr).group(1)) for x in (r, r0))
        assert d_on <= d_off
        if name == "cyp_rna_single":  # (thousands of reads of one short gene: every window repeats sequences of the ones before)
            assert d_on < d_off, (d_on, d_off)


@pytest.mark.parametrize("case", ["empty", "one_pair", "shorter_than_k", "single_end_empty"])
@pytest.mark.parametrize("gpus", ["", "0,0", "0,0,0+own-input"])
def test_degenerate_inputs_vs_reference_binary(built, tmp_path, case, gpus):
    """read files with no records, with one pair, with reads shorter than a k-mer: the reference finishes and writes (empty) files;
    so must this build, with one rank and with ranks that have nothing to do -- every file byte for byte"""
    util.need(util.REF_BIN)
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    r1, r2 = str(tmp_path / "x_1.fq"), str(tmp_path / "x_2.fq")
    l1, l2 = open(c.r1).read().split("\n"), open(c.r2).read().split("\n")
    if case in ("empty", "single_end_empty"):
        text1 = text2 = ""
    elif case == "one_pair":
        text1, text2 = "\n".join(l1[:4]) + "\n", "\n".join(l2[:4]) + "\n"
    else:
        text1, text2 = "@s\nACGTACG\n+\nIIIIIII\n@t\nACG\n+\nIII\n", "@s\nTTTTACG\n+\nIIIIIII\n@t\nN\n+\nI\n"
    open(r1, "w").write(text1)
    open(r2, "w").write(text2)
    reads = ["-u", r1] if case == "single_end_empty" else ["-1", r1, "-2", r2]
    o_ref, o_gpu = str(tmp_path / "ref"), str(tmp_path / "gpu")
    a = subprocess.run([util.REF_BIN, "-f", ref] + reads + ["-o", o_ref], stderr=subprocess.PIPE, text=True)
    env = dict(os.environ)
    if gpus:
        env["T1K_GPUS"] = gpus.split("+")[0]
        if "+" in gpus:
            env["T1K_SHARD_INPUT"] = "1"
    b = subprocess.run([GENO, "-f", ref] + reads + ["-o", o_gpu], stderr=subprocess.PIPE, text=True, env=env)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-500:], b.stderr[-500:])
    sufs = ["_genotype.tsv", "_allele.tsv"] + (["_aligned.fa"] if case == "single_end_empty" else ["_aligned_1.fa", "_aligned_2.fa"])
    for suf in sufs:
        assert open(o_ref + suf, "rb").read() == open(o_gpu + suf, "rb").read(), suf


@pytest.mark.parametrize("layout", ["plain", "crlf", "no_final_newline"])
def test_odd_but_legal_records_vs_reference_binary(built, tmp_path, layout):
    """a fixture with odd records mixed in -- empty sequences (one mate, both mates), reads of 5 / 11 / 37 bases, all-N reads, tabs and
    comments in the header lines -- as LF, as CRLF and without a final newline: every output file against the reference binary's"""
    util.need(util.REF_BIN)
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    out = {}
    for m, path in ((1, c.r1), (2, c.r2)):
        lines = open(path).read().split("\n")
        out[m] = [lines[i:i + 4] for i in range(0, len(lines) - 1, 4)]
    for i, (a, b) in enumerate(zip(out[1], out[2])):
        k = i % 9
        if k == 1: a[1] = a[3] = ""
        if k == 2: b[1], b[3] = b[1][:5], b[3][:5]
        if k == 3: a[1] = "N" * len(a[1])
        if k == 4: a[0] += "\tcomment with tab"; b[0] += " comment"
        if k == 5: a[1], a[3], b[1], b[3] = a[1][:37], a[3][:37], b[1][:11], b[3][:11]
        if k == 6: a[1] = a[3] = b[1] = b[3] = ""
    files = []
    for m in (1, 2):
        text = "".join("\n".join(r) + "\n" for r in out[m])
        if layout == "crlf":
            text = text.replace("\n", "\r\n")
        if layout == "no_final_newline":
            text = text.rstrip("\r\n")
        files.append(str(tmp_path / ("odd_%d.fq" % m)))
        open(files[-1], "w", newline="").write(text)
    args = ["-f", ref, "-1", files[0], "-2", files[1]] + util.CYP_FLAGS
    o_ref, o_gpu = str(tmp_path / "ref"), str(tmp_path / "gpu")
    a = subprocess.run([util.REF_BIN] + args + ["-o", o_ref], stderr=subprocess.PIPE, text=True)
    b = subprocess.run([GENO] + args + ["-o", o_gpu], stderr=subprocess.PIPE, text=True)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-500:], b.stderr[-500:])
    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        assert open(o_ref + suf, "rb").read() == open(o_gpu + suf, "rb").read(), suf
    assert os.path.getsize(o_gpu + "_aligned_1.fa") > 1000


@pytest.mark.parametrize("with_barcode", [False, True])
def test_analyzer_with_nothing_genotyped_vs_reference_binary(built, tmp_path, with_barcode):
    """run-t1k starts the analyzer after the genotyper even when no allele was reported: an empty <prefix>_allele.tsv.  The reference
    finishes with an empty VCF (and a header-only per-barcode table); so must this build, for empty and for non-empty read files"""
    util.need(util.REF_ANALYZER)
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    empty = str(tmp_path / "empty.fq")
    open(empty, "w").close()
    alleles = str(tmp_path / "none_allele.tsv")
    open(alleles, "w").close()
    for tag, r1, r2 in (("e", empty, empty), ("r", c.r1, c.r2)):
        if with_barcode and tag == "r":
            continue  # (a barcode file has to match the read file record for record)
        extra = ["--barcode", empty] if with_barcode else []
        outs = {}
        for who, binary in (("ref", util.REF_ANALYZER), ("gpu", ANALYZER)):
            o = str(tmp_path / ("%s_%s" % (who, tag)))
            r = subprocess.run([binary, "-f", ref, "-a", alleles, "-1", r1, "-2", r2, "-o", o, "--varMaxGroup", "0"] + extra, stderr=subprocess.PIPE, text=True)
            assert r.returncode == 0, (who, r.stderr[-500:])
            outs[who] = {f[len(os.path.basename(o)):]: open(os.path.join(str(tmp_path), f), "rb").read() for f in os.listdir(str(tmp_path)) if f.startswith(os.path.basename(o) + "_")}
        assert outs["ref"] == outs["gpu"] and "_allele.vcf" in outs["gpu"], (sorted(outs["ref"]), sorted(outs["gpu"]))


@pytest.mark.parametrize("which", ["all_missing", "first_and_last_kept", "every_7th_missing"])
def test_missing_barcodes_vs_reference_binary(built, tmp_path, which):
    """records whose barcode is "missing_barcode" are dropped with their mates (Genotyper.cpp:376-381): all of them, all but the first and
    the last record, every seventh -- one GPU, two ranks, and many tiny windows with the writer (and the release of the mapped input)
    behind them; every file against the reference binary's"""
    util.need(util.REF_BIN)
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    names = [l[1:].split()[0].rsplit("/", 1)[0] for i, l in enumerate(open(c.r1)) if i % 4 == 0]
    n = len(names)
    keep = {"all_missing": lambda i: False, "first_and_last_kept": lambda i: i in (0, n - 1), "every_7th_missing": lambda i: i % 7 != 0}[which]
    bc = str(tmp_path / "bc.fa")
    with open(bc, "w") as f:
        for i in range(n):
            f.write(">%s\n%s\n" % (names[i], ("ACGTACGTAC" + "ACGT"[i % 4] * 2) if keep(i) else "missing_barcode"))
    args = ["-f", ref, "-1", c.r1, "-2", c.r2, "--barcode", bc] + util.CYP_FLAGS
    o_ref = str(tmp_path / "ref")
    a = subprocess.run([util.REF_BIN] + args + ["-o", o_ref], stderr=subprocess.PIPE, text=True)
    assert a.returncode == 0, a.stderr[-500:]
    for tag, env in (("one", {}), ("two_ranks", {"T1K_GPUS": "0,0"}), ("tiny_windows", {"T1K_FIRST_WINDOW": "8", "T1K_WINDOW": "40", "T1K_BATCH": "8", "T1K_PAIR_BATCH": "8"})):
        o = str(tmp_path / tag)
        b = subprocess.run([GENO] + args + ["-o", o], stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        assert b.returncode == 0, (tag, b.stderr[-500:])
        for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa", "_aligned_bc.fa"):
            assert open(o_ref + suf, "rb").read() == open(o + suf, "rb").read(), (tag, suf)


@pytest.mark.parametrize("kind,length,flags", [("ref-rna", 250, ["-s", "0.9"]), ("ref-dna", 300, ["-s", "0.9", "--relaxIntronAlign"]), ("ref-rna", 320, ["-s", "0.97"])])
def test_long_reads_end_to_end_vs_reference_binary(built, tmp_path, kind, length, flags):
    """2 x 250 / 300 / 320 bp reads (the 320-position instantiations of the seeding and chaining kernels, the general window loop of
    k_fullalign, wider sort keys) through the whole stage: one GPU, two ranks, many small windows -- every file against the reference's"""
    util.need(util.REF_BIN)
    ref = str(tmp_path / "ref.fa")
    util.synth_ref(kind, ref, genes=6, scale=0.2, seed=77)
    pfx = str(tmp_path / "r")
    util.synth_reads(ref, pfx, pairs=4000, len=length, seed=5, sub=0.008, fragmean=2 * length + 40)
    args = ["-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] + flags
    o_ref = str(tmp_path / "ref")
    a = subprocess.run([util.REF_BIN] + args + ["-t", "32", "-o", o_ref], stderr=subprocess.PIPE, text=True)
    assert a.returncode == 0, a.stderr[-500:]
    for tag, env in (("one", {}), ("two_ranks", {"T1K_GPUS": "0,0"}), ("small_windows", {"T1K_FIRST_WINDOW": "64", "T1K_WINDOW": "1500", "T1K_BATCH": "128", "T1K_PAIR_BATCH": "256"})):
        o = str(tmp_path / tag)
        b = subprocess.run([GENO] + args + ["-o", o], stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        assert b.returncode == 0, (tag, b.stderr[-500:])
        for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
            assert open(o_ref + suf, "rb").read() == open(o + suf, "rb").read(), (tag, suf)
        assert os.path.getsize(o + "_aligned_1.fa") > 100000


@pytest.mark.parametrize("kind,length,flags,indel", [("ref-rna", 400, ["-s", "0.97"], 0.0005), ("ref-dna", 600, ["-s", "0.9", "--relaxIntronAlign"], 0.002),
                                                     ("ref-rna", 1000, ["-s", "0.9"], 0.002)])
def test_reads_beyond_the_hit_masks_vs_reference_binary(built, tmp_path, kind, length, flags, indel):
    """2 x 400 / 600 / 1000 bp reads (beyond T1K_MAX_READ_LEN = 320: k_seed_long, every group through the explicit-hit-list kernels, the
    wide key fields of k_select / k_truncate, full alignments of more than 320 columns through the traced DP) through the whole stage:
    deferred and eager coverage, two ranks, small windows -- every file against the reference's"""
    util.need(util.REF_BIN)
    ref = str(tmp_path / "ref.fa")
    util.synth_ref(kind, ref, genes=4, scale=0.1, seed=77)
    pfx = str(tmp_path / "r")
    util.synth_reads(ref, pfx, pairs=1200, len=length, seed=5, sub=0.008, indel=indel, fragmean=2 * length + 40)
    args = ["-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] + flags
    o_ref = str(tmp_path / "ref")
    a = subprocess.run([util.REF_BIN] + args + ["-t", "32", "-o", o_ref], stderr=subprocess.PIPE, text=True)
    assert a.returncode == 0, a.stderr[-500:]
    for tag, env in (("one", {}), ("eager", {"T1K_COVERAGE": "eager"}), ("two_ranks", {"T1K_GPUS": "0,0"}),
                     ("small_windows", {"T1K_FIRST_WINDOW": "64", "T1K_WINDOW": "500", "T1K_BATCH": "128", "T1K_PAIR_BATCH": "256"})):
        o = str(tmp_path / tag)
        b = subprocess.run([GENO] + args + ["-o", o], stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        assert b.returncode == 0, (tag, b.stderr[-500:])
        for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
            assert open(o_ref + suf, "rb").read() == open(o + suf, "rb").read(), (tag, suf)
        assert os.path.getsize(o + "_aligned_1.fa") > 100000


def test_a_few_long_reads_among_short_ones_vs_reference_binary(built, tmp_path):
    """the case the limit used to turn into an aborted run: a 2 x 150 bp file that holds a handful of 2 x 500 bp pairs (and one end of
    700 bases whose mate is short).  Only the windows that hold such a read take the long path; every file equals the reference's,
    for one window, for many small ones (most of them all-short) and for two ranks"""
    util.need(util.REF_BIN)
    ref = str(tmp_path / "ref.fa")
    util.synth_ref("ref-rna", ref, genes=5, scale=0.1, seed=78)
    util.synth_reads(ref, str(tmp_path / "s"), pairs=3000, len=150, seed=6, sub=0.006)
    util.synth_reads(ref, str(tmp_path / "l"), pairs=40, len=500, seed=7, sub=0.006, indel=0.001, fragmean=1040)
    util.synth_reads(ref, str(tmp_path / "x"), pairs=4, len=700, seed=8, sub=0.006, fragmean=1500)
    rec = {m: [open(str(tmp_path / ("%s_%d.fq" % (t, m)))).read().split("\n") for t in "slx"] for m in (1, 2)}
    out = {1: [], 2: []}
    nl, nx = 0, 0
    for i in range(3000):
        for m in (1, 2):
            out[m] += rec[m][0][4 * i:4 * i + 4]
        if i % 75 == 37:       # a long pair
            for m in (1, 2):
                out[m] += rec[m][1][4 * nl:4 * nl + 4]
            nl += 1
        if i % 750 == 400:     # one long end, its mate cut back to 150 bases
            q = rec[1][2][4 * nx:4 * nx + 4]
            out[1] += q
            r2 = rec[2][2][4 * nx:4 * nx + 4]
            out[2] += [r2[0], r2[1][:150], r2[2], r2[3][:150]]
            nx += 1
    for m in (1, 2):
        open(str(tmp_path / ("mix_%d.fq" % m)), "w").write("\n".join(out[m]) + "\n")
    args = ["-f", ref, "-1", str(tmp_path / "mix_1.fq"), "-2", str(tmp_path / "mix_2.fq"), "-s", "0.97"]
    o_ref = str(tmp_path / "ref")
    a = subprocess.run([util.REF_BIN] + args + ["-t", "32", "-o", o_ref], stderr=subprocess.PIPE, text=True)
    assert a.returncode == 0, a.stderr[-500:]
    for tag, env in (("one", {}), ("two_ranks", {"T1K_GPUS": "0,0"}), ("small_windows", {"T1K_FIRST_WINDOW": "64", "T1K_WINDOW": "400", "T1K_BATCH": "128", "T1K_PAIR_BATCH": "256"})):
        o = str(tmp_path / tag)
        b = subprocess.run([GENO] + args + ["-o", o], stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        assert b.returncode == 0, (tag, b.stderr[-500:])
        for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
            assert open(o_ref + suf, "rb").read() == open(o + suf, "rb").read(), (tag, suf)


def test_long_reads_against_the_benchmark_sized_reference(built, tmp_path):
    """the long path next to lists of benchmark size: the HLA-like reference at full scale (29 135 alleles, 15 allele bits in the sort
    keys; short read-ends there hold more than 8192 candidates, which the 1000-base key layout could not index -- the layout follows
    the read-end) with 2 x 150 bp pairs and a few 2 x 500 / 2 x 900 bp pairs among them, against the reference binary"""
    util.need(util.REF_BIN)
    ref = str(tmp_path / "ref.fa")
    util.synth_ref("ref-rna", ref, genes=24, scale=1.0, seed=20250614)
    util.synth_reads(ref, str(tmp_path / "s"), pairs=5000, len=150, seed=16)
    util.synth_reads(ref, str(tmp_path / "l"), pairs=4, len=500, seed=17, fragmean=1040)
    util.synth_reads(ref, str(tmp_path / "x"), pairs=2, len=900, seed=18, indel=0.001, fragmean=1900)
    for m in (1, 2):
        sh = open(str(tmp_path / ("s_%d.fq" % m))).read().split("\n")
        lo = open(str(tmp_path / ("l_%d.fq" % m))).read().split("\n")[:16]
        xl = open(str(tmp_path / ("x_%d.fq" % m))).read().split("\n")[:8]
        open(str(tmp_path / ("mix_%d.fq" % m)), "w").write("\n".join(sh[:8000] + lo + sh[8000:16000] + xl + sh[16000:]))
    args = ["-f", ref, "-1", str(tmp_path / "mix_1.fq"), "-2", str(tmp_path / "mix_2.fq"), "-s", "0.97"]
    o_ref = str(tmp_path / "ref")
    a = subprocess.run([util.REF_BIN] + args + ["-t", "32", "-o", o_ref], stderr=subprocess.PIPE, text=True)
    assert a.returncode == 0, a.stderr[-500:]
    o = str(tmp_path / "gpu")
    b = subprocess.run([GENO] + args + ["-o", o], stderr=subprocess.PIPE, text=True)
    assert b.returncode == 0, b.stderr[-800:]
    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        assert open(o_ref + suf, "rb").read() == open(o + suf, "rb").read(), suf


@pytest.mark.parametrize("fasta_gz,length,sim,relax", [("CYP_RNA", 420, 0.8, False), ("CYP_DNA", 700, 0.9, True), ("CYP_RNA", 1000, 0.8, False)])
def test_long_read_overlap_lists_and_coverage_vs_oracle(built, tmp_path, fasta_gz, length, sim, relax):
    """AssignRead stage on reads beyond the hit masks: every overlap list (coordinates, match counts, relaxed counts, similarity bits) and
    the per-base coverage of every allele against the oracle's.  Includes reads that repeat themselves (hits on many diagonals 100
    apart), the shape the old over-long-read test used"""
    import gpu_assign_check
    fa = util.gunzip_to(getattr(util, fasta_gz), str(tmp_path / "ref.fa"))
    util.synth_reads(fa, str(tmp_path / "r"), pairs=40, len=length, seed=21, sub=0.006, indel=0.002, fragmean=2 * length + 40)
    reads = [s for _, _, s in t1k_amd.read_fastx(str(tmp_path / "r_1.fq"))] + [s for _, _, s in t1k_amd.read_fastx(str(tmp_path / "r_2.fq"))]
    util.synth_reads(fa, str(tmp_path / "q"), pairs=10, len=100, seed=22, sub=0.004)
    short = [s for _, _, s in t1k_amd.read_fastx(str(tmp_path / "q_1.fq"))]
    reads += short + [s * 4 for s in short[:4]] + [short[4] + short[5] + short[4]]
    assert gpu_assign_check.compare(fa, reads, sim, relax, "long reads %s %d" % (fasta_gz, length)) == 0


def test_over_long_read_fails_before_any_output(built, tmp_path):
    """reads longer than max_read_len (1000: read coordinates of the packed overlap records) are not handled by this build: the run must
    stop with a message BEFORE an output file exists"""
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    long1 = os.path.join(str(tmp_path), "long_1.fq")
    lines = open(c.r1).read().split("\n")
    lines[4 * 7 + 1] = lines[4 * 7 + 1] * 12         # one 1200-base read in the middle of the file
    lines[4 * 7 + 3] = lines[4 * 7 + 3] * 12
    open(long1, "w").write("\n".join(lines))
    out = os.path.join(str(tmp_path), "long")
    r = subprocess.run([GENO, "-f", c.ref, "-1", long1, "-2", c.r2, "-o", out] + c.flags, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "longer than this build handles" in r.stderr
    assert not any(f.startswith("long_") and f != "long_1.fq" for f in os.listdir(str(tmp_path)))


@pytest.mark.parametrize("gpus", ["", "0,0"])
def test_over_long_reads_can_be_set_aside(built, tmp_path, gpus):
    """T1K_LONG_READS=drop: the fragments of reads beyond 1000 bases take no part, the run finishes with a warning, and every output
    file equals the run on the same files without those fragments (the reference itself would have genotyped them: an escape hatch
    for the odd over-long read in millions, not parity)"""
    c = goldens.Case("cyp_rna_2x100", str(tmp_path))
    l1, l2 = open(c.r1).read().split("\n"), open(c.r2).read().split("\n")
    long_at = (7, 40)
    with_long, without = [list(l1), list(l2)], [[], []]
    for k in long_at:
        with_long[k % 2][4 * k + 1] = with_long[k % 2][4 * k + 1] * 12   # a 1200-base read, once in mate 1's file and once in mate 2's
        with_long[k % 2][4 * k + 3] = with_long[k % 2][4 * k + 3] * 12
    for m, src in enumerate((l1, l2)):
        for i in range(0, len(src) - 1, 4):
            if i // 4 not in long_at:
                without[m] += src[i:i + 4]
    paths = {}
    for tag, data in (("long", with_long), ("cut", without)):
        for m in (0, 1):
            paths[tag, m] = os.path.join(str(tmp_path), "%s_%d.fq" % (tag, m + 1))
            open(paths[tag, m], "w").write("\n".join(data[m]) + ("\n" if tag == "cut" else ""))
    env = dict(os.environ, T1K_LONG_READS="drop")
    if gpus:
        env["T1K_GPUS"] = gpus
    a, b = os.path.join(str(tmp_path), "a"), os.path.join(str(tmp_path), "b")
    r = subprocess.run([GENO, "-f", c.ref, "-1", paths["long", 0], "-2", paths["long", 1], "-o", a] + c.flags, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "set aside" in r.stderr and "WARNING" in r.stderr
    r = subprocess.run([GENO, "-f", c.ref, "-1", paths["cut", 0], "-2", paths["cut", 1], "-o", b] + c.flags, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        assert open(a + suf, "rb").read() == open(b + suf, "rb").read(), suf


@pytest.mark.parametrize("label,kind,scale,genes,pairs,flags", [
    ("config2_hla_rna_100k", "ref-rna", 1.0, 24, 100000, ["-s", "0.97"]),                                  # BASELINE configs[1], 100 k of its pairs
    ("config3_kir_wgs_100k", "ref-dna", 1.0, 17, 100000, ["-s", "0.9", "--relaxIntronAlign"]),           # --preset kir-wgs (run-t1k:300-304)
])
def test_baseline_configs_live_vs_reference_binary(built, tmp_path, label, kind, scale, genes, pairs, flags):
    """BASELINE.json configs 2 and 3 at 100 k pairs against the reference binary run here (all host cores): every file identical"""
    util.need(util.REF_BIN)
    ref = os.path.join(str(tmp_path), "ref.fa")
    util.synth_ref(kind, ref, genes=genes, scale=scale, seed=20250614)
    pfx = os.path.join(str(tmp_path), "r")
    util.synth_reads(ref, pfx, pairs=pairs, len=150, seed=2)
    common = ["-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] + flags
    o_ref, o_gpu = os.path.join(str(tmp_path), "ref"), os.path.join(str(tmp_path), "gpu")
    r1 = subprocess.run([util.REF_BIN] + common + ["-t", str(min(os.cpu_count() or 8, 128)), "-o", o_ref], stderr=subprocess.PIPE, text=True)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([GENO] + common + ["-o", o_gpu], stderr=subprocess.PIPE, text=True)
    assert r2.returncode == 0, r2.stderr[-2000:]
    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        assert open(o_ref + suf).read() == open(o_gpu + suf).read(), suf
    it = [re.search(r"in (\d+) EM iterations", x.stderr).group(1) for x in (r1, r2)]
    assert it[0] == it[1]
    # the same input (tens of 1 MiB blocks per file) through three ranks that each index and write only their own reads
    o_own = os.path.join(str(tmp_path), "own")
    r3 = subprocess.run([GENO] + common + ["-o", o_own], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_GPUS="0,0,0", T1K_SHARD_INPUT="1", T1K_DEBUG_PHASES="1"))
    assert r3.returncode == 0, r3.stderr[-2000:]
    assert re.search(r"indexed: 3333\d of 100000 fragments", r3.stderr), r3.stderr[-2000:]
    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        assert open(o_ref + suf).read() == open(o_own + suf).read(), suf


ANALYZER = os.path.join(util.ROOT, "t1k_amd", "bin", "analyzer")


def test_analyzer_vs_golden_reference_outputs(built, tmp_path):
    """genotyper -> analyzer as run-t1k:438-449 chains them; the per-barcode table must equal the file the reference's
    analyzer wrote on the same inputs (tools/make_analyzer_goldens.py); the reference called no variant there, so its VCF is empty"""
    c = goldens.Case("hla_synth_2x150", str(tmp_path))
    g, a = os.path.join(str(tmp_path), "g"), os.path.join(str(tmp_path), "a")
    r = subprocess.run([GENO] + c.args() + ["-o", g], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([ANALYZER, "-f", c.ref, "-a", g + "_allele.tsv", "-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa", "--barcode", g + "_aligned_bc.fa", "-o", a] + c.flags,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    # (the default mode, --varMaxGroup 8, in which the fixture was written: the variant caller runs and finds nothing to call)
    assert open(a + "_barcode_expr.tsv").read() == c.expected("analyzer_barcode_expr.tsv")
    assert open(a + "_allele.vcf").read() == c.expected("analyzer_allele.vcf") == ""


@pytest.mark.parametrize("seed,paired,flags", [(21, True, ["-s", "0.9"]), (22, False, ["-s", "0.8"]), (23, True, ["-s", "0.97", "-n", "3"])])
def test_analyzer_live_vs_reference_binary(built, tmp_path, seed, paired, flags):
    """the analyzer against the reference's analyzer run here, on reads with many multi-allele fragments (a lenient -s, a small -n that the
    summary must NOT apply -- but the analyzer's EM must, Genotyper.hpp:783-784 -- and single-end): _allele.vcf and _barcode_expr.tsv byte
    for byte, both analyzers in their default mode (--varMaxGroup 8)"""
    util.need(util.REF_ANALYZER)
    ref = os.path.join(str(tmp_path), "ref.fa")
    util.synth_ref("ref-rna", ref, genes=5, scale=0.05, seed=seed)
    pfx = os.path.join(str(tmp_path), "r")
    util.synth_reads(ref, pfx, pairs=4000, len=150, seed=seed, barcodes=120, sub=0.0)
    reads = ["-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] if paired else ["-u", pfx + "_1.fq"]
    g = os.path.join(str(tmp_path), "g")
    r = subprocess.run([GENO, "-f", ref] + reads + ["--barcode", pfx + "_bc.fa", "-o", g] + flags, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    aligned = ["-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa"] if paired else ["-u", g + "_aligned.fa"]
    outs = []
    for binary, tag in ((util.REF_ANALYZER, "ref"), (ANALYZER, "gpu")):
        o = os.path.join(str(tmp_path), tag)
        r = subprocess.run([binary, "-f", ref, "-a", g + "_allele.tsv"] + aligned + ["--barcode", g + "_aligned_bc.fa", "-o", o, "-t", "4"] + flags, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(o)
    assert open(outs[0] + "_allele.vcf").read() == open(outs[1] + "_allele.vcf").read()
    a, b = open(outs[0] + "_barcode_expr.tsv").read(), open(outs[1] + "_barcode_expr.tsv").read()
    assert a.count("\n") > 50
    assert a == b


def test_identical_read_end_collapse_stage(built, tmp_path):
    """t1k_reads_dedupe (Genotyper.cpp:451-480): equal sequences <=> equal distinct index, one representative per distinct sequence, and
    the assignment of the collapsed set with multiplicities gives the same lists and the same per-base coverage as assigning every
    read-end separately (the weight only feeds the coverage, SeqSet.hpp:2253-2274)"""
    c = goldens.Case("hla_synth_2x150", str(tmp_path))
    rng = np.random.default_rng(5)
    base = [s for _, _, s in t1k_amd.read_fastx(c.r1)][:150] + [s for _, _, s in t1k_amd.read_fastx(c.r2)][:150]
    extra = ["", "A", "ACGT", "ACGTN", "N" * 40, "ACGT" * 40, "ACGT" * 40 + "A", ("ACGT" * 40)[:-1] + "N"]     # lengths / N differ where the bases agree
    pool = base + extra
    reads = [pool[i] for i in rng.integers(0, len(pool), size=3000)]
    names, seqs, masks, _ = t1k_amd.load_reference_fasta(c.ref)
    full = t1k_amd.Context(ref_seq_similarity=0.97)
    full.ref_upload(seqs, masks)
    full.reads_upload(reads)
    full.assign()
    cnt_full, ovl_full = full.overlaps()
    cov_full = full.coverage()
    full.close()
    ctx = t1k_amd.Context(ref_seq_similarity=0.97)
    ctx.ref_upload(seqs, masks)
    ctx.reads_upload(reads)
    d = ctx.reads_dedupe()
    assert ctx.n_read_ends == len(set(reads))
    first = {}
    for i, r in enumerate(reads):
        assert first.setdefault(r, int(d[i])) == int(d[i])
    assert len(set(first.values())) == len(first)
    # numbered in the order of first use: the first m read-ends name exactly the distinct read-ends [0, D_m) -- what lets the job pair a
    # range of fragments as soon as a prefix of the assignment ranges is done (host/job.cpp)
    seen = 0
    for i in range(len(reads)):
        assert int(d[i]) <= seen
        seen = max(seen, int(d[i]) + 1)
    ctx.assign()
    cnt, ovl = ctx.overlaps()
    start_full = np.concatenate([[0], np.cumsum(cnt_full)]).astype(np.int64)
    start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    for i in range(0, len(reads), 7):   # every seventh read-end: its list equals the list of its distinct representative
        a = ovl_full[start_full[i]:start_full[i + 1]]
        b = ovl[start[int(d[i])]:start[int(d[i]) + 1]]
        assert len(a) == len(b)
        for f in ("seq_idx", "read_start", "read_end", "seq_start", "seq_end", "strand", "match_cnt", "left_clip", "right_clip", "relaxed_match_cnt", "similarity"):
            assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(ctx.coverage(), cov_full)
    ctx.close()


def test_device_coalescing_equals_host_restatement(built, tmp_path):
    """t1k_rowset_coalesce (the job's read-group table) against the host restatement of CoalesceReadAssignments fed with the rows of
    t1k_pair_batch: same groups in the same order, same start / end, float weights bit for bit"""
    import test_distributed_gloo as tdg
    for name in ("hla_synth_2x150", "cyp_dna_relax_2x150"):
        c = goldens.Case(name, str(tmp_path / name))
        kw = dict(ref_seq_similarity=float(c.flags[c.flags.index("-s") + 1]) if "-s" in c.flags else 0.8, relax_intron_align=1 if "--relaxIntronAlign" in c.flags else 0)
        extra = dict(allele_digit_units=1, allele_delimiter=".") if "cyp" in name else {}
        job = t1k_amd.Job(c.ref, **kw, **extra)
        job.load_reads(c.r1, c.r2, c.bc)
        job.run()
        g2, a2, p2, f2, e2 = tdg.parse_table(job.groups_serialize())
        job.close()
        r1 = [s for _, _, s in t1k_amd.read_fastx(c.r1)]
        r2 = [s for _, _, s in t1k_amd.read_fastx(c.r2)]
        if c.bc:
            keep = [i for i, (_, _, s) in enumerate(t1k_amd.read_fastx(c.bc)) if s != "missing_barcode"]
            r1, r2 = [r1[i] for i in keep], [r2[i] for i in keep]
        names, seqs, masks, _ = t1k_amd.load_reference_fasta(c.ref)
        ctx = t1k_amd.Context(**kw)
        ctx.ref_upload(seqs, masks)
        ctx.reads_upload([s for pr in zip(r1, r2) for s in pr])
        ctx.assign()
        n = len(r1)
        ctx.pair(np.arange(n) * 2, np.arange(n) * 2 + 1, [("N" in a) or ("N" in b) for a, b in zip(r1, r2)])
        counts, assigned, rows = ctx.rows()
        ctx.close()
        host = t1k_amd.Job(c.ref, device=-1, **extra)
        host.coalesce_rows(rows, counts)
        g1, a1, p1, f1, e1 = tdg.parse_table(host.groups_serialize())
        host.close()
        assert (g1, a1) == (g2, a2) and np.array_equal(p1, p2) and np.array_equal(f1, f2)
        for field in ("allele", "start", "end"):
            assert np.array_equal(e1[field], e2[field]), field
        assert np.array_equal(e1["w"].view(np.uint32), e2["w"].view(np.uint32)) and np.array_equal(e1["aw"].view(np.uint32), e2["aw"].view(np.uint32))



def analyzers_agree(tmp, ref, pfx, single=False, flags=(), geno_flags=(), env=None):
    """this build's genotyper, then both analyzers on its files: <prefix>_allele.vcf and <prefix>_barcode_expr.tsv byte for byte"""
    util.need(util.REF_ANALYZER)  # (where the reference binary is absent the test is skipped; the committed fixtures are compared in test_analyzer_variant_fixtures)
    g = os.path.join(tmp, "g")
    reads = ["-u", pfx + "_1.fq"] if single else ["-1", pfx + "_1.fq", "-2", pfx + "_2.fq"]
    r = subprocess.run([GENO, "-f", ref] + reads + ["--barcode", pfx + "_bc.fa", "-o", g] + list(geno_flags), stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    aligned = ["-u", g + "_aligned.fa"] if single else ["-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa"]
    common = ["-f", ref, "-a", g + "_allele.tsv"] + aligned + ["--barcode", g + "_aligned_bc.fa", "-t", "4"] + list(flags)
    outs = []
    for binary, tag in ((util.REF_ANALYZER, "ref"), (ANALYZER, "gpu")):
        o = os.path.join(tmp, tag + "".join(flags).replace("-", ""))
        r = subprocess.run([binary] + common + ["-o", o], stderr=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(o)
    vcf = [open(o + "_allele.vcf").read() for o in outs]
    table = [open(o + "_barcode_expr.tsv").read() for o in outs]
    assert vcf[0] == vcf[1]
    assert table[0] == table[1] and table[0].count("\n") > 5
    return vcf[0], table[0]


@pytest.mark.parametrize("het", [False, True])
def test_analyzer_on_a_sample_with_a_novel_snp(built, tmp_path, het):
    """SURVEY 8f row 2 with its VariantCaller: on a sample whose reads carry a consistent SNP absent from the database the reference's
    analyzer (default --varMaxGroup 8) calls the variant -- asserted, so that the case stays one where the stage matters -- and this
    build's analyzer writes the same VCF line(s) and the same per-barcode table; with --varMaxGroup 0 both write an empty VCF and the
    table of the raw assignment lists."""
    tmp = str(tmp_path)
    ref, pfx = util.novel_snp_sample(tmp, het)
    vcf, table = analyzers_agree(tmp, ref, pfx)
    assert vcf.count("\n") >= 1 and " 401 . " in vcf, vcf
    # the files the reference's own genotyper -> analyzer chain wrote for this seeded sample (tools/make_analyzer_variant_goldens.py)
    gold = os.path.join(util.GOLDEN, "analyzer_variants", "het" if het else "homo")
    assert vcf == open(gold + "_allele.vcf").read() and table == open(gold + "_barcode_expr.tsv").read()
    vcf0, table0 = analyzers_agree(tmp, ref, pfx, flags=["--varMaxGroup", "0"])
    assert vcf0 == "" and table0.count("\n") > 20


@pytest.mark.parametrize("case", ["several_per_gene", "genomic_reference", "single_end", "real_database"])
def test_analyzer_variant_calling_vs_reference_binary(built, tmp_path, case):
    """several unknown bases per gene on two alleles in three with sequencing errors on top (groups of two candidates, candidates expanded
    to the other selected alleles, fragments whose assignment list the called variants shorten); the same on a genomic reference (introns,
    exonic coordinates, ties that are written as FAIL pairs); a -u run.  --varMaxGroup 1 leaves the two-candidate groups unresolved."""
    tmp = str(tmp_path)
    if case == "several_per_gene":
        ref, pfx = util.several_snps_sample(tmp, 3)
        vcf, _ = analyzers_agree(tmp, ref, pfx)
        assert vcf.count("\n") >= 8
        vcf1, _ = analyzers_agree(tmp, ref, pfx, flags=["--varMaxGroup", "1"])
        assert vcf1 != vcf
    elif case == "genomic_reference":
        ref, pfx = util.several_snps_sample(tmp, 41, genes=3, kind="ref-dna", scale=0.05, positions=tuple(range(120, 2400, 97)), pairs=6000)
        vcf, _ = analyzers_agree(tmp, ref, pfx)
        assert vcf.count("\n") >= 4 and "FAIL" in vcf
    elif case == "real_database":  # the CYP2D6 genomic database of the reference's example: exon lists, its own allele-name structure, FAIL pairs
        ref, pfx = util.several_snps_sample(tmp, 83, kind=util.CYP_DNA, positions=tuple(range(140, 9000, 53)), pairs=5000, sub=0.001)
        vcf, _ = analyzers_agree(tmp, ref, pfx, flags=util.CYP_FLAGS, geno_flags=util.CYP_FLAGS)
        assert vcf.count("\n") >= 10 and "FAIL" in vcf
    else:
        ref, pfx = util.several_snps_sample(tmp, 29, genes=3, pairs=2500)
        vcf, table = analyzers_agree(tmp, ref, pfx, single=True)
        assert vcf.count("\n") >= 1
        # the variant pass in several pieces (re-assignment of the read-ends, their alignments): the same files
        vcf2, table2 = analyzers_agree(tmp, ref, pfx, single=True, flags=["-n", "1999"], env={"T1K_ANALYZER_PIECE": "600"})
        assert (vcf2, table2) == (vcf, table)


@pytest.mark.parametrize("het", [False, True])
@pytest.mark.parametrize("env", [{}, {"T1K_ANALYZER_NO_FAST": "1"}, {"T1K_VARIANTS_THREADS": "5", "T1K_ANALYZER_PIECE": "64"}])
def test_analyzer_variant_fixtures(built, tmp_path, het, env):
    """the same two samples without the reference binary at hand: this build's genotyper and analyzer against the committed files the
    reference's chain wrote (tests/golden/analyzer_variants, tools/make_analyzer_variant_goldens.py).  Round 6, three ways: as shipped (equal-length
    alignments with at most two mismatches get their edit strings from the host threads), every alignment through the device's traced DP
    (T1K_ANALYZER_NO_FAST), and with the booking sweeps of the variant caller threaded over five pieces of the fragments and pieces of 64 read-ends."""
    tmp = str(tmp_path)
    ref, pfx = util.novel_snp_sample(tmp, het)
    g, a = os.path.join(tmp, "g"), os.path.join(tmp, "a")
    r = subprocess.run([GENO, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "--barcode", pfx + "_bc.fa", "-o", g], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([ANALYZER, "-f", ref, "-a", g + "_allele.tsv", "-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa", "--barcode", g + "_aligned_bc.fa", "-o", a, "-t", "4"],
                       stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr
    gold = os.path.join(util.GOLDEN, "analyzer_variants", "het" if het else "homo")
    assert open(a + "_allele.vcf").read() == open(gold + "_allele.vcf").read()
    assert open(a + "_barcode_expr.tsv").read() == open(gold + "_barcode_expr.tsv").read()
