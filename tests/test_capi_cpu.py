"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/t1k_gpu.h declares, and the product fails
loudly (no CPU fallback) when no GPU is present; the executable keeps the reference's exit-code contract."""
import ctypes as C
import os
import re
import subprocess

import pytest

import util
import t1k_amd

GENO = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")


def declared_symbols():
    text = open(os.path.join(util.ROOT, "include", "t1k_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t1k_[a-z_0-9]+)\s*\(", text)) - {"t1k_allreduce_fn"})


def test_library_exports_every_declared_symbol(built):
    L = C.CDLL(t1k_amd.lib_path())
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "missing export " + n
    t1k_amd.lib()  # argtypes binding resolves every function it names


def test_comm_bind_accepts_a_job_and_a_context(monkeypatch):
    """bench.py binds its communicator to each step's Job and back to the anchor Context between steps (bind(Context)
    used to raise AttributeError on every rank): both owner kinds must reach t1k_comm_bind with their context handle"""
    import t1k_amd.capi as capi
    calls = []

    class FakeLib:
        def t1k_comm_bind(self, h, ctx):
            calls.append((h, ctx))
            return 0

        def t1k_job_ctx(self, h):
            return 1000 + h

    monkeypatch.setattr(capi, "lib", lambda: FakeLib())
    comm = capi.Comm.__new__(capi.Comm)
    comm.h = 7
    job = capi.Job.__new__(capi.Job)
    job.h = 5
    cx = capi.Context.__new__(capi.Context)
    cx.h = 22
    try:
        comm.bind(job)
        comm.bind(cx)
        assert calls == [(7, 1005), (7, 22)]
    finally:
        comm.h = job.h = cx.h = None  # nothing real to destroy


def test_defaults_match_reference(built):
    p = t1k_amd.JobParams()
    t1k_amd.lib().t1k_job_params_default(C.byref(p))
    assert (p.dev.kmer_length, p.dev.radius, p.dev.hit_len_required, p.dev.max_assign_cnt) == (11, 10, 31, 2000)  # Genotyper.cpp:207,218; SeqSet.hpp:763-764
    assert (p.dev.ref_seq_similarity, p.filter_frac, p.filter_cov, p.cross_gene_rate) == (0.8, 0.15, 1.0, 0.04)    # Genotyper.cpp:222-225


def have_gpu():
    return t1k_amd.lib().t1k_device_count() > 0


@pytest.mark.skipif(have_gpu(), reason="GPU present")
def test_no_gpu_fails_loudly(built, tmp_path):
    with pytest.raises(t1k_amd.T1kError):
        t1k_amd.Context()
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    with pytest.raises(t1k_amd.T1kError):
        t1k_amd.Job(ref)
    r = subprocess.run([GENO, "-f", ref, "-u", ref], stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "GPU" in r.stderr
    x = subprocess.run([os.path.join(util.ROOT, "t1k_amd", "bin", "fastq-extractor"), "-f", ref, "-u", ref, "-o", str(tmp_path / "x")], stderr=subprocess.PIPE, text=True)
    assert x.returncode != 0 and "no HIP device" in x.stderr


def test_executable_exit_codes(built, tmp_path):
    """Genotyper.cpp:199-203 (no arguments: usage, exit 0), 321-331 (unknown flag / missing -f: EXIT_FAILURE)."""
    r = subprocess.run([GENO], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "-f STRING" in r.stderr
    r = subprocess.run([GENO, "--noSuchFlag"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1
    r = subprocess.run([GENO, "-u", "x.fq"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Need to use -f" in r.stderr
    r = subprocess.run([GENO, "-f", str(tmp_path / "missing.fa"), "-u", "x.fq"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1  # clean failure instead of the reference's NULL dereference


def test_fastx_reader_conventions(built, tmp_path):
    p = tmp_path / "a.fq"
    p.write_text("@r1/1 extra words\nACGT\nNN\n+\nIIII\nII\n@r2\nTTTT\n+\nIIII\n>f3/2 7 0 3\nAC\nGT\n")
    recs = t1k_amd.read_fastx(str(p))
    assert recs == [("r1", "extra words", "ACGTNN"), ("r2", "", "TTTT"), ("f3", "7 0 3", "ACGT")]


def test_read_input_opens_without_a_job(built, tmp_path):
    """t1k_reads_open is host-only (no GPU, no job): fragment count of the mapped files, error text of a missing one"""
    a, b = tmp_path / "a_1.fq", tmp_path / "a_2.fq"
    a.write_text("@r1/1\nACGT\n+\nIIII\n@r2/1\nACGTA\n+\nIIIII\n")
    b.write_text("@r1/2\nTTTT\n+\nIIII\n@r2/2\nTTTTA\n+\nIIIII\n")
    r = t1k_amd.Reads(str(a), str(b))
    assert r.fragments() == 2
    r.close()
    with pytest.raises(t1k_amd.T1kError, match="missing.fq"):
        t1k_amd.Reads(str(tmp_path / "missing.fq"))
    # the committed fixtures as they lie (gzip): inflated, then indexed like a mapped file
    d = os.path.join(util.GOLDEN, "cyp_rna_2x100")
    r = t1k_amd.Reads(os.path.join(d, "reads_1.fq.gz"), os.path.join(d, "reads_2.fq.gz"))
    assert r.fragments() == 300
    r.close()


def test_reference_loader_merges_identical_sequences(built, tmp_path):
    p = tmp_path / "r.fa"
    p.write_text(">A*01 2 0 3 6 9\nACGTACGTAC\n>A*02 2 0 3 6 9\nACGTACGTAC\n>A*03\nACGTTTTTAC\n")
    names, seqs, masks, w = t1k_amd.load_reference_fasta(str(p))
    assert names == ["A*01", "A*03"] and w == [2, 1]
    assert masks[0].tolist() == [1, 1, 1, 1, 0, 0, 1, 1, 1, 1] and masks[1].tolist() == [1] * 10


def test_bench_input_plan_respects_scratch_disk(tmp_path):
    """bench.py at N > 1: every rank gets its own read set when the scratch disk holds them beside the output files, else the
    first k sets are reused (and the JSON line says how many were distinct); N = 1 is never touched."""
    import bench
    G = 1 << 30
    d = str(tmp_path)
    assert bench.distinct_input_files(d, 10_000_000, 1, free_bytes=1 * G) == 1
    assert bench.distinct_input_files(d, 10_000_000, 8, free_bytes=500 * G) == 8
    assert bench.distinct_input_files(d, 10_000_000, 4, free_bytes=71 * G) == 4
    k = bench.distinct_input_files(d, 10_000_000, 8, free_bytes=71 * G)
    assert 1 <= k < 8
    assert bench.distinct_input_files(d, 10_000_000, 8, free_bytes=5 * G) == 1
