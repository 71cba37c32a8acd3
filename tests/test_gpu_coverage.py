"""GPU tests (pytest -m gpu) of the per-base coverage modes.  posWeight (SeqSet.hpp:2253-2274) is read through
GetSeqMissingBaseCoverage (2717-2755) for the alleles on selection's candidate lists only (Genotyper.hpp:1754, 1870-1878), so a job keeps
the windows' overlap lists and adds coverage for those alleles inside select() (t1k_coverage_selected) instead of for all alleles in
every range.  Checked here: the stage against the eager per-range updates base by base, and the executable in every mode (deferred,
eager, a memory budget that makes later windows fall back to eager, small batches, two ranks) against the reference binary on
samples mixed from several individuals -- genes with more than two allele types are where missingCoverage decides the call."""
import os
import re
import subprocess

import numpy as np
import pytest

import util
import t1k_amd
import test_oracle_golden as tog

pytestmark = pytest.mark.gpu
GENO = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")


def _reads(prefix):
    return [s for _, _, s in t1k_amd.read_fastx(prefix + "_1.fq")] + [s for _, _, s in t1k_amd.read_fastx(prefix + "_2.fq")]


@pytest.mark.parametrize("kind,sim,relax,batch", [("rna", 0.9, 0, None), ("rna", 0.97, 0, "64"), ("dna", 0.9, 1, None), ("dna", 0.8, 1, "100")])
def test_coverage_selected_stage_equals_eager_updates(built, tmp_path, kind, sim, relax, batch):
    """t1k_assign_range in deferred mode leaves the coverage arrays untouched and produces the same overlap lists (relaxed counts
    included); t1k_coverage_selected then yields, for every selected allele, exactly the per-base coverage the eager mode accumulates,
    and nothing for the others.  Weighted read-ends, indels (traced DP alignments), several gather batches."""
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref("ref-" + kind, ref, genes=4, scale=0.04, seed=31)
    pfx = os.path.join(tmp, "r")
    util.synth_reads(ref, pfx, pairs=180, len=150, seed=32, sub=0.006, indel=0.004, fragmean=420 if kind == "dna" else 350)
    reads = _reads(pfx)
    rng = np.random.default_rng(7)
    weights = rng.integers(1, 6, size=len(reads)).astype(np.uint32)
    names, seqs, masks, _ = t1k_amd.load_reference_fasta(ref)
    kw = dict(ref_seq_similarity=sim, relax_intron_align=relax)
    eager = t1k_amd.Context(**kw)
    eager.ref_upload(seqs, masks)
    eager.reads_upload(reads, weights)
    eager.assign()
    cnt_e, ovl_e = eager.overlaps()
    cov_e = eager.coverage()
    assert cov_e.sum() > 0
    lazy = t1k_amd.Context(**kw)
    lazy.set_coverage_mode(True)
    lazy.ref_upload(seqs, masks)
    lazy.reads_upload(reads, weights)
    lazy.assign()
    cnt_d, ovl_d = lazy.overlaps()
    assert np.array_equal(cnt_e, cnt_d) and ovl_e.tobytes() == ovl_d.tobytes()
    assert not lazy.coverage().any()
    rs = lazy.detach_readset()
    assert rs.size() == len(reads) and rs.bytes() > 0
    sel = (rng.random(len(seqs)) < 0.3).astype(np.uint8)
    covered = [a for a in range(len(seqs)) if cov_e[sum(len(s) for s in seqs[:a]):sum(len(s) for s in seqs[:a + 1])].any()]
    sel[covered[:3]] = 1
    if batch:
        os.environ["T1K_COVER_BATCH"] = batch
    try:
        n = lazy.coverage_selected(rs, sel)
    finally:
        os.environ.pop("T1K_COVER_BATCH", None)
    assert n > 0
    cov_d = lazy.coverage()
    off = 0
    for a, s in enumerate(seqs):
        want = cov_e[off:off + len(s)] if sel[a] else np.zeros(len(s), np.int32)
        assert np.array_equal(cov_d[off:off + len(s)], want), "allele %d (%s)" % (a, "selected" if sel[a] else "not selected")
        off += len(s)
    # a second call adds the same again (the arrays are sums): the read set is still intact
    assert lazy.coverage_selected(rs, sel) == n
    assert np.array_equal(lazy.coverage(), 2 * cov_d)
    rs.close(); lazy.close(); eager.close()


MODES = {
    "deferred": {},
    "eager": {"T1K_COVERAGE": "eager"},
    "budget_fallback": {"T1K_FIRST_WINDOW": "16", "T1K_WINDOW": "64", "T1K_WINDOW_GROWTH": "1", "T1K_BATCH": "16", "T1K_PAIR_BATCH": "16", "T1K_ARCHIVE_GB": "0.0000001"},
    # the rule of round 4 (a window's read set is kept only beside what is allocated, the windows in flight, the job's projected rows): the
    # device "ends" 1 MB above what is allocated at the first decision after a paired fragment, so the later windows update coverage per range
    "memory_rule_fallback": {"T1K_FIRST_WINDOW": "16", "T1K_WINDOW": "64", "T1K_WINDOW_GROWTH": "1", "T1K_BATCH": "16", "T1K_PAIR_BATCH": "16", "T1K_TEST_ARCHIVE_HEADROOM_MB": "1"},
    "small_windows_small_batches": {"T1K_FIRST_WINDOW": "24", "T1K_WINDOW": "96", "T1K_BATCH": "16", "T1K_PAIR_BATCH": "32", "T1K_COVER_BATCH": "64", "T1K_PIPELINES": "2"},
    "hash_order_small_batches": {"T1K_DISTINCT_ORDER": "hash", "T1K_FIRST_WINDOW": "32", "T1K_WINDOW": "128", "T1K_BATCH": "16", "T1K_PAIR_BATCH": "16"},
    "first_use_order_small_batches": {"T1K_FIRST_WINDOW": "32", "T1K_WINDOW": "128", "T1K_BATCH": "16", "T1K_PAIR_BATCH": "16", "T1K_PIPELINES": "4"},
    # the alignment queues' stripes start at 8 entries: they overflow AFTER k_fullalign has added the range's ungapped coverage; the phase
    # takes that back and the range runs again with larger stripes (this used to end the job with T1K_ERR_COMMITTED)
    "eager_queue_overflow": {"T1K_COVERAGE": "eager", "T1K_TEST_SMALL_QUEUES": "1"},
    "deferred_queue_overflow": {"T1K_TEST_SMALL_QUEUES": "1", "T1K_COVER_BATCH": "64"},
    # every group of at least 34 fragments through the four-wavefront fold of the large groups (k_co_reduce_long; 4096 by default)
    "long_group_fold": {"T1K_CO_LONG_RUN": "34"},
    "long_group_fold_two_ranks": {"T1K_CO_LONG_RUN": "40", "T1K_GPUS": "0,0"},
    # no device-memory pool: a freed block goes back to the driver at once, so a kept window's list table that points into an EARLIER
    # window's overlap-store chunks (identical read-ends linked across windows) faults or reads garbage if that window's read set was
    # released before the later one is scanned (every kept set stays alive until all are scanned)
    "no_pool_small_windows": {"T1K_POOL_GB": "0", "T1K_FIRST_WINDOW": "24", "T1K_WINDOW": "96", "T1K_BATCH": "16", "T1K_PAIR_BATCH": "32", "T1K_COVER_BATCH": "64"},
    # k_pair: every fragment with more than 16 overlaps goes to the second launch (scratch from the big arena), whose 64 entries force the
    # "arena grows to the demand the device counted, the call runs again" path
    "pair_second_launch_arena_growth": {"T1K_PAIR_FRAGCAP": "16", "T1K_PAIR_BIGCAP": "64", "T1K_PAIR_BATCH": "32"},
    # k_pair with the joined fragments materialised in the workgroup's scratch (round 3's form; the default streams the join twice instead)
    "pair_list_form": {"T1K_PAIR_LIST": "1", "T1K_PAIR_BATCH": "64"},
    "two_ranks": {"T1K_GPUS": "0,0"},
    "three_ranks_small_windows": {"T1K_GPUS": "0,0,0", "T1K_FIRST_WINDOW": "16", "T1K_WINDOW": "48", "T1K_BATCH": "16"},
}


@pytest.mark.parametrize("case", range(len(tog.LIVE_TABLE_CASES)))
def test_coverage_modes_on_mixed_samples_vs_reference_binary(built, tmp_path, case):
    """every coverage mode of the job against the reference binary, on samples mixed from several simulated individuals (genes with
    three and more allele types: Genotyper.hpp:1697-1996 reads missingCoverage there) with the selection options of the oracle's live
    cases, --relaxIntronAlign among them: all output files byte for byte"""
    util.need(util.REF_BIN)
    kind, genes, scale, parts, length, flags = tog.LIVE_TABLE_CASES[case]
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref(kind, ref, genes=genes, scale=scale, seed=900 + case)
    for p in range(parts):
        util.synth_reads(ref, os.path.join(tmp, "p%d" % p), pairs=120 + 40 * case, len=length, seed=1000 + 10 * case + p, sub=0.004)
    for m in ("1", "2"):
        with open(os.path.join(tmp, "r_%s.fq" % m), "w") as o:
            for p in range(parts):
                o.write(open(os.path.join(tmp, "p%d_%s.fq" % (p, m))).read())
    args = ["-f", ref, "-1", os.path.join(tmp, "r_1.fq"), "-2", os.path.join(tmp, "r_2.fq")] + flags
    a = os.path.join(tmp, "ref_out")
    subprocess.run([util.REF_BIN] + args + ["-o", a, "-t", "2"], check=True, stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    asked = {}
    for mode, env in MODES.items():
        b = os.path.join(tmp, mode)
        r = subprocess.run([GENO] + args + ["-o", b], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_DEBUG_PHASES="1", **env))
        assert r.returncode == 0, (mode, r.stderr[-2000:])
        for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
            assert open(a + suf, "rb").read() == open(b + suf, "rb").read(), (mode, suf)
        m = re.findall(r"coverage of the (\d+) alleles on selection's lists: (\d+) records", r.stderr)
        asked[mode] = [(int(x), int(y)) for x, y in m]
        kept = re.findall(r"read sets of (\d+) of (\d+) windows kept", r.stderr)
        if mode == "eager":
            assert not m and not kept
        if mode == "eager_queue_overflow" and case in (0, 1, 2, 3):  # (the other cases queue too few alignments to fill a stripe)
            assert "the range's coverage was taken back" in r.stderr and "runs again with stripes" in r.stderr, r.stderr[-1500:]
        if mode == "budget_fallback":
            assert kept and all(0 < int(k) < int(w) for k, w in kept), r.stderr[-1500:]
        if mode == "memory_rule_fallback":
            assert "read sets are not kept from here on" in r.stderr and kept and all(0 < int(k) < int(w) for k, w in kept), r.stderr[-1500:]
    if case in (0, 1):  # three individuals, permissive filters: some gene carries more than two types, so selection asked for coverage
        assert asked["deferred"] and asked["deferred"][0][0] > 0 and asked["deferred"][0][1] > 0, asked


def test_barcode_config_at_size_vs_committed_reference_hashes(built, tmp_path):
    """BASELINE configs[4] on one GPU, at size: 1 M 2x150 bp pairs carrying ~82 k distinct 10x-style barcodes (log-uniform usage),
    genotyper -> analyzer.  Expected = md5 of the files the REFERENCE binaries wrote for this very input on an MI355X host
    (tools/full_size_parity_r03.sh, profiles/r03_full_size_parity.log; 103 s + 11 s there), committed as tests/golden/full_size_md5.json."""
    import hashlib
    import json
    want = json.load(open(os.path.join(util.GOLDEN, "full_size_md5.json")))["barcode_1M_100k"]
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "hla.fa")
    util.synth_ref("ref-rna", ref, genes=24, scale=1.0, seed=20250614)
    pfx = os.path.join(tmp, "b")
    util.synth_reads(ref, pfx, pairs=want["pairs"], len=150, seed=want["seed"], barcodes=want["barcodes"], sub=want["sub"])
    g = os.path.join(tmp, "g")
    r = subprocess.run([GENO, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "--barcode", pfx + "_bc.fa", "-s", "0.97", "-o", g], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

    def md5(path):
        h = hashlib.md5()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa", "_aligned_bc.fa"):
        assert md5(g + suf) == want[suf], suf
    a = os.path.join(tmp, "a")
    r = subprocess.run([os.path.join(util.ROOT, "t1k_amd", "bin", "analyzer"), "-f", ref, "-a", g + "_allele.tsv", "-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa",
                        "--barcode", g + "_aligned_bc.fa", "-s", "0.97", "-o", a, "--varMaxGroup", "0"], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert md5(a + "_barcode_expr.tsv") == want["analyzer_barcode_expr.tsv"]
    assert os.path.getsize(a + "_allele.vcf") == want["analyzer_vcf_bytes"] == 0


def test_barcode_config_at_its_stated_size_vs_committed_reference_hashes(built, tmp_path):
    """BASELINE configs[4] AT ITS SIZE on one GPU: 10 M 2x150 bp pairs carrying 100 k 10x-style barcodes (99 995 distinct in the
    file), -s 0.97, genotyper -> analyzer in its DEFAULT mode (--varMaxGroup 8: the variant pass runs over all 10 M fragments).  Expected = md5 of the
    files the REFERENCE binaries (oracle/_ref/genotyper -t 32: 2 234 s, 186 GB; oracle/_ref/analyzer -t 64: 113 s) wrote for this very input on an MI355X
    host (tools/barcode_10M_r06.sh, profiles/r06_barcode_10M.log), committed as tests/golden/full_size_md5.json: barcode_10M_100k.  The input is
    bench.py's own (`python bench.py --barcodes 100000`).  Genotyper.cpp:372-392, 709-718; BarcodeSummary.hpp:24-80; Analyzer.cpp:611-696."""
    import hashlib
    import json
    want = json.load(open(os.path.join(util.GOLDEN, "full_size_md5.json")))["barcode_10M_100k"]
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "hla.fa")
    util.synth_ref("ref-rna", ref, genes=24, scale=1.0, seed=20250614)
    pfx = os.path.join(tmp, "b")
    util.synth_reads(ref, pfx, pairs=want["pairs"], len=150, seed=want["seed"], barcodes=want["barcodes"])
    g = os.path.join(tmp, "g")
    r = subprocess.run([GENO, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "--barcode", pfx + "_bc.fa"] + want["flags"].split() + ["-o", g], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

    def md5(path):
        h = hashlib.md5()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa", "_aligned_bc.fa"):
        assert md5(g + suf) == want[suf], suf
    assert open(g + "_genotype.tsv").read() == open(os.path.join(util.GOLDEN, "barcode_10M_ref_genotype.tsv")).read()   # (the reference's table itself is committed)
    for f in (pfx + "_1.fq", pfx + "_2.fq", pfx + "_bc.fa"):
        os.remove(f)
    a = os.path.join(tmp, "a")
    r = subprocess.run([os.path.join(util.ROOT, "t1k_amd", "bin", "analyzer"), "-f", ref, "-a", g + "_allele.tsv", "-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa",
                        "--barcode", g + "_aligned_bc.fa"] + want["flags"].split() + ["-o", a], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert md5(a + "_barcode_expr.tsv") == want["analyzer_barcode_expr.tsv"]
    assert md5(a + "_allele.vcf") == want["analyzer_allele.vcf"]


def test_kir_wgs_config_at_size_vs_committed_reference_hashes(built, tmp_path):
    """BASELINE configs[2] on one GPU, at size: 10 M 2x150 bp pairs against the KIR-like dna reference with the kir-wgs preset
    (-s 0.9 --relaxIntronAlign, run-t1k:300-304): the near-best alignments run in every range for the relaxed counts and their
    coverage is added behind selection.  Expected = md5 of the files the REFERENCE binary wrote for this very input on an MI355X host
    (tools/kir_10M_parity_r03.sh, profiles/r03_kir_10M_parity.log: 243 s at -t 32), committed in tests/golden/full_size_md5.json."""
    import hashlib
    import json
    want = json.load(open(os.path.join(util.GOLDEN, "full_size_md5.json")))["kir_10M"]
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "kir.fa")
    util.synth_ref("ref-dna", ref, genes=want["genes"], scale=1.0, seed=20250614)
    pfx = os.path.join(tmp, "k")
    util.synth_reads(ref, pfx, pairs=want["pairs"], len=150, seed=want["seed"])
    g = os.path.join(tmp, "g")
    r = subprocess.run([GENO, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] + want["flags"].split() + ["-o", g], stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, T1K_DEBUG_PHASES="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "coverage deferred to selection" in r.stderr
    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        h = hashlib.md5()
        with open(g + suf, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        assert h.hexdigest() == want[suf], suf


@pytest.fixture(scope="module")
def hla_10M_input(tmp_path_factory):
    """the bench's own sample (BASELINE's headline workload): 10 M 2x150 bp pairs against the HLA-like rna reference, seed 2 -- the input the
    committed `hla_10M` hashes of the reference binary belong to"""
    import json
    want = json.load(open(os.path.join(util.GOLDEN, "full_size_md5.json")))["hla_10M"]
    tmp = str(tmp_path_factory.mktemp("hla10m"))
    ref = os.path.join(tmp, "hla.fa")
    util.synth_ref("ref-rna", ref, genes=24, scale=1.0, seed=20250614)
    pfx = os.path.join(tmp, "r")
    util.synth_reads(ref, pfx, pairs=want["pairs"], len=150, seed=want["seed"])
    return want, ref, pfx, tmp


@pytest.mark.parametrize("mode", ["read_sets_not_kept", "two_ranks_rank_local_input", "two_ranks_allreduce_em"])
def test_hla_10M_per_rank_paths_of_the_sharded_config_vs_committed_reference_hashes(built, hla_10M_input, mode):
    """BASELINE configs[3] (50 M HLA pairs over 8 GPUs) cannot run here; the two code paths its ranks depend on can, on the 10 M-pair
    workload whose reference hashes are committed, so that they are compared with the REFERENCE's files and not with this build's own
    eager run: (a) windows that lose their kept read set under the memory rule and fall back to the per-range coverage updates (at 50 M
    pairs most windows do: profiles/r04_size_curve.log) -- forced here with a 2 GB budget for the kept sets and windows of 1.5 M fragments; (b) rank-local input:
    two in-process ranks on the one device, each indexing and writing only its own fragments (T1K_SHARD_INPUT), coverage all-reduce,
    row exchange, group gather and the sharded E-step through the in-process communicator (Genotyper.cpp:523-621, SeqSet.hpp:2253-2274);
    (c) the same two ranks with the EM collective BASELINE.json's north_star names (T1K_EM_COLLECTIVE=allreduce: every rank adds up its own
    read groups' contributions per class, E doubles are all-reduced per EM update, Genotyper.hpp:372-421) -- the sums are re-associated, so
    the stated tolerance applies instead of the hashes: the REFERENCE's own _genotype.tsv / _allele.tsv for this input (committed:
    tests/golden/hla_10M_ref_genotype.tsv / _allele.tsv, md5 = full_size_md5.json's) with identical calls and qualities, abundances within
    1e-4 relative, the same number of EM iterations; the aligned-read files still byte for byte."""
    import hashlib
    want, ref, pfx, tmp = hla_10M_input
    env = dict(os.environ, T1K_DEBUG_PHASES="1")
    if mode == "read_sets_not_kept":
        env.update(T1K_ARCHIVE_GB="2", T1K_WINDOW="1500000")  # (windows of at most 1.5 M fragments: the first ones, cut before any set's size is known, are kept; the rest is not)
    else:
        env.update(T1K_GPUS="0,0", T1K_SHARD_INPUT="1")
    if mode == "two_ranks_allreduce_em":
        env.update(T1K_EM_COLLECTIVE="allreduce")
    g = os.path.join(tmp, "g_" + mode)
    r = subprocess.run([GENO, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] + want["flags"].split() + ["-o", g], stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    if mode == "read_sets_not_kept":
        import re
        m = re.search(r"read sets of (\d+) of (\d+) windows kept", r.stderr)
        assert m and int(m.group(1)) * 2 <= int(m.group(2)), r.stderr[-1500:]   # most windows took the per-range coverage path
    if mode == "two_ranks_allreduce_em":
        import re

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
        for suf in ("_genotype.tsv", "_allele.tsv"):
            ref_file = os.path.join(util.GOLDEN, "hla_10M_ref" + suf)
            assert hashlib.md5(open(ref_file, "rb").read()).hexdigest() == want[suf]   # the committed file IS the reference's
            got, exp = open(g + suf).read().splitlines(), open(ref_file).read().splitlines()
            assert len(got) == len(exp), suf
            for a, b in zip(got, exp):
                assert close(a.split("\t"), b.split("\t")), (suf, a, b)
            os.remove(g + suf)
        assert re.search(r"in (\d+) EM iterations", r.stderr).group(1) == str(want["reference_run"].get("em_iterations", 11))
    for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
        if not os.path.exists(g + suf):
            continue
        h = hashlib.md5()
        with open(g + suf, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        assert h.hexdigest() == want[suf], (mode, suf)
        os.remove(g + suf)
