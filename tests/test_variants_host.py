"""Novel-variant calling of the analyzer stage (t1k_amd/csrc/host/variants.cpp behind t1k_variants_call; host code, no GPU): against the
REFERENCE's analyzer (oracle/_ref/analyzer, VariantCaller.hpp) on samples whose reads carry SNPs the database does not know.

The reference analyzer is a whole program; what its VariantCaller is handed -- every fragment's assignment list with both read-ends'
overlaps and their edit strings, and the allele abundances of the analyzer's EM -- is produced here by the CPU restatement
(oracle/t1k_oracle_cli --fragDump, the checker of the assignment path), given to the product's host code through the C ABI, and the two
files that depend on it are compared with the reference analyzer's own: <prefix>_allele.vcf byte for byte, and <prefix>_barcode_expr.tsv
rebuilt from the adjusted assignment lists (BarcodeSummary::AddFragment, BarcodeSummary.hpp:24-57)."""
import os
import subprocess

import numpy as np
import pytest

import util
import t1k_amd


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    if not os.path.exists(t1k_amd.lib_path()):
        g.build()
    return True


def reference_run(tmp, ref, pfx, geno_flags=(), ana_flags=(), single=False):
    """reference genotyper -> reference analyzer (default --varMaxGroup 8 unless ana_flags say otherwise)"""
    util.need(util.REF_BIN)
    util.need(util.REF_ANALYZER)
    g = os.path.join(tmp, "g")
    reads = ["-u", pfx + "_1.fq"] if single else ["-1", pfx + "_1.fq", "-2", pfx + "_2.fq"]
    r = subprocess.run([util.REF_BIN, "-f", ref] + reads + ["--barcode", pfx + "_bc.fa", "-o", g, "-t", "8"] + list(geno_flags), stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    aligned = ["-u", g + "_aligned.fa"] if single else ["-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa"]
    a = os.path.join(tmp, "ana")
    r = subprocess.run([util.REF_ANALYZER, "-f", ref, "-a", g + "_allele.tsv"] + aligned + ["--barcode", g + "_aligned_bc.fa", "-o", a, "-t", "4"] + list(ana_flags),
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    return g, a, aligned


def oracle_dump(tmp, ref, g, aligned, flags=()):
    """the analyzer's view of the sample from the CPU restatement: the selected alleles only (Genotyper::InitRefSet with selectedAlleles,
    Genotyper.hpp:732-757 = the reference file without the other records), assignment lists with edit strings, abundances"""
    selected = set(line.split()[0] for line in open(g + "_allele.tsv") if line.strip())
    sel = os.path.join(tmp, "selected.fa")
    with open(sel, "w") as o:
        for name, head, seq in util.read_fa(ref):
            if name in selected:
                o.write(head + "\n" + seq + "\n")
    out = os.path.join(tmp, "orc")
    r = subprocess.run([util.ORACLE_CLI, "-f", sel] + aligned + ["-o", out, "--fragDump", "--dumpOverlaps"] + list(flags), stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    names = [line.split("\t")[0] for line in open(out + "_abundance.tsv")]
    abundance = [float(line.split("\t")[1]) for line in open(out + "_abundance.tsv")]
    return sel, out, names, abundance


def parse_dump(path, n_frag):
    """<o>_fragdump.tsv -> (asg_ptr, asg, ops)"""
    rows, ops = [], []
    per = [0] * n_frag

    def overlap(f, allele):
        e = [] if f[10] == "-" else [int(c) for c in f[10]]
        o = (allele, int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[7]), int(f[8]), int(f[9]), float(f[6]))
        at = len(ops)
        ops.extend(e)
        return o, at, len(e)
    zero = (0,) * 10 + (0.0,)
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        frag, allele, mate, from2 = int(f[0]), int(f[1]), int(f[2]), int(f[3])
        o1, at1, n1 = overlap(f[4:15], allele)
        o2, at2, n2 = overlap(f[15:26], allele) if mate else (zero, 0, 0)
        rows.append((allele, mate, from2, 0, o1, o2, at1, at2, n1, n2))
        per[frag] += 1
    asg = np.array(rows, dtype=t1k_amd.FRAG_ASG_DTYPE) if rows else np.zeros(0, dtype=t1k_amd.FRAG_ASG_DTYPE)
    ptr = np.zeros(n_frag + 1, dtype=np.uint64)
    ptr[1:] = np.cumsum(per)
    return ptr, asg, np.array(ops, dtype=np.int8)


def check_details(out, ids, ptr, asg, single):
    """t1k_fragment_details on the read-ends' overlap lists (<o>_overlaps.tsv: AssignRead's lists, the order t1k_overlaps_download returns)
    must name, for every allele a fragment kept, the overlaps the restatement's pairing chose"""
    at = {name: i for i, name in enumerate(ids)}
    lists = {}
    for line in open(out + "_overlaps.tsv"):
        f = line.rstrip("\n").split("\t")
        lists.setdefault((at[f[0]], int(f[1])), []).append((int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]), int(f[10]), int(f[11]), int(f[12]), float(f[9])))
    checked = dangling = 0
    for f in range(len(ids)):
        lo, hi = int(ptr[f]), int(ptr[f + 1])
        if hi == lo:
            continue
        want = asg[lo:hi]
        got = t1k_amd.fragment_details(lists.get((f, 1), []), None if single else lists.get((f, 2), []), want["allele_idx"], paired=not single)
        for field in ("allele_idx", "has_mate_pair", "o1_from_r2", "o1"):
            assert np.array_equal(got[field], want[field]), (f, field, got[field], want[field])
        mates = want["has_mate_pair"] != 0
        assert np.array_equal(got["o2"][mates], want["o2"][mates]), f
        checked += hi - lo
        dangling += int((want["o1_from_r2"] != 0).sum())
    return checked, dangling


def barcode_table(names, barcodes, ptr, asg, keep_of):
    """BarcodeSummary::AddFragment / Output (BarcodeSummary.hpp:24-80) over the adjusted lists"""
    ids = {}
    for b in barcodes:  # ids in order of first appearance over all loaded fragments (Analyzer.cpp:380-392)
        ids.setdefault(b, len(ids))
    A = len(names)
    table = {}
    for f, b in enumerate(barcodes):
        lo, hi = int(ptr[f]), int(ptr[f + 1])
        if hi == lo:
            continue  # fragmentAssigned is false (Analyzer.cpp:692-696)
        slot = table.setdefault(ids[b], ([0.0] * A, [0] * A))
        kept = [int(asg["allele_idx"][lo + i]) for i in range(hi - lo) if keep_of(f)[i]]
        for a in kept:
            slot[0][a] += 1.0 / len(kept)
            if len(kept) == 1:
                slot[1][a] += 1
    by_id = {v: k for k, v in ids.items()}
    out = "#barcode" + "".join("\t" + n for n in names) + "".join("\t%s_uniq" % n for n in names) + "\n"
    for i in sorted(table):
        out += by_id[i] + "".join("\t%f" % x for x in table[i][0]) + "".join("\t%d" % x for x in table[i][1]) + "\n"
    return out


HOST = os.path.join(util.ROOT, "t1k_amd", "csrc", "host")
_SAN = {}


def sanitizer_harness(tmp, kind="address,undefined"):
    """tests/harness/variants_san.cpp + host/variants.cpp under -fsanitize=<kind> (None: no such sanitizer runtime here)"""
    if kind not in _SAN:
        exe = os.path.join(tmp, "variants_san_" + kind.split(",")[0])
        r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=" + kind] + (["-fno-sanitize-recover=undefined"] if "undefined" in kind else []) +
                           ["-std=c++17", "-o", exe, os.path.join(util.ROOT, "tests", "harness", "variants_san.cpp"), os.path.join(HOST, "variants.cpp"), "-lpthread"],
                           stderr=subprocess.PIPE, text=True)
        _SAN[kind] = exe if r.returncode == 0 else None
    return _SAN[kind]


def run_under_sanitizers(tmp, sel, abundance, ptr, asg, ops, r1, r2, var_max_group, want_vcf, keep_of):
    """the same case through the sanitizer build: same VCF text, same kept flags, no report"""
    exes = [e for e in (sanitizer_harness(tmp), sanitizer_harness(tmp, "thread")) if e]
    if not exes:
        return False
    names, seqs, masks, _ = t1k_amd.load_reference_fasta(sel)
    genes = {}
    path = os.path.join(tmp, "case.txt")

    def ov(o, at, n):
        e = "".join(str(int(c)) for c in ops[int(at):int(at) + int(n)]) or "-"
        return "%d %d %d %d %d %d %d %d %d %d %.17g %s" % (o["seq_idx"], o["read_start"], o["read_end"], o["seq_start"], o["seq_end"], o["strand"], o["match_cnt"], o["left_clip"],
                                                              o["right_clip"], o["relaxed_match_cnt"], o["similarity"], e)
    with open(path, "w") as f:
        f.write("%d\n" % len(names))
        for n, sq, m, ab in zip(names, seqs, masks, abundance):
            f.write("%s %d %.17g %s %s\n" % (n, genes.setdefault(n.split("*")[0], len(genes)), ab, sq, "".join("1" if x else "0" for x in m)))
        f.write("%d %d\n" % (len(r1), 0 if r2 is None else 1))
        for i in range(len(r1)):
            lo, hi = int(ptr[i]), int(ptr[i + 1])
            f.write("%d %s%s\n" % (hi - lo, r1[i] or "-", "" if r2 is None else " " + (r2[i] or "-")))
            for a in asg[lo:hi]:
                f.write("%d %d %d %s%s\n" % (a["allele_idx"], a["has_mate_pair"], a["o1_from_r2"], ov(a["o1"], a["ops1"], a["n_ops1"]),
                                              " " + ov(a["o2"], a["ops2"], a["n_ops2"]) if a["has_mate_pair"] else ""))
    for exe in exes:  # (the thread sanitizer: the sweeps on threads that own the alleles)
        r = subprocess.run([exe, path, str(var_max_group)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_VARIANTS_THREADS="4"))
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[:3000]
        out = r.stdout.split("\n")
        nv = int(out[0])
        assert "\n".join(out[1:1 + nv]) + ("\n" if nv else "") == want_vcf
        flags = out[1 + nv:1 + nv + len(r1)]
        for i in range(len(r1)):
            lo, hi = int(ptr[i]), int(ptr[i + 1])
            assert flags[i] == ("".join(str(int(k)) for k in keep_of(i)) if hi > lo else "-"), i
    return len(exes)


LAST = {}  # what the last run_case saw (tests assert that their sample holds what they are about)


def run_case(tmp, ref, pfx, single=False, var_max_group=8, geno_flags=(), ana_flags=(), orc_flags=(), job_kw=None, sanitize=False):
    g, a, aligned = reference_run(tmp, ref, pfx, geno_flags, ana_flags, single)
    sel, out, names, abundance = oracle_dump(tmp, ref, g, aligned, orc_flags)
    r1 = [s for _, _, s in t1k_amd.read_fastx(aligned[1])]
    r2 = None if single else [s for _, _, s in t1k_amd.read_fastx(aligned[3])]
    bcs = [s for _, _, s in t1k_amd.read_fastx(g + "_aligned_bc.fa")]
    ptr, asg, ops = parse_dump(out + "_fragdump.tsv", len(r1))
    ids = [i for i, _, _ in t1k_amd.read_fastx(aligned[1])]
    assert len(set(ids)) == len(ids)
    checked, dangling = check_details(out, ids, ptr, asg, single)
    assert checked == len(asg)
    LAST.update(assignments=len(asg), from_r2=dangling, unpaired=int((asg["has_mate_pair"] == 0).sum()), zero_abundance=sum(1 for x in abundance if x == 0))
    job = t1k_amd.Job(sel, device=-1, **(job_kw or {}))
    v = job.call_variants(abundance, var_max_group, ptr, asg, ops, r1, r2)
    want_vcf = open(a + "_allele.vcf").read()
    got_vcf = v.vcf()
    # the sweeps that large inputs run on threads owning the alleles (base tables, fragment links): the same variants, the same numbers
    os.environ["T1K_VARIANTS_THREADS"] = "3"
    try:
        v3 = job.call_variants(abundance, var_max_group, ptr, asg, ops, r1, r2)
        assert v3.vcf() == got_vcf and v3.records().tobytes() == v.records().tobytes()
        v3.close()
    finally:
        del os.environ["T1K_VARIANTS_THREADS"]
    cache = {}

    def keep_of(f):
        if f not in cache:
            lo, hi = int(ptr[f]), int(ptr[f + 1])
            cache[f] = v.adjust(asg[lo:hi], ops, r1[f], None if single else r2[f])
        return cache[f]
    got_table = barcode_table(names, bcs, ptr, asg, keep_of)
    want_table = open(a + "_barcode_expr.tsv").read()
    moved = sum(1 for f in cache if not cache[f].all())
    LAST["sanitized"] = sanitize and got_vcf == want_vcf and run_under_sanitizers(tmp, sel, abundance, ptr, asg, ops, r1, r2, var_max_group, want_vcf, keep_of)
    recs = v.records()
    v.close()
    job.close()
    return want_vcf, got_vcf, want_table, got_table, moved, recs


@pytest.mark.parametrize("het", [False, True])
def test_called_variants_and_adjusted_counts_vs_reference_analyzer(built, tmp_path, het):
    """one consistent exonic SNP absent from the database (in every allele of a gene, or in every second one): the VCF line(s) the
    reference writes, byte for byte, and its per-barcode table"""
    util.need(util.ORACLE_CLI)
    tmp = str(tmp_path)
    ref, pfx = util.novel_snp_sample(tmp, het)
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx)
    assert want_vcf.count("\n") >= 1 and " 401 . " in want_vcf, want_vcf
    assert got_vcf == want_vcf
    assert got_table == want_table and want_table.count("\n") > 20
    assert len(recs) == want_vcf.count("\n") and all(r["exon_pos"] == r["ref_pos"] for r in recs)  # (an rna reference: every base is exonic)


@pytest.mark.parametrize("seed", [3, 17])
def test_several_variants_per_gene_with_sequencing_errors(built, tmp_path, seed):
    """groups of more than one candidate (two unknown bases within a read's reach), candidates expanded to the other selected alleles,
    sequencing errors that must stay below the thresholds; also --varMaxGroup 1, which leaves the two-candidate groups unresolved"""
    util.need(util.ORACLE_CLI)
    tmp = str(tmp_path)
    ref, pfx = util.several_snps_sample(tmp, seed)
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx)
    assert want_vcf.count("\n") >= 2, want_vcf
    assert got_vcf == want_vcf
    assert got_table == want_table
    sub = os.path.join(tmp, "g1")
    os.makedirs(sub)
    want1, got1, wt1, gt1, _, _ = run_case(sub, ref, pfx, var_max_group=1, ana_flags=["--varMaxGroup", "1"])
    assert got1 == want1 and gt1 == wt1


def test_single_end_run_and_no_variant_calling(built, tmp_path):
    """-u input (no second read anywhere) and --varMaxGroup 0 (ComputeVariant returns at once: empty VCF, raw lists counted)"""
    util.need(util.ORACLE_CLI)
    tmp = str(tmp_path)
    ref, pfx = util.several_snps_sample(tmp, 29, genes=3, pairs=2500)
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx, single=True)
    assert want_vcf.count("\n") >= 1 and got_vcf == want_vcf and got_table == want_table
    sub = os.path.join(tmp, "g0")
    os.makedirs(sub)
    want0, got0, wt0, gt0, moved0, _ = run_case(sub, ref, pfx, single=True, var_max_group=0, ana_flags=["--varMaxGroup", "0"])
    assert want0 == got0 == "" and gt0 == wt0 and moved0 == 0


def test_genomic_reference_with_introns_and_separators(built, tmp_path):
    """a dna reference (exon coordinates in the records' comments, N separators): unknown bases inside an exon are called with their
    exonic coordinate, the ones inside an intron are not written"""
    util.need(util.ORACLE_CLI)
    tmp = str(tmp_path)
    ref, pfx = util.several_snps_sample(tmp, 41, genes=3, kind="ref-dna", scale=0.05, positions=tuple(range(120, 2400, 97)), pairs=6000)
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx, sanitize=True)
    assert got_vcf == want_vcf
    assert got_table == want_table
    assert want_vcf.count("\n") >= 1, "the sample calls no variant: it does not test what it is meant to"
    assert any(r["exon_pos"] != r["ref_pos"] for r in recs)


@pytest.mark.parametrize("seed", [7, 53])
def test_reads_with_indels_and_unknown_bases(built, tmp_path, seed):
    """reads with insertions, deletions and N on top of the unknown SNPs: edit strings with gap columns, the reference's base walk that
    stays where it is when it skips a column (an N in the read, an overlap that is not good for the base), overlaps of lower match counts
    that IsGoodAssignment filters"""
    util.need(util.ORACLE_CLI)
    tmp = str(tmp_path)
    ref, pfx = util.several_snps_sample(tmp, seed, genes=4, pairs=5000, sub=0.004, indel=0.004, nrate=0.004)
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx, sanitize=seed == 7)
    assert want_vcf.count("\n") >= 4, want_vcf
    assert got_vcf == want_vcf
    assert got_table == want_table
    if seed == 7:
        assert LAST["sanitized"] or sanitizer_harness(tmp) is None


def test_relaxed_intron_alignment_on_a_genomic_reference(built, tmp_path):
    """--relaxIntronAlign on both stages (the analyzer's assignment keeps fragments with intronic mismatches, SeqSet.hpp:2491-2519): more
    assignments per fragment reach the variant caller"""
    util.need(util.ORACLE_CLI)
    tmp = str(tmp_path)
    ref, pfx = util.several_snps_sample(tmp, 61, genes=3, kind="ref-dna", scale=0.05, positions=tuple(range(90, 2400, 61)), pairs=6000, sub=0.003)
    flags = ["--relaxIntronAlign"]
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx, geno_flags=flags, ana_flags=flags, orc_flags=flags)
    assert want_vcf.count("\n") >= 1, "the sample calls no variant"
    assert got_vcf == want_vcf
    assert got_table == want_table


def test_real_database_with_exon_structure(built, tmp_path):
    """the CYP2D6 genomic database of the reference's own example (exon lists in the record comments, allele names in its own digit
    structure): unknown bases every 53 positions in two alleles of three -- the exonic ones must come out with the coordinate
    SeqSet::GetExonicPosition gives them, the intronic ones must not be written"""
    util.need(util.ORACLE_CLI)
    tmp = str(tmp_path)
    ref, pfx = util.several_snps_sample(tmp, 83, kind=util.CYP_DNA, positions=tuple(range(140, 9000, 53)), pairs=5000, sub=0.001)
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx, geno_flags=util.CYP_FLAGS, ana_flags=util.CYP_FLAGS, orc_flags=util.CYP_FLAGS,
                                                                       job_kw=dict(allele_digit_units=1, allele_delimiter="."))
    assert want_vcf.count("\n") >= 3, want_vcf
    assert got_vcf == want_vcf
    assert got_table == want_table
    assert any(r["exon_pos"] != r["ref_pos"] for r in recs)


def test_fragments_with_one_unalignable_mate(built, tmp_path):
    """every fifth first mate and every seventh second mate replaced by random bases: fragments assigned through one read-end only
    (SeqSet.hpp:2330-2346, 2563-2590 -- kept when that end is a perfect match at the allele's edge), among them the ones whose single
    overlap comes from the SECOND read (o1FromR2: VariantCaller reads read 2 for overlap 1, VariantCaller.hpp:296-301, 376-380, 1241-1243)"""
    util.need(util.ORACLE_CLI)
    import random
    tmp = str(tmp_path)
    ref, pfx = util.several_snps_sample(tmp, 71, genes=4, pairs=9000, sub=0.0, indel=0.0, nrate=0.0, fragmean=230, fragsd=60)
    rng = random.Random(5)
    for mate, every in ((1, 5), (2, 7)):
        path = "%s_%d.fq" % (pfx, mate)
        lines = open(path).read().split("\n")
        for i in range(0, len(lines) - 3, 4):
            if (i // 4) % every == 0:
                lines[i + 1] = "".join(rng.choice("ACGT") for _ in lines[i + 1])
        open(path, "w").write("\n".join(lines))
    want_vcf, got_vcf, want_table, got_table, moved, recs = run_case(tmp, ref, pfx, sanitize=True)
    assert LAST["from_r2"] > 0 and LAST["unpaired"] > LAST["from_r2"], LAST
    assert got_vcf == want_vcf and want_vcf.count("\n") >= 1
    assert got_table == want_table


def test_bad_input_is_refused(built, tmp_path):
    """an assignment whose window leaves its allele, or whose edit string does not spell its windows, is an argument error, not a crash"""
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref("ref-rna", ref, genes=1, scale=0.05, seed=1)
    name, head, seq = util.read_fa(ref)[0]
    one = os.path.join(tmp, "one.fa")
    open(one, "w").write(head + "\n" + seq + "\n")
    job = t1k_amd.Job(one, device=-1)
    read = seq[10:60]
    o = (0, 0, 49, 10, 59, 1, 50, 0, 0, 50, 1.0)
    zero = (0,) * 10 + (0.0,)
    good = np.array([(0, 0, 0, 0, o, zero, 0, 0, 50, 0)], dtype=t1k_amd.FRAG_ASG_DTYPE)
    ops = np.zeros(50, dtype=np.int8)
    v = job.call_variants([1.0], 8, [0, 1], good, ops, [read])
    assert v.vcf() == "" and list(v.adjust(good, ops, read)) == [1]
    v.close()
    for field, value in (("seq_end", len(seq)), ("read_end", 50), ("seq_start", -1), ("strand", 0), ("seq_idx", 1)):
        bad = good.copy()
        bad["o1"][field] = value
        with pytest.raises(t1k_amd.T1kError):
            job.call_variants([1.0], 8, [0, 1], bad, ops, [read])
    short = good.copy()
    short["n_ops1"] = 49
    with pytest.raises(t1k_amd.T1kError):
        job.call_variants([1.0], 8, [0, 1], short, ops, [read])
    job.close()
