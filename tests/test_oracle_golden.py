"""CPU tests (no GPU): the oracle (CPU restatement under oracle/) against golden vectors captured from the reference itself."""
import gzip
import os
import re
import subprocess

import pytest

import goldens
import util


def test_global_alignment_golden_vectors(built):
    """AlignAlgo::GlobalAlignment restatement vs 1872 input/output vectors produced by the reference's own routine."""
    orc = util.Oracle(None)
    n = 0
    with gzip.open(os.path.join(util.GOLDEN, "ga_vectors.tsv.gz"), "rt") as f:
        for line in f:
            t, p, score, ops = line.rstrip("\n").split("\t")
            s, o = orc.global_alignment(t, p)
            assert s == int(score), (t, p)
            assert "".join(str(int(x)) for x in o) == (ops if ops != "-" else ""), (t, p)
            n += 1
    assert n > 1800


@pytest.mark.parametrize("name", goldens.CASES)
def test_oracle_pipeline_matches_reference_outputs(built, tmp_path, name):
    """read-end assignment + pairing + rows (_assign.tsv), fragmentAssigned ids, EM iteration count, the last EM
    iteration's per-class read counts / abundances and the final _genotype.tsv / _allele.tsv must equal what the reference
    printed for the same inputs."""
    c = goldens.Case(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "orc")
    r = subprocess.run([util.ORACLE_CLI] + c.args() + ["-o", out], stderr=subprocess.PIPE, text=True)  # (--barcode: fragments without one are not loaded)
    assert r.returncode == 0, r.stderr
    # likelihood pruning, selection, genotype quality and the writers -- the two tables, byte for byte
    assert open(out + "_genotype.tsv").read() == c.expected("genotype.tsv")
    assert open(out + "_allele.tsv").read() == c.expected("allele.tsv")
    if True:
        assert open(out + "_assign.tsv").read() == c.expected("assign.tsv.gz")
        assert open(out + "_aligned_ids.txt").read().split() == c.expected("aligned_ids.txt.gz").split()
        it = int(open(out + "_em.tsv").readline().split()[1])
        assert it == c.meta["em_iterations"]
        exp = [l for l in c.expected("em_last_iteration.txt.gz").splitlines() if l.strip()]
        got = [l.rstrip("\n").split("\t") for l in open(out + "_em.tsv") if not l.startswith("#")]
        assert len(exp) == len(got)
        for e, g in zip(exp, got):
            m = re.match(r"^(\d+) (\S+) (\d+): (\S+) (\d+)\. (\S+)$", e)
            assert m, e
            assert int(m.group(1)) == int(g[0]) and m.group(2) == g[1] and int(m.group(3)) == len(g[1].split(","))
            assert int(m.group(5)) == int(g[2])
            assert m.group(4) == "%lf" % float(g[3]) if False else m.group(4) == ("%.6f" % float(g[3]))
            assert m.group(6) == ("%.6f" % float(g[4]))


def test_oracle_vs_live_reference_binary(built, tmp_path):
    """fresh seeded input, oracle CLI vs the reference binary built from /root/reference (oracle/_ref/genotyper)."""
    util.need(util.REF_BIN)  # decided when the test runs, after the `built` fixture had its chance to build oracle/_ref
    tmp = str(tmp_path)
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref("ref-rna", ref, genes=3, scale=0.02, seed=77)
    util.synth_reads(ref, os.path.join(tmp, "r"), pairs=150, len=150, seed=78)
    args = ["-f", ref, "-1", os.path.join(tmp, "r_1.fq"), "-2", os.path.join(tmp, "r_2.fq"), "-s", "0.95"]
    subprocess.run([util.REF_BIN] + args + ["-o", os.path.join(tmp, "a"), "--outputReadAssignment", "-t", "1"], check=True, stderr=subprocess.PIPE)
    subprocess.run([util.ORACLE_CLI] + args + ["-o", os.path.join(tmp, "b")], check=True, stderr=subprocess.PIPE)
    assert open(os.path.join(tmp, "a_assign.tsv")).read() == open(os.path.join(tmp, "b_assign.tsv")).read()


LIVE_TABLE_CASES = [  # (kind, genes, scale, read sets mixed into one sample, read length, flags): mixtures give genes more than two allele types
    ("ref-rna", 5, 0.02, 3, 150, ["-s", "0.9", "--crossGeneRate", "1.0"]),
    ("ref-rna", 4, 0.04, 3, 100, ["-s", "0.95", "--frac", "0.3", "--crossGeneRate", "0"]),
    ("ref-rna", 3, 0.02, 2, 150, ["-s", "0.9", "--frac", "0.05"]),
    ("ref-dna", 4, 0.02, 2, 150, ["-s", "0.95", "--relaxIntronAlign", "--cov", "3"]),
    ("ref-dna", 3, 0.02, 1, 75, ["-s", "0.8", "--squaremMinAlpha", "-2"]),
    ("ref-rna", 6, 0.01, 3, 150, ["-s", "0.97", "--cov", "20", "--frac", "0.5"]),
]


@pytest.mark.parametrize("case", range(len(LIVE_TABLE_CASES)))
def test_oracle_tables_vs_live_reference_binary(built, tmp_path, case):
    """pruning + selection + quality + writers of the oracle against the reference binary on fresh inputs, options included;
    samples mixed from several simulated individuals so that genes carry three and more allele types (the type-pair search,
    Genotyper.hpp:1697-1996)."""
    util.need(util.REF_BIN)
    kind, genes, scale, parts, length, flags = LIVE_TABLE_CASES[case]
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
    subprocess.run([util.REF_BIN] + args + ["-o", os.path.join(tmp, "a"), "-t", "1"], check=True, stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    subprocess.run([util.ORACLE_CLI] + args + ["-o", os.path.join(tmp, "b")], check=True, stderr=subprocess.PIPE)
    for what in ("_genotype.tsv", "_allele.tsv"):
        assert open(os.path.join(tmp, "a" + what)).read() == open(os.path.join(tmp, "b" + what)).read(), what
