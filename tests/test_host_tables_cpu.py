"""CPU tests of the product's host side after the device loop (t1k_amd/csrc/host/refset.cpp + genotype.cpp, plain C++, built into
tests/harness/host_tables_harness.cpp with g++): reference loading and naming, the class build, likelihood pruning, allele selection
with the type-pair search, genotype quality and the two tables -- fed with the read groups and the EM result the oracle CLI dumps
(the E-step itself is a device stage and is not run here), compared with the reference's own files.  The GPU suite covers the
same code end to end; this is its CPU-side pin."""
import os
import subprocess

import pytest

import goldens
import test_oracle_golden as tog
import util

HOST = os.path.join(util.ROOT, "t1k_amd", "csrc", "host")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("harness") / "host_tables_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(util.ROOT, "tests", "harness", "host_tables_harness.cpp"),
                    os.path.join(HOST, "refset.cpp"), os.path.join(HOST, "genotype.cpp"), "-lz", "-lpthread"], check=True)
    return exe


def longest_read(*paths):
    m = 0
    for p in paths:
        for i, line in enumerate(open(p)):
            if i % 4 == 1:
                m = max(m, len(line.rstrip("\r\n")))
    return m


def run_host(harness, ref, oracle_prefix, read_length, flags, out):
    opt = {flags[i]: flags[i + 1] for i in range(len(flags) - 1) if flags[i].startswith("-")}
    args = [harness, ref, oracle_prefix, str(read_length), opt.get("--frac", "0.15"), opt.get("--cov", "1.0"), opt.get("--crossGeneRate", "0.04"), out,
            opt.get("--alleleDigitUnits", "-1")] + ([opt["--alleleDelimiter"]] if "--alleleDelimiter" in opt else [])
    r = subprocess.run(args, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("name", goldens.CASES)
def test_host_tables_match_reference_outputs(built, harness, tmp_path, name):
    c = goldens.Case(name, str(tmp_path))
    orc = str(tmp_path / "orc")
    subprocess.run([util.ORACLE_CLI] + c.args() + ["-o", orc], check=True, stderr=subprocess.PIPE)
    out = str(tmp_path / "host")
    run_host(harness, c.ref, orc, longest_read(*([c.r1, c.r2] if c.paired else [c.r1])), c.flags, out)
    assert open(out + "_genotype.tsv").read() == c.expected("genotype.tsv")
    assert open(out + "_allele.tsv").read() == c.expected("allele.tsv")


@pytest.mark.parametrize("case", range(len(tog.LIVE_TABLE_CASES)))
def test_host_tables_on_mixed_samples_with_options(built, harness, tmp_path, case):
    """samples mixed from several simulated individuals (genes with three and more allele types) and the selection options; expected
    = the oracle's tables, which test_oracle_golden.py pins to the reference binary on these very inputs"""
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
    r1, r2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
    orc = os.path.join(tmp, "orc")
    subprocess.run([util.ORACLE_CLI, "-f", ref, "-1", r1, "-2", r2] + flags + ["-o", orc], check=True, stderr=subprocess.PIPE)
    out = os.path.join(tmp, "host")
    run_host(harness, ref, orc, longest_read(r1, r2), flags, out)
    for what in ("_genotype.tsv", "_allele.tsv"):
        assert open(out + what).read() == open(orc + what).read(), what
