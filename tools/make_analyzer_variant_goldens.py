#!/usr/bin/env python3
"""Golden fixtures of the analyzer's variant calling, produced by the REFERENCE binaries built by oracle/Makefile from
/root/reference: the two seeded samples of tests/util.novel_snp_sample (one unknown exonic SNP in every / every second allele of a gene)
run through oracle/_ref/genotyper, then oracle/_ref/analyzer in its default mode (--varMaxGroup 8), as run-t1k:438-449 chains them;
tests/golden/analyzer_variants/{homo,het}_allele.vcf and {homo,het}_barcode_expr.tsv are what the reference wrote.  The GPU test
test_analyzer_on_a_sample_with_a_novel_snp compares this build's files with them, also where oracle/_ref is absent.
  python tools/make_analyzer_variant_goldens.py"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import util  # noqa: E402

out = os.path.join(util.GOLDEN, "analyzer_variants")
os.makedirs(out, exist_ok=True)
for het, tag in ((False, "homo"), (True, "het")):
    tmp = tempfile.mkdtemp(prefix="t1k_anv_")
    ref, pfx = util.novel_snp_sample(tmp, het)
    g, a = os.path.join(tmp, "g"), os.path.join(tmp, "a")
    subprocess.run([util.REF_BIN, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "--barcode", pfx + "_bc.fa", "-o", g, "-t", "4"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([util.REF_ANALYZER, "-f", ref, "-a", g + "_allele.tsv", "-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa", "--barcode", g + "_aligned_bc.fa", "-o", a, "-t", "4"],
                   check=True, stderr=subprocess.DEVNULL)
    shutil.copy(a + "_allele.vcf", os.path.join(out, tag + "_allele.vcf"))
    shutil.copy(a + "_barcode_expr.tsv", os.path.join(out, tag + "_barcode_expr.tsv"))
    print(tag, open(a + "_allele.vcf").read().strip().replace("\n", " | "), os.path.getsize(a + "_barcode_expr.tsv"), "bytes of table")
    shutil.rmtree(tmp)
