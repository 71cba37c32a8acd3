#!/usr/bin/env python3
"""Golden fixture of the analyzer row, produced by the REFERENCE binaries built by oracle/Makefile from /root/reference:
the committed hla_synth_2x150 case runs through oracle/_ref/genotyper, then oracle/_ref/analyzer on its outputs (as run-t1k:438-449 does);
tests/golden/hla_synth_2x150/analyzer_barcode_expr.tsv and analyzer_allele.vcf are what the reference wrote.
  python tools/make_analyzer_goldens.py"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import goldens  # noqa: E402
import util  # noqa: E402

tmp = tempfile.mkdtemp(prefix="t1k_an_")
c = goldens.Case("hla_synth_2x150", tmp)
g, a = os.path.join(tmp, "g"), os.path.join(tmp, "a")
subprocess.run([util.REF_BIN] + c.args() + ["-o", g, "-t", "1"], check=True, stderr=subprocess.DEVNULL)
subprocess.run([util.REF_ANALYZER, "-f", c.ref, "-a", g + "_allele.tsv", "-1", g + "_aligned_1.fa", "-2", g + "_aligned_2.fa", "--barcode", g + "_aligned_bc.fa",
                "-o", a, "-t", "1"] + c.flags, check=True, stderr=subprocess.DEVNULL)
shutil.copy(a + "_barcode_expr.tsv", os.path.join(c.dir, "analyzer_barcode_expr.tsv"))
shutil.copy(a + "_allele.vcf", os.path.join(c.dir, "analyzer_allele.vcf"))
print("wrote", os.path.join(c.dir, "analyzer_barcode_expr.tsv"), os.path.getsize(a + "_allele.vcf"), "bytes of VCF")
