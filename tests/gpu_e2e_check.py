"""End-to-end GPU check: the drop-in `genotyper` executable vs the reference binary (oracle/_ref/genotyper) on the same
inputs: *_genotype.tsv, *_allele.tsv, *_assign.tsv, *_aligned_*.fa must be byte-identical; the EM iteration count printed
in the log must agree."""
import os
import re
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import util  # noqa: E402

GENO = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")


def run(binary, args, log):
    t0 = time.time()
    p = subprocess.run([binary] + args, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    dt = time.time() - t0
    with open(log, "w") as f:
        f.write(p.stderr)
    if p.returncode != 0:
        print(p.stderr[-2000:])
        raise RuntimeError("%s exited with %d" % (binary, p.returncode))
    m = re.search(r"in (\d+) EM iterations", p.stderr)
    m2 = re.search(r"(\d+) read fragments can be assigned \(average ([-\d.naninf]+) alleles/read\)", p.stderr)
    return dt, (int(m.group(1)) if m else None), (m2.groups() if m2 else None)


def case(tmp, label, ref, pfx, flags, paired=True, barcode=None):
    reads = ["-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] if paired else ["-u", pfx + "_1.fq"]
    if barcode:
        reads += ["--barcode", barcode]
    common = ["-f", ref] + reads + flags + ["--outputReadAssignment"]
    o_ref, o_gpu = os.path.join(tmp, label + "_ref"), os.path.join(tmp, label + "_gpu")
    t_ref, it_ref, as_ref = run(util.REF_BIN, common + ["-t", "1", "-o", o_ref], o_ref + ".log")
    t_gpu, it_gpu, as_gpu = run(GENO, common + ["-t", "1", "-o", o_gpu], o_gpu + ".log")
    bad = 0
    sufs = ["_genotype.tsv", "_allele.tsv", "_assign.tsv"] + (["_aligned_1.fa", "_aligned_2.fa"] if paired else ["_aligned.fa"]) + (["_aligned_bc.fa"] if barcode else [])
    for s in sufs:
        a, b = open(o_ref + s).read(), open(o_gpu + s).read()
        if a != b:
            bad += 1
            la, lb = a.splitlines(), b.splitlines()
            print("DIFF %s%s: %d vs %d lines" % (label, s, len(la), len(lb)))
            shown = 0
            for i in range(min(len(la), len(lb))):
                if la[i] != lb[i]:
                    print("   ref:", la[i][:200])
                    print("   gpu:", lb[i][:200])
                    shown += 1
                    if shown >= 4:
                        break
    if it_ref != it_gpu or as_ref != as_gpu:
        bad += 1
        print("LOG DIFF %s: EM iterations %s vs %s, assigned %s vs %s" % (label, it_ref, it_gpu, as_ref, as_gpu))
    print("%s: %s; reference %.2fs, gpu genotyper %.2fs, EM iterations %s" % (label, "IDENTICAL" if not bad else "%d DIFFERENCES" % bad, t_ref, t_gpu, it_gpu))
    print(open(o_gpu + "_genotype.tsv").read().rstrip()[:600])
    return bad


def main():
    tmp = tempfile.mkdtemp(prefix="t1k_e2e_")
    total = 0
    rna = util.gunzip_to(util.CYP_RNA, os.path.join(tmp, "cyp_rna.fa"))
    dna = util.gunzip_to(util.CYP_DNA, os.path.join(tmp, "cyp_dna.fa"))
    util.synth_reads(rna, os.path.join(tmp, "c1"), pairs=1500, len=100, seed=11, sub=0.005)
    total += case(tmp, "cyp2d6_rna", rna, os.path.join(tmp, "c1"), util.CYP_FLAGS)
    total += case(tmp, "cyp2d6_rna_single", rna, os.path.join(tmp, "c1"), util.CYP_FLAGS, paired=False)
    util.synth_reads(dna, os.path.join(tmp, "c2"), pairs=1000, len=150, seed=12, sub=0.005, fragmean=420)
    total += case(tmp, "cyp2d6_dna_relax", dna, os.path.join(tmp, "c2"), util.CYP_FLAGS + ["-s", "0.9", "--relaxIntronAlign"])
    hla = os.path.join(tmp, "hla.fa")
    util.synth_ref("ref-rna", hla, genes=4, scale=0.05)
    util.synth_reads(hla, os.path.join(tmp, "h1"), pairs=2000, len=150, seed=3, barcodes=50)
    total += case(tmp, "synthetic_hla", hla, os.path.join(tmp, "h1"), ["-s", "0.97"])
    total += case(tmp, "synthetic_hla_barcode", hla, os.path.join(tmp, "h1"), ["-s", "0.97"], barcode=os.path.join(tmp, "h1_bc.fa"))
    kir = os.path.join(tmp, "kir.fa")
    util.synth_ref("ref-dna", kir, genes=5, scale=0.3)
    util.synth_reads(kir, os.path.join(tmp, "k1"), pairs=1500, len=150, seed=5)
    total += case(tmp, "synthetic_kir_dna", kir, os.path.join(tmp, "k1"), ["-s", "0.9", "--relaxIntronAlign"])
    print("TOTAL DIFFERENCES", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
