"""Stand-alone GPU check of the AssignRead stage against the oracle (also wrapped by tests/test_gpu_assign.py)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import util  # noqa: E402
import t1k_amd  # noqa: E402


def compare(fasta, reads, sim, relax, label, max_report=5):
    names, seqs, masks, weights = t1k_amd.load_reference_fasta(fasta)
    ctx = t1k_amd.Context(ref_seq_similarity=sim, relax_intron_align=1 if relax else 0)
    ctx.ref_upload(seqs, masks)
    ctx.reads_upload(reads)
    t0 = time.time()
    ctx.assign()
    dt = time.time() - t0
    counts, ovl = ctx.overlaps()
    st = ctx.stats()
    orc = util.Oracle(fasta, similarity=sim, relax=relax)
    bad = 0
    pos = 0
    t1 = time.time()
    for i, r in enumerate(reads):
        o, s = orc.assign_read(r)
        g = ovl[pos:pos + counts[i]]
        pos += counts[i]
        ok = len(o) == len(g)
        if ok and len(o):
            ok = (np.array_equal(o[:, 0], g["seq_idx"]) and np.array_equal(o[:, 1], g["read_start"]) and np.array_equal(o[:, 2], g["read_end"])
                  and np.array_equal(o[:, 3], g["seq_start"]) and np.array_equal(o[:, 4], g["seq_end"]) and np.array_equal(o[:, 5], g["strand"])
                  and np.array_equal(o[:, 6], g["match_cnt"]) and np.array_equal(o[:, 7], g["left_clip"]) and np.array_equal(o[:, 8], g["right_clip"])
                  and np.array_equal(o[:, 9], g["relaxed_match_cnt"]) and np.array_equal(s, g["similarity"]))
        if not ok:
            bad += 1
            if bad <= max_report:
                print("MISMATCH %s read %d: oracle %d overlaps, gpu %d" % (label, i, len(o), len(g)))
                for k in range(min(3, max(len(o), len(g)))):
                    print("   orc", o[k].tolist() if k < len(o) else None, s[k] if k < len(o) else None)
                    print("   gpu", g[k] if k < len(g) else None)
    tcpu = time.time() - t1
    # coverage
    cov = ctx.coverage()
    cbad = 0
    off = 0
    for a, sq in enumerate(seqs):
        oc = orc.coverage(a, len(sq))
        if not np.array_equal(oc, cov[off:off + len(sq)]):
            cbad += 1
            if cbad <= 2:
                d = np.nonzero(oc != cov[off:off + len(sq)])[0]
                print("COVERAGE MISMATCH %s allele %d at %s: orc %s gpu %s" % (label, a, d[:8], oc[d[:8]], cov[off + d[:8]]))
        off += len(sq)
    print("%s: %d read-ends, %d overlaps, %d mismatching read-ends, %d alleles with coverage mismatch; gpu %.3fs (cpu oracle %.1fs) stats %s"
          % (label, len(reads), len(ovl), bad, cbad, dt, tcpu, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}))
    ctx.close()
    return bad + cbad


def main():
    tmp = tempfile.mkdtemp(prefix="t1k_")
    total = 0
    # 1. CYP2D6 rna, default -s 0.8
    rna = util.gunzip_to(util.CYP_RNA, os.path.join(tmp, "cyp_rna.fa"))
    util.synth_reads(rna, os.path.join(tmp, "c1"), pairs=300, len=100, seed=11, sub=0.005)
    reads = [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "c1_1.fq"))] + [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "c1_2.fq"))]
    total += compare(rna, reads, 0.8, False, "cyp2d6_rna")
    # 2. CYP2D6 dna, kir-wgs flags
    dna = util.gunzip_to(util.CYP_DNA, os.path.join(tmp, "cyp_dna.fa"))
    util.synth_reads(dna, os.path.join(tmp, "c2"), pairs=200, len=150, seed=12, sub=0.005, fragmean=420)
    reads = [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "c2_1.fq"))] + [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "c2_2.fq"))]
    total += compare(dna, reads, 0.9, True, "cyp2d6_dna_relax")
    # 3. synthetic HLA-like rna, -s 0.97
    hla = os.path.join(tmp, "hla.fa")
    util.synth_ref("ref-rna", hla, genes=4, scale=0.05)
    util.synth_reads(hla, os.path.join(tmp, "h1"), pairs=200, len=150, seed=3)
    reads = [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "h1_1.fq"))] + [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "h1_2.fq"))]
    total += compare(hla, reads, 0.97, False, "synthetic_hla")
    # 4. 2 x 250 bp reads (the 320-position instantiations of the seeding / chaining kernels), more substitutions
    util.synth_reads(hla, os.path.join(tmp, "h2"), pairs=120, len=250, seed=5, sub=0.01, fragmean=520)
    reads = [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "h2_1.fq"))] + [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "h2_2.fq"))]
    total += compare(hla, reads, 0.9, False, "synthetic_hla_250bp")
    # 5. indel-rich reads: groups with several diagonals (gather / LDS / wave-cooperative chaining, band alignments)
    util.synth_reads(hla, os.path.join(tmp, "h3"), pairs=150, len=150, seed=9, sub=0.004, indel=0.006)
    reads = [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "h3_1.fq"))] + [s for _, _, s in t1k_amd.read_fastx(os.path.join(tmp, "h3_2.fq"))]
    total += compare(hla, reads, 0.9, False, "synthetic_hla_indels")
    print("TOTAL MISMATCHES", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
