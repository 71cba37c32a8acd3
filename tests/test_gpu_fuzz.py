"""GPU fuzz of the AssignRead stage against the oracle (pytest -m gpu): adversarial read-ends built from the reference itself -- the
shapes seeded synthetic sequencing never produces.  Every overlap list (coordinates, matchCnt, clips, similarity) and the per-base
coverage must equal the oracle's."""
import os
import random

import pytest

import gpu_assign_check
import util

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("T1K_FUZZ_SCALE", "1"))      # more reads per case
SEED0 = int(os.environ.get("T1K_FUZZ_SEED", "0"))        # shifts every seed: a fresh fuzz run
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def alleles(path):
    out, cur = [], []
    for l in open(path):
        if l[0] == ">":
            if cur:
                out.append("".join(cur))
            cur = []
        else:
            cur.append(l.strip())
    if cur:
        out.append("".join(cur))
    return out


def adversarial_reads(al, rng, n):
    rnd = lambda m: "".join(rng.choice("ACGT") for _ in range(m))
    reads = []
    while len(reads) < n:
        a = rng.choice(al)
        L = rng.choice([11, 12, 20, 31, 32, 33, 36, 50, 64, 75, 96, 100, 128, 150, 151, 160, 161, 192, 200, 250, 256, 288, 319, 320])
        kind = rng.randrange(12)
        if len(a) < L + 2:
            continue
        p = rng.randrange(0, len(a) - L)
        s = a[p:p + L]
        if kind == 0:    # hangs over the start / end of the allele
            cut = rng.randrange(1, max(2, L // 2))
            s = (rnd(cut) + a[:L - cut]) if rng.random() < 0.5 else (a[len(a) - (L - cut):] + rnd(cut))
        elif kind == 1:  # chimera of two alleles
            b = rng.choice(al)
            q = rng.randrange(0, max(1, len(b) - L))
            h = rng.randrange(5, max(6, L - 5))
            s = a[p:p + h] + b[q:q + L - h]
        elif kind == 2:  # internal deletion / insertion
            h = rng.randrange(3, max(4, L - 3))
            d = rng.choice([1, 2, 3, 5, 9, 14, 30])
            s = (s[:h] + s[h + d:]) if rng.random() < 0.5 else (s[:h] + rnd(d) + s[h:])
        elif kind == 3:  # many N
            t = list(s)
            for _ in range(rng.choice([1, 2, 5, 12, 30])):
                t[rng.randrange(len(t))] = "N"
            s = "".join(t)
        elif kind == 4:  # clustered substitutions
            t = list(s)
            c0 = rng.randrange(len(t))
            for i in range(c0, min(len(t), c0 + rng.choice([2, 4, 8, 16]))):
                if rng.random() < 0.6:
                    t[i] = rng.choice("ACGT")
            s = "".join(t)
        elif kind == 5:  # tandem duplication of a segment of the read
            h = rng.randrange(0, max(1, L - 20))
            seg = s[h:h + rng.choice([3, 7, 12, 25])]
            s = s[:h] + seg * rng.choice([2, 3]) + s[h:]
        elif kind == 6:  # low-complexity tail
            s = s[:L // 2] + rng.choice(["A", "T", "AC", "GT", "CAG"]) * (L // 2)
        elif kind == 7:  # substitution at every k-th base (few or no k-mer hits)
            t = list(s)
            step = rng.choice([6, 9, 11, 12, 15])
            for i in range(rng.randrange(step), len(t), step):
                t[i] = COMP[t[i]]
            s = "".join(t)
        elif kind == 8:  # pure noise / homopolymer
            s = rnd(L) if rng.random() < 0.5 else rng.choice("ACGT") * L
        elif kind == 9:  # N at the very ends, N next to the ends
            t = list(s)
            for i in rng.sample([0, 1, len(t) - 2, len(t) - 1], 2):
                t[i] = "N"
            s = "".join(t)
        elif kind == 10:  # inverted middle
            h = L // 3
            s = s[:h] + rc(s[h:2 * h]) + s[2 * h:]
        # kind 11: exact window
        s = s[:320]
        reads.append(s if rng.random() < 0.5 else rc(s))
    return reads


@pytest.mark.parametrize("ref_gz,sim,relax,seed", [("rna", 0.8, False, 1), ("dna", 0.9, True, 2), ("rna", 0.97, False, 3), ("dna", 0.8, False, 4)])
def test_adversarial_reads_vs_oracle(built, tmp_path, ref_gz, sim, relax, seed):
    ref = util.gunzip_to(util.CYP_RNA if ref_gz == "rna" else util.CYP_DNA, str(tmp_path / "ref.fa"))
    reads = adversarial_reads(alleles(ref), random.Random(seed + SEED0), 1500 * SCALE)
    assert gpu_assign_check.compare(ref, reads, sim, relax, "fuzz %s s=%s" % (ref_gz, sim)) == 0


@pytest.mark.parametrize("kind,sim,relax,seed", [("ref-rna", 0.8, False, 5), ("ref-dna", 0.8, True, 6)])
def test_adversarial_reads_many_alleles_vs_oracle(built, tmp_path, kind, sim, relax, seed):
    """the same on a synthetic reference with hundreds of near-identical alleles per gene (large hit groups, ties, long posting lists)"""
    ref = str(tmp_path / "ref.fa")
    util.synth_ref(kind, ref, seed=seed + SEED0, genes=6, scale=0.3)
    reads = adversarial_reads(alleles(ref), random.Random(seed + SEED0), 800 * SCALE)
    assert gpu_assign_check.compare(ref, reads, sim, relax, "fuzz %s s=%s" % (kind, sim)) == 0


def adversarial_pairs(al, rng, n):
    """fragments whose mates relate in the ways a sequencer rarely produces: same orientation, swapped, overlapping, contained, from
    different alleles, one mate noise; every mate then goes through one of the read-level distortions"""
    rnd = lambda m: "".join(rng.choice("ACGT") for _ in range(m))
    pairs = []
    while len(pairs) < n:
        a = rng.choice(al)
        L = rng.choice([50, 75, 100, 150])
        F = rng.choice([L, L + 10, 2 * L - 20, 2 * L + 50, 400])
        if len(a) < F + 2:
            continue
        p = rng.randrange(0, len(a) - F)
        frag = a[p:p + F]
        m1, m2 = frag[:L], rc(frag)[:L]
        kind = rng.randrange(9)
        if kind == 0:
            m2 = rc(m2)                       # both mates on the same strand
        elif kind == 1:
            m1, m2 = rc(m1), rc(m2)           # outward facing
        elif kind == 2:
            b = rng.choice(al)
            q = rng.randrange(0, max(1, len(b) - L))
            m2 = rc(b[q:q + L])               # mate from another allele (maybe another gene)
        elif kind == 3:
            m2 = rnd(L)                       # one mate is noise
        elif kind == 4:
            m2 = rc(m1)                       # mates are reverse complements of each other
        elif kind == 5:
            m2 = m1                           # identical mates
        elif kind == 6:
            m2 = rc(frag[L // 2:L // 2 + L // 3])  # short mate inside the other
        elif kind == 7:
            m1 = adversarial_reads([a], rng, 1)[0][:L]
        # kind 8: a proper pair
        if rng.random() < 0.3:
            t = list(m2)
            for _ in range(rng.choice([1, 3, 8])):
                t[rng.randrange(len(t))] = rng.choice("ACGTN")
            m2 = "".join(t)
        if rng.random() < 0.5:
            m1, m2 = m2, m1
        pairs.append((m1, m2))
    return pairs


@pytest.mark.parametrize("kind,flags,seed", [("ref-rna", ["-s", "0.8"], 7), ("ref-dna", ["-s", "0.9", "--relaxIntronAlign"], 8), ("ref-rna", ["-s", "0.97", "-n", "40"], 9)])
def test_adversarial_pairs_executable_vs_reference_binary(built, tmp_path, kind, flags, seed):
    """mate pairing, row weights, coalescing, EM and selection on adversarial fragments: every output file of the executable (including
    the per-fragment assignment table) against the reference binary's"""
    util.need(util.REF_BIN)  # decided when the test runs, after the `built` fixture had its chance to build oracle/_ref
    import subprocess
    ref = str(tmp_path / "ref.fa")
    util.synth_ref(kind, ref, seed=seed + SEED0, genes=5, scale=0.15)
    pairs = adversarial_pairs(alleles(ref), random.Random(seed + SEED0), 6000 * SCALE)
    for i, suffix in enumerate(("_1.fq", "_2.fq")):
        with open(str(tmp_path / "p") + suffix, "w") as f:
            for j, pr in enumerate(pairs):
                f.write("@f%d/%d\n%s\n+\n%s\n" % (j, i + 1, pr[i], "I" * len(pr[i])))
    args = ["-f", ref, "-1", str(tmp_path / "p_1.fq"), "-2", str(tmp_path / "p_2.fq")] + flags + ["--outputReadAssignment"]
    exe = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")
    subprocess.run([exe] + args + ["-o", str(tmp_path / "ours")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([util.REF_BIN] + args + ["-t", "16", "-o", str(tmp_path / "ref")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    n = 0
    for fn in sorted(os.listdir(str(tmp_path))):
        if fn.startswith("ref_"):
            a, b = open(str(tmp_path / fn)).read(), open(str(tmp_path / ("ours_" + fn[4:]))).read()
            assert a == b, fn
            n += 1
    assert n >= 5 and open(str(tmp_path / "ref_assign.tsv")).read().count("\n") > 1000
    # the same read-ends as a single-end run (-u) with per-read barcodes: the dangling-mate rules do not apply, the barcode path does
    with open(str(tmp_path / "bc.fa"), "w") as f:
        for j in range(len(pairs)):
            f.write(">f%d\n%s\n" % (j, "missing_barcode" if j % 97 == 0 else "ACGTTGCA"[j % 3:] + "ACGT"[j % 4] * 4))
    args = ["-f", ref, "-u", str(tmp_path / "p_2.fq"), "--barcode", str(tmp_path / "bc.fa")] + flags + ["--outputReadAssignment"]
    subprocess.run([exe] + args + ["-o", str(tmp_path / "sours")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([util.REF_BIN] + args + ["-t", "16", "-o", str(tmp_path / "sref")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m = 0
    for fn in sorted(os.listdir(str(tmp_path))):
        if fn.startswith("sref_"):
            assert open(str(tmp_path / fn)).read() == open(str(tmp_path / ("sours_" + fn[5:]))).read(), fn
            m += 1
    assert m >= 4


@pytest.mark.parametrize("option", ["--frac", "--cov", "--crossGeneRate", "--squaremMinAlpha", "--alleleWhitelist", "-a"])
def test_selection_options_executable_vs_reference_binary(built, tmp_path, option):
    """the options of the quantification / selection half that no other test sets (Genotyper.cpp:13-57): filter fraction and coverage,
    cross-gene rate, SQUAREM step bound, an allele whitelist (whole major-allele series, Genotyper.hpp:684-705) and an abundance file
    that replaces the EM (Genotyper.hpp:1016-1051) -- every output file against the reference binary's"""
    util.need(util.REF_BIN)
    import subprocess
    ref = str(tmp_path / "ref.fa")
    util.synth_ref("ref-rna", ref, seed=31 + SEED0, genes=6, scale=0.15)
    names = [l[1:].split()[0] for l in open(ref) if l.startswith(">")]
    pairs = adversarial_pairs(alleles(ref), random.Random(31 + SEED0), 5000 * SCALE)
    for i, suffix in enumerate(("_1.fq", "_2.fq")):
        with open(str(tmp_path / "p") + suffix, "w") as f:
            for j, pr in enumerate(pairs):
                f.write("@f%d/%d\n%s\n+\n%s\n" % (j, i + 1, pr[i], "I" * len(pr[i])))
    flags = {"--frac": ["--frac", "0.4"], "--cov": ["--cov", "3.5"], "--crossGeneRate": ["--crossGeneRate", "0.3"],
             "--squaremMinAlpha": ["--squaremMinAlpha", "-1.5"]}.get(option)
    if option == "--alleleWhitelist":
        wl = str(tmp_path / "whitelist.txt")
        open(wl, "w").write("\n".join(names[::3]) + "\n")
        flags = ["--alleleWhitelist", wl]
    if option == "-a":
        ab = str(tmp_path / "abundance.tsv")
        rng = random.Random(5)
        with open(ab, "w") as f:
            f.write("allele len efflen count abundance\n")
            for n in names[::2]:
                f.write("%s 1000 900 %.3f %.4f\n" % (n, rng.random() * 200, rng.random()))
        flags = ["-a", ab]
    args = ["-f", ref, "-1", str(tmp_path / "p_1.fq"), "-2", str(tmp_path / "p_2.fq"), "-s", "0.9"] + flags
    exe = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")
    a = subprocess.run([exe] + args + ["-o", str(tmp_path / "ours")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    b = subprocess.run([util.REF_BIN] + args + ["-t", "16", "-o", str(tmp_path / "ref")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-800:], b.stderr[-800:])
    n = 0
    for fn in sorted(os.listdir(str(tmp_path))):
        if fn.startswith("ref_"):
            assert open(str(tmp_path / fn)).read() == open(str(tmp_path / ("ours_" + fn[4:]))).read(), fn
            n += 1
    assert n >= 4 and os.path.getsize(str(tmp_path / "ref_genotype.tsv")) > 100


@pytest.mark.parametrize("kind,flags,seed", [("ref-rna", [], 11), ("ref-dna", ["-s", "0.95"], 12), ("ref-rna", ["-s", "0.99", "-t", "4"], 13)])
def test_adversarial_reads_extractor_vs_reference_binary(built, tmp_path, kind, flags, seed):
    """the candidate extractor on the same adversarial read-ends and mate pairs, files against the reference's fastq-extractor"""
    util.need(util.REF_EXTRACT)  # decided when the test runs, after the `built` fixture had its chance to build oracle/_ref
    import subprocess
    ref = str(tmp_path / "ref.fa")
    util.synth_ref(kind, ref, seed=seed + SEED0, genes=5, scale=0.15)
    al = alleles(ref)
    rng = random.Random(seed + SEED0)
    pairs = adversarial_pairs(al, rng, 3000 * SCALE) + list(zip(adversarial_reads(al, rng, 3000 * SCALE), adversarial_reads(al, rng, 3000 * SCALE)))
    for i, suffix in enumerate(("_1.fq", "_2.fq")):
        with open(str(tmp_path / "p") + suffix, "w") as f:
            for j, pr in enumerate(pairs):
                f.write("@f%d/%d\n%s\n+\n%s\n" % (j, i + 1, pr[i], "I" * len(pr[i])))
    exe = os.path.join(util.ROOT, "t1k_amd", "bin", "fastq-extractor")
    for mode, reads in (("pe", ["-1", str(tmp_path / "p_1.fq"), "-2", str(tmp_path / "p_2.fq")]), ("se", ["-u", str(tmp_path / "p_2.fq")])):
        args = ["-f", ref] + reads + flags
        subprocess.run([exe] + args + ["-o", str(tmp_path / ("ours_" + mode))], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([util.REF_EXTRACT] + args + ["-o", str(tmp_path / ("ref_" + mode))], check=True, stderr=subprocess.DEVNULL)
        for suffix in (["_1.fq", "_2.fq"] if mode == "pe" else [".fq"]):
            a, b = open(str(tmp_path / ("ref_" + mode)) + suffix).read(), open(str(tmp_path / ("ours_" + mode)) + suffix).read()
            assert a == b, (mode, suffix)
            assert 0 < a.count("\n") // 4 < len(pairs)
