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
        L = rng.choice([11, 12, 20, 31, 36, 50, 75, 100, 150, 151, 200, 250, 320])
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
