"""Test infrastructure: a seeded writer of small coordinate-sorted BAM files (BGZF container, BAM records, aux tags) and of the
"coordinate FASTA" bam-extractor reads (">gene chrom start end strand" + sequence), so that this build's bam-extractor can be compared
with the reference's (oracle/_ref/bam-extractor, the reference's BamExtractor.cpp + its vendored samtools built by oracle/Makefile)
on inputs that exercise every branch of BamExtractor.cpp:613-938: reads over gene intervals, reads beside them, spliced / clipped
alignments, secondary and supplementary records, reads on alternative contigs, half-aligned pairs, unaligned pairs, low-complexity
reads, barcode / UMI tags, single-end data."""
import struct
import zlib

import numpy as np

SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def bgzf(data, block=0xff00):
    out = bytearray()
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        if i is not None and not chunk:
            continue
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = len(comp) + 25
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    return bytes(out)


def reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


CIGAR_OPS = "MIDNSHP=X"


def record(name, flag, tid, pos, cigar, seq, qual, mtid=-1, mpos=-1, tlen=0, mapq=30, tags=()):
    """seq / qual as they are STORED (i.e. reverse-complemented already when flag & 0x10); cigar = [(op char, length)]"""
    nm = name.encode() + b"\0"
    cg = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for op, n in cigar)
    ref_len = sum(n for op, n in cigar if op in "MDN=X")
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= SEQ_CODE[c] << (4 if i % 2 == 0 else 0)
    q = bytes(qual) if not isinstance(qual, str) else bytes(ord(c) - 33 for c in qual)
    aux = b""
    for tag, ty, val in tags:
        if ty == "Z":
            aux += tag.encode() + b"Z" + val.encode() + b"\0"
        elif ty == "i":
            aux += tag.encode() + b"i" + struct.pack("<i", val)
        elif ty == "A":
            aux += tag.encode() + b"A" + val.encode()
        elif ty == "B":  # array of uint16
            aux += tag.encode() + b"BS" + struct.pack("<I", len(val)) + b"".join(struct.pack("<H", v) for v in val)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(nm), mapq, reg2bin(max(pos, 0), max(pos, 0) + max(ref_len, 1)), len(cg) // 4, flag, len(seq), mtid, mpos, tlen)
    body += nm + cg + bytes(packed) + q + aux
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, records, block=0xff00):
    """refs = [(name, length)]; records = list of bytes from record(), already in file order"""
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, l in refs:
        head += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    with open(path, "wb") as f:
        f.write(bgzf(head + b"".join(records), block))


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def mutate(rng, s, rate):
    return "".join(("ACGT"[rng.integers(0, 4)] if rng.random() < rate else c) for c in s)


class Scenario:
    """A little genome: chr1 and chr6 (genes live on chr6), an alternative contig "chr6_GL000250v2_alt" and a decoy "chrUn.1"."""

    def __init__(self, seed, genes=4, gene_len=900, read_len=100, paired=True, barcodes=False):
        self.rng = np.random.default_rng(seed)
        rng = self.rng
        self.read_len, self.paired, self.barcodes = read_len, paired, barcodes
        self.refs = [("chr1", 300000), ("chr6", 400000), ("chr6_GL000250v2_alt", 50000), ("chrUn.1", 20000)]
        self.genome = {1: rand_seq(rng, 400000)}
        self.genes = []
        pos = 30000
        g = self.genome[1]
        for i in range(genes):
            seq = rand_seq(rng, gene_len)
            g = g[:pos] + seq + g[pos + gene_len:]
            self.genes.append(("GENE%d*01" % (i + 1), "chr6" if i % 2 == 0 else "6", pos, pos + gene_len - 1, "+", seq))
            pos += gene_len + int(rng.integers(3000, 30000))
        self.genome[1] = g
        self.genome[0] = rand_seq(rng, 300000)
        self.records = []   # (tid or big, pos, order, bytes)
        self.n = 0

    def write_fasta(self, path):
        with open(path, "w") as f:
            for name, chrom, s, e, strand, seq in self.genes:
                f.write(">%s %s %d %d %s\n%s\n" % (name, chrom, s, e, strand, seq))

    def _tags(self):
        if not self.barcodes:
            return ()
        t = []
        r = self.rng.random()
        if r < 0.8:
            t.append(("CB", "Z", rand_seq(self.rng, 16) + "-1"))
        if r > 0.3:
            t.append(("UB", "Z", rand_seq(self.rng, 10)))
        t.append(("NH", "i", 1))
        if r < 0.2:
            t.append(("XQ", "B", [1, 2, 3]))
        return tuple(t)

    def _qual(self, n):
        return bytes(int(x) for x in self.rng.integers(2, 41, n))

    def _add(self, tid, pos, rec):
        self.records.append((tid if tid >= 0 else 1 << 30, pos, len(self.records), rec))

    def aligned_pair(self, tid, start, frag=None, alt_seq=None, name=None, spliced=False, clip=False, low=False, flags_extra=0, suffix=False):
        """a properly oriented pair on contig tid; the read sequences come from the genome (or from alt_seq for alt contigs)"""
        L = self.read_len
        frag = frag or int(self.rng.integers(2 * L + 20, 3 * L + 100))
        self.n += 1
        name = name or "frag%06d" % self.n
        src = alt_seq if alt_seq is not None else self.genome[tid]
        a0, b0 = start, start + frag - L
        s1 = src[a0:a0 + L] if alt_seq is None else alt_seq[:L]
        s2 = src[b0:b0 + L] if alt_seq is None else alt_seq[-L:]
        if low:
            s1 = "A" * (L - 10) + s1[:10]
        c1, c2 = [("M", L)], [("M", L)]
        if spliced:  # the first mate jumps 1 000 bases after 40: its second segment lies elsewhere
            c1 = [("M", 40), ("N", 1000), ("M", L - 40)]
        if clip:
            c2 = [("S", 7), ("M", L - 12), ("S", 5)]
        tags = self._tags()
        n1, n2 = (name + "/1", name + "/2") if suffix else (name, name)
        self._add(tid, a0, record(n1, 0x1 | 0x2 | 0x20 | 0x40 | flags_extra, tid, a0, c1, s1, self._qual(L), tid, b0, frag, tags=tags))
        self._add(tid, b0, record(n2, 0x1 | 0x2 | 0x10 | 0x80 | flags_extra, tid, b0, c2, s2, self._qual(L), tid, a0, -frag, tags=tags))  # stored as aligned (forward strand of the reference)
        return name

    def secondary(self, tid, pos, name, first=True):
        L = self.read_len
        self._add(tid, pos, record(name, 0x1 | 0x100 | (0x40 if first else 0x80), tid, pos, [("M", L)], self.genome[tid][pos:pos + L] if tid in self.genome else rand_seq(self.rng, L), self._qual(L), tid, pos + 200, 0))

    def half_aligned_pair(self, tid, pos, from_gene=None):
        """one mate aligned, the other unaligned and placed at its mate's position (flag 0x4, tid set)"""
        L = self.read_len
        self.n += 1
        name = "half%06d" % self.n
        s1 = self.genome[tid][pos:pos + L]
        s2 = from_gene if from_gene is not None else rand_seq(self.rng, L)
        self._add(tid, pos, record(name, 0x1 | 0x8 | 0x40, tid, pos, [("M", L)], s1, self._qual(L), tid, pos, 0))
        self._add(tid, pos, record(name, 0x1 | 0x4 | 0x80, tid, pos, [], s2, self._qual(L), tid, pos, 0, mapq=0))
        return name

    def unaligned_pair(self, s1, s2, first_first=True, suffix=False):
        self.n += 1
        name = "unal%06d" % self.n
        tags = self._tags()
        n1, n2 = (name + "/1", name + "/2") if suffix else (name, name)
        a = record(n1, 0x1 | 0x4 | 0x8 | 0x40, -1, -1, [], s1, self._qual(len(s1)), -1, -1, 0, mapq=0, tags=tags)
        b = record(n2, 0x1 | 0x4 | 0x8 | 0x80, -1, -1, [], s2, self._qual(len(s2)), -1, -1, 0, mapq=0, tags=tags)
        for r in ((a, b) if first_first else (b, a)):
            self._add(-1, 0, r)
        return name

    def single(self, tid, pos, seq=None, reverse=False, cigar=None, name=None, flag_extra=0):
        L = self.read_len
        self.n += 1
        name = name or "read%06d" % self.n
        if tid >= 0:
            s = seq if seq is not None else self.genome[tid][pos:pos + L]
            self._add(tid, pos, record(name, (0x10 if reverse else 0) | flag_extra, tid, pos, cigar or [("M", len(s))], s, self._qual(len(s)), tags=self._tags()))
        else:
            self._add(-1, 0, record(name, 0x4, -1, -1, [], seq, self._qual(len(seq)), mapq=0, tags=self._tags()))
        return name

    def gene_read(self, gi, off=None, sub=0.01):
        """a read-length piece of gene gi with a few substitutions, and its reverse complement"""
        name, chrom, s, e, strand, seq = self.genes[gi]
        off = int(self.rng.integers(0, len(seq) - self.read_len)) if off is None else off
        return mutate(self.rng, seq[off:off + self.read_len], sub)

    def write(self, path, block=0xff00):
        self.records.sort(key=lambda r: (r[0], r[1], r[2]))
        write_bam(path, self.refs, [r[3] for r in self.records], block)


def paired_scenario(seed, with_unaligned=True, with_alt=True, barcodes=False, suffix=False, n=60, read_len=100, gene_len=900):
    sc = Scenario(seed, paired=True, barcodes=barcodes, read_len=read_len, gene_len=gene_len)
    rng = sc.rng
    L = sc.read_len
    for _ in range(n):  # pairs over and around the genes (overlap decided by the interval walk), a few far away
        name, chrom, s, e, strand, seq = sc.genes[int(rng.integers(0, len(sc.genes)))]
        start = int(rng.integers(s - 400, e + 100))
        sc.aligned_pair(1, start, spliced=rng.random() < 0.1, clip=rng.random() < 0.15, low=rng.random() < 0.05, suffix=suffix)
    for _ in range(n // 2):
        sc.aligned_pair(1, int(rng.integers(1000, 25000)), suffix=suffix)
        sc.aligned_pair(0, int(rng.integers(1000, 250000)), suffix=suffix)
    nm = sc.aligned_pair(1, sc.genes[0][2] + 50, suffix=False)
    sc.secondary(1, sc.genes[1][2] + 10, nm)
    for _ in range(n // 4):
        sc.half_aligned_pair(1, sc.genes[2][2] + int(rng.integers(0, 500)), from_gene=sc.gene_read(2) if rng.random() < 0.5 else None)
    if with_alt:
        for i in range(n // 3):  # alternative contig: reads that look like a gene (kept through HasHitInSet) and reads that do not
            g = int(rng.integers(0, len(sc.genes)))
            gene = sc.genes[g][5]
            off = int(rng.integers(0, len(gene) - 2 * L - 60))
            frag = mutate(rng, gene[off:off + 2 * L + 60], 0.01) if rng.random() < 0.6 else rand_seq(rng, 2 * L + 60)
            sc.aligned_pair(2, int(rng.integers(100, 40000)), frag=2 * L + 60, alt_seq=frag, suffix=suffix)
        sc.aligned_pair(3, 500, frag=2 * L + 60, alt_seq=mutate(rng, sc.genes[0][5][100:100 + 2 * L + 60], 0.02), suffix=suffix)
    if with_unaligned:
        for i in range(n):
            r = rng.random()
            g = int(rng.integers(0, len(sc.genes)))
            if r < 0.35:
                s1, s2 = sc.gene_read(g), revcomp(sc.gene_read(g))
            elif r < 0.5:
                s1, s2 = rand_seq(rng, L), revcomp(sc.gene_read(g, sub=0.03))
            elif r < 0.6:
                s1, s2 = "ACAC" * (L // 4), sc.gene_read(g)             # a low-complexity mate rejects the pair
            elif r < 0.65:
                s1, s2 = sc.gene_read(g)[:L // 2] + "N" * (L - L // 2), sc.gene_read(g)
            else:
                s1, s2 = rand_seq(rng, L), rand_seq(rng, L)
            sc.unaligned_pair(s1, s2, first_first=rng.random() < 0.7, suffix=suffix)
    return sc


def single_scenario(seed, with_unaligned=True, with_alt=True, barcodes=False, n=80):
    sc = Scenario(seed, paired=False, barcodes=barcodes)
    rng = sc.rng
    L = sc.read_len
    for _ in range(n):
        name, chrom, s, e, strand, seq = sc.genes[int(rng.integers(0, len(sc.genes)))]
        pos = int(rng.integers(s - 200, e + 50))
        cigar = [("M", 30), ("N", 700), ("M", L - 30)] if rng.random() < 0.1 else None
        sc.single(1, pos, reverse=rng.random() < 0.5, cigar=cigar)
    for _ in range(n // 2):
        sc.single(0, int(rng.integers(1000, 250000)))
    dup = sc.single(1, sc.genes[0][2] + 20)                 # the same read id twice over a gene: written once
    sc.single(1, sc.genes[0][2] + 60, name=dup, flag_extra=0x100)
    if with_alt:
        for _ in range(n // 3):
            g = int(rng.integers(0, len(sc.genes)))
            s = sc.gene_read(g) if rng.random() < 0.6 else rand_seq(rng, L)
            nm = sc.single(2, int(rng.integers(100, 40000)), seq=s, reverse=rng.random() < 0.5)
            if rng.random() < 0.3:  # multiply aligned on the alternative contigs: one output
                sc.single(3, int(rng.integers(100, 15000)), seq=s, name=nm, flag_extra=0x100)
    if with_unaligned:
        for _ in range(n):
            r = rng.random()
            g = int(rng.integers(0, len(sc.genes)))
            s = sc.gene_read(g) if r < 0.4 else revcomp(sc.gene_read(g, sub=0.03)) if r < 0.55 else "GT" * (L // 2) if r < 0.62 else rand_seq(rng, L)
            sc.single(-1, 0, seq=s)
    return sc
