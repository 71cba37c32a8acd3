"""Test infrastructure: a seeded writer of odd and damaged FASTQ / FASTA text for the differential tests of the read-file index against the
reference's own reader (oracle/_ref/reads_harness = ReadFiles.hpp + kseq.h compiled from /root/reference by oracle/Makefile).  A variant is
a small file of records of mixed lengths (one- and several-line sequences, /1 /2 suffixes, comments) with one to three of: a line removed,
doubled, emptied, cut, lengthened; header characters ('@', '+', '>') put in front of a line or taken away; lower case and non-ACGT letters;
CR at line ends, blanks in front of and behind a line, a NUL byte, a VT / FF / CR inside a line; a lone '@'; a very long line; then written with LF or CRLF, with or
without the last line end, with empty lines behind it, or cut at a random byte."""
import random


def base_records(rng, n=24, fasta=False, wrap=0):
    out = []
    for i in range(n):
        L = rng.choice([1, 5, 36, 75, 100, 150, 151, 250])
        s = "".join(rng.choice("ACGT") for _ in range(L))
        if rng.random() < 0.2:
            s = s[:L // 2] + "N" + s[L // 2 + 1:]
        name = "r%d" % i + rng.choice(["", "/1", "/2", " comment here", "\tx"])
        if fasta:
            out.append(">" + name)
            out += [s[j:j + wrap] for j in range(0, L, wrap)] if wrap else [s]
        else:
            q = "".join(chr(rng.randint(33, 73)) for _ in range(L))
            out.append("@" + name)
            if wrap:
                out += [s[j:j + wrap] for j in range(0, L, wrap)] + ["+"] + [q[j:j + wrap] for j in range(0, L, wrap)]
            else:
                out += [s, "+" + rng.choice(["", "r%d" % i]), q]
    return out


def damage(rng, lines):
    lines = list(lines)
    kinds = []
    for _ in range(rng.randint(1, 3)):
        if not lines:
            break
        k = rng.randint(0, 18)
        kinds.append(k)
        i = rng.randrange(len(lines))
        if k == 0: del lines[i]
        elif k == 1: lines.insert(i, lines[i])
        elif k == 2: lines.insert(i, "")
        elif k == 3: lines[i] = lines[i][:rng.randint(0, len(lines[i]))]
        elif k == 4: lines[i] = lines[i] + "".join(rng.choice("ACGT@+>I ") for _ in range(rng.randint(1, 9)))
        elif k == 5: lines[i] = "@" + lines[i]
        elif k == 6: lines[i] = "+" + lines[i]
        elif k == 7: lines[i] = ">" + lines[i]
        elif k == 8: lines[i] = lines[i].lower()
        elif k == 9: lines[i] = lines[i].replace("A", "R").replace("C", ".")
        elif k == 10: lines[i] = lines[i] + "\r"
        elif k == 11: lines[i] = " " + lines[i]
        elif k == 12: lines[i] = lines[i] + " \t "
        elif k == 13: lines[i] = lines[i][:len(lines[i]) // 2] + "\0" + lines[i][len(lines[i]) // 2:]
        elif k == 14: lines[i] = "@"
        elif k == 15: lines[i] = "A" * rng.choice([1000, 5000, 70000])
        elif k == 16: lines.insert(i, "\t")
        elif k == 17: lines[i] = lines[i][1:]
        elif k == 18:
            j = rng.randint(0, len(lines[i]))
            lines[i] = lines[i][:j] + rng.choice("\v\f\r") + lines[i][j:]
    return lines, kinds


def render(rng, lines):
    mode = rng.randint(0, 5)
    nl = "\r\n" if mode == 0 else "\n"
    t = nl.join(lines)
    if mode != 1:
        t += nl
    if mode == 2:
        t += "\n\n"
    b = t.encode("latin1")
    if mode == 3 and len(b) > 2:
        b = b[:rng.randrange(len(b))]
    return b


def variant(seed, undamaged_share=0.1):
    """(bytes of the file, what was done to it)"""
    rng = random.Random(seed)
    fasta = rng.random() < 0.25
    wrap = rng.choice([0, 0, 0, 60, 7])
    lines = base_records(rng, fasta=fasta, wrap=wrap)
    kinds = []
    if rng.random() >= undamaged_share:
        lines, kinds = damage(rng, lines)
    return render(rng, lines), ("fasta" if fasta else "fastq", wrap, kinds)
