"""CPU tests of the host side's read-file index (t1k_amd/csrc/host/reads.cpp, plain C++): the index a single process builds over
the whole input against the indexes N process ranks build over their own fragments only (ReadInput::openSharded: newline counts
per MiB block all-gathered, every rank cuts records in its own slice).  The ranks run as threads of a small harness
(tests/harness/reads_shard_harness.cpp, built with g++ here); the GPU suite covers the same path end to end."""
import os
import subprocess

import pytest

import util

HARNESS_SRC = os.path.join(util.ROOT, "tests", "harness", "reads_shard_harness.cpp")
HOST = os.path.join(util.ROOT, "t1k_amd", "csrc", "host")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("harness") / "reads_shard_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, HARNESS_SRC, os.path.join(HOST, "reads.cpp"), os.path.join(HOST, "refset.cpp"), os.path.join(HOST, "inflate.cpp"),
                    "-lz", "-lpthread", "-ldl"], check=True)
    return exe


@pytest.fixture(scope="module")
def read_sets(built, tmp_path_factory):
    d = str(tmp_path_factory.mktemp("reads"))
    ref = os.path.join(d, "ref.fa")
    util.synth_ref("ref-rna", ref, genes=3, scale=0.02, seed=5)
    util.synth_reads(ref, os.path.join(d, "a"), pairs=30000, len=150, seed=7)   # 9 MB per mate: several 1 MiB blocks per rank
    util.synth_reads(ref, os.path.join(d, "b"), pairs=12345, len=100, seed=8)
    util.synth_reads(ref, os.path.join(d, "c"), pairs=7, len=150, seed=9)       # smaller than a block
    for m in ("1", "2"):
        p = os.path.join(d, "b_%s.fq" % m)
        text = open(p, "rb").read()
        open(p, "wb").write(text.replace(b"\n", b"\r\n"))                       # CRLF
    p = os.path.join(d, "a_1.fq")
    text = open(p, "rb").read()
    open(p, "wb").write(text.rstrip(b"\n"))                                     # no final newline
    open(os.path.join(d, "empty_1.fq"), "w").close()
    open(os.path.join(d, "empty_2.fq"), "w").close()
    return d


@pytest.mark.parametrize("ranks", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("layout", ["one file", "three files + an empty one", "single-end"])
def test_rank_slices_concatenate_to_the_whole_index(harness, read_sets, tmp_path, ranks, layout):
    d = read_sets
    names = {"one file": ["a"], "three files + an empty one": ["a", "empty", "b", "c"], "single-end": ["a", "b"]}[layout]
    f1 = [os.path.join(d, n + "_1.fq") for n in names]
    f2 = [] if layout == "single-end" else [os.path.join(d, n + "_2.fq") for n in names]
    out = str(tmp_path / "o")
    r = subprocess.run([harness, str(ranks), "4", out, str(len(f1))] + f1 + f2, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    whole = open(out + "_whole.tsv").read()
    assert whole.count("\n") == {"one file": 30000, "three files + an empty one": 42352, "single-end": 42345}[layout]
    assert "\r" not in whole
    assert open(out + "_sharded.tsv").read() == whole


def test_more_ranks_than_blocks(harness, read_sets, tmp_path):
    """seven fragments over five ranks: some ranks own nothing, the slices still tile the input"""
    d = read_sets
    out = str(tmp_path / "o")
    r = subprocess.run([harness, "5", "2", out, "1", os.path.join(d, "c_1.fq"), os.path.join(d, "c_2.fq")], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out + "_sharded.tsv").read() == open(out + "_whole.tsv").read()
    assert open(out + "_whole.tsv").read().count("\n") == 7


def test_compressed_input_is_left_to_the_whole_file_reader(harness, read_sets, tmp_path):
    """a .gz file cannot be cut by byte blocks: openSharded declines (0) and the job falls back to every rank reading everything"""
    import gzip
    d = read_sets
    for m in ("1", "2"):
        with open(os.path.join(d, "c_%s.fq" % m), "rb") as a, gzip.open(str(tmp_path / ("c_%s.fq.gz" % m)), "wb") as b:
            b.write(a.read())
    r = subprocess.run([harness, "2", "2", str(tmp_path / "o"), "1", str(tmp_path / "c_1.fq.gz"), str(tmp_path / "c_2.fq.gz")], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 3, r.stderr
    assert open(str(tmp_path / "o") + "_whole.tsv").read().count("\n") == 7


def test_bgzip_framed_and_plain_gzip_input_index_like_the_plain_file(harness, read_sets, tmp_path):
    """a .gz read file written by bgzip (independent 64 KiB gzip members) is inflated block-parallel by the host
    threads; an ordinary .gz is one dependent stream and goes through libdeflate's whole-buffer decoder into reserved memory (member
    after member: concatenated .gz files, zero padding behind the last one), or through gzread where that library is missing
    (T1K_NO_LIBDEFLATE): every way the index equals the plain file's"""
    import gzip
    import bamsynth
    d = read_sets
    out = {}
    for kind in ("plain", "bgzf", "gzip", "gzip_zlib", "gzip_members"):
        files = []
        for m in ("1", "2"):
            src = os.path.join(d, "a_%s.fq" % m)
            if kind == "plain":
                files.append(src)
                continue
            dst = str(tmp_path / ("a_%s_%s.fq.gz" % (kind, m)))
            raw = open(src, "rb").read()
            if kind == "bgzf":
                blob = bamsynth.bgzf(raw)
            elif kind == "gzip_members":  # three members cut at record boundaries (as `cat a.gz b.gz c.gz` gives), then zero padding
                lines = raw.split(b"\n")
                cut1, cut2 = 4 * 7000, 4 * 7001
                parts = [b"\n".join(lines[:cut1]) + b"\n", b"\n".join(lines[cut1:cut2]) + b"\n", b"\n".join(lines[cut2:])]
                blob = b"".join(gzip.compress(x, 6) for x in parts) + b"\0" * 37
            else:
                blob = gzip.compress(raw, 1)
            open(dst, "wb").write(blob)
            files.append(dst)
        o = str(tmp_path / kind)
        env = dict(os.environ, T1K_DEBUG_PHASES="1")
        if kind == "gzip_zlib":
            env["T1K_NO_LIBDEFLATE"] = "1"
        r = subprocess.run([harness, "1", "4", o, "1"] + files, stderr=subprocess.PIPE, text=True, env=env)
        assert r.returncode in (0, 3), r.stderr
        assert ("bgzip-framed read file" in r.stderr) == (kind == "bgzf"), r.stderr
        import ctypes.util
        have = ctypes.util.find_library("deflate") is not None or os.path.exists("/usr/lib/x86_64-linux-gnu/libdeflate.so.0")
        assert ("through libdeflate" in r.stderr) == (kind in ("gzip", "gzip_members") and have), r.stderr
        out[kind] = open(o + "_whole.tsv").read()
    assert out["plain"].count("\n") == 30000
    for kind in out:
        assert out[kind] == out["plain"], kind


# ---- the index against the REFERENCE's reader on odd and damaged files (ReadFiles::Next over kseq) ----
def _ref_records(files):
    r = subprocess.run([util.REF_READS] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-500:]
    return r.stdout


def test_index_equals_the_reference_reader_on_odd_and_damaged_files(built, harness, tmp_path):
    """1 200 seeded files (tests/damaged_reads.py): what ReadInput::open indexes -- in place where the layout is the strict one, through the
    general reader (host/refset.cpp: the rules of kseq.h:185-224 as a state machine) where it is not -- must be the records the reference's
    own ReadFiles::Next hands out (oracle/_ref/reads_harness), ids and sequences byte for byte: a record whose quality string has another
    length ENDS the file there, the next record after a FASTQ record starts at the next '@' or '>' wherever it stands, a CR goes only from
    lines of more than one character, text in front of the first header is skipped.  Where the file also qualifies for rank slices (three
    ranks as threads), their concatenation is compared as well.  Every third case reads two files back to back, the first one damaged:
    the reader goes on with the second (ReadFiles.hpp:161-164)."""
    import damaged_reads
    util.need(util.REF_READS)
    refused = sliced = 0
    for v in range(1200):
        data, what = damaged_reads.variant(1000003 * 7 + v)
        files = [str(tmp_path / "v.fq")]
        open(files[0], "wb").write(data)
        if v % 3 == 2:
            data2, what2 = damaged_reads.variant(1000003 * 11 + v, undamaged_share=0.7)
            files.append(str(tmp_path / "w.fq"))
            open(files[1], "wb").write(data2)
        want = _ref_records(files)
        # (the restated reader of oracle/ on the same files: the checker of the GPU parity tests is pinned here as well)
        assert subprocess.run([util.ORACLE_CLI, "--dumpReads"] + files, stdout=subprocess.PIPE, check=True).stdout == want, (v, what, "oracle")
        out = str(tmp_path / "o")
        r = subprocess.run([harness, "3", "2", out, str(len(files))] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode == 1 and "open:" in r.stderr:
            refused += 1
            continue
        assert r.returncode in (0, 3), (v, what, r.stderr[-500:])
        assert open(out + "_whole.tsv", "rb").read() == want, (v, what)
        if r.returncode == 0:
            sliced += 1
            assert open(out + "_sharded.tsv", "rb").read() == want, (v, what)
    assert refused <= 6 and sliced >= 50, (refused, sliced)


def test_streamed_gz_index_equals_the_reference_reader_on_damaged_text(stream_harness, tmp_path):
    """the same files as .gz through the streamed reader: what the consumer was handed when the stream finished, and what the whole-file
    open (the job's fallback when the stream gives up) indexes, against the reference's reader (which reads the .gz through zlib as well)"""
    import gzip
    import damaged_reads
    util.need(util.REF_READS)
    streamed = gave_up = 0
    for v in range(300):
        data, what = damaged_reads.variant(1000003 * 13 + v, undamaged_share=0.3)
        p = str(tmp_path / "v.fq.gz")
        open(p, "wb").write(gzip.compress(data, 1))
        want = _ref_records([p])
        out = str(tmp_path / "s")
        for f in (out + "_whole.tsv", out + "_stream.tsv"):
            if os.path.exists(f):
                os.remove(f)
        r = subprocess.run([stream_harness, out, "1", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_STREAM_GZ_MIN_MB="0.0001"))
        assert "whole: ERROR" not in r.stdout, (v, what, r.stdout)
        assert open(out + "_whole.tsv", "rb").read() == want, (v, what)
        if r.returncode == 0 and "stream: " in r.stdout and "fragments" in r.stdout.split("stream: ")[1]:
            streamed += 1
            assert open(out + "_stream.tsv", "rb").read() == want, (v, what, r.stdout)
        else:
            gave_up += 1
            assert "not eligible" in r.stdout or "gave up" in r.stdout, (v, what, r.stdout)
    assert streamed >= 30 and gave_up >= 30, (streamed, gave_up)


# ---- streamed .gz input (ReadInput::openStreaming + host/inflate.cpp): the index the window loop reads while the files are still being inflated ----
STREAM_SRC = os.path.join(util.ROOT, "tests", "harness", "reads_stream_harness.cpp")


@pytest.fixture(scope="module")
def stream_harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("harness") / "reads_stream_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, STREAM_SRC, os.path.join(HOST, "reads.cpp"), os.path.join(HOST, "refset.cpp"), os.path.join(HOST, "inflate.cpp"),
                    "-lz", "-lpthread", "-ldl"], check=True)
    return exe


def _stream(exe, prefix, files, min_mb="0.0001", per_mate=1, barcode=None):
    env = dict(os.environ, T1K_STREAM_GZ_MIN_MB=min_mb)
    if barcode:
        env["HARNESS_BARCODE"] = barcode
    r = subprocess.run([exe, prefix, str(per_mate)] + files, stdout=subprocess.PIPE, text=True, env=env)
    return r.returncode, r.stdout


def _gz(src, dst, level=6):
    import gzip
    with open(src, "rb") as f, gzip.open(dst, "wb", compresslevel=level) as g:
        g.write(f.read())
    return dst


@pytest.mark.parametrize("sample", ["a", "b", "c"])   # a: no final newline in mate 1; b: CRLF, 100-base reads; c: seven pairs
def test_streamed_gz_index_equals_the_index_of_the_file_opened_whole(stream_harness, read_sets, tmp_path, sample):
    f = [_gz(os.path.join(read_sets, "%s_%s.fq" % (sample, m)), str(tmp_path / ("%s_%s.fq.gz" % (sample, m))), level) for m, level in (("1", 6), ("2", 1))]
    rc, out = _stream(stream_harness, str(tmp_path / "p"), f)
    assert rc == 0 and "stream:" in out and "ERROR" not in out and "not eligible" not in out, out
    assert open(str(tmp_path / "p_whole.tsv"), "rb").read() == open(str(tmp_path / "p_stream.tsv"), "rb").read()
    whole, stream = [l for l in out.splitlines() if l.startswith("whole:")][0], [l for l in out.splitlines() if l.startswith("stream:")][0]
    assert whole.split("fragments")[0].split()[-1] == stream.split("fragments")[0].split()[-1]
    assert whole.split("longest read")[1].split()[0].rstrip(",") == stream.split("longest read")[1].split()[0].rstrip(",")
    # single-end, and reads that get shorter towards the end of the file (the tables are sized from the head of the text)
    rc, out = _stream(stream_harness, str(tmp_path / "s"), f[:1])
    assert rc == 0 and open(str(tmp_path / "s_whole.tsv"), "rb").read() == open(str(tmp_path / "s_stream.tsv"), "rb").read(), out


def test_streamed_gz_several_files_per_mate(stream_harness, read_sets, tmp_path):
    """lanes: the files of a mate are read back to back (ReadFiles::currentFpInd); a file may end without a line end (a_1) or with CRLF (b), and
    the next file's first record must not be glued to it"""
    f = [_gz(os.path.join(read_sets, "%s_%s.fq" % (smp, m)), str(tmp_path / ("%s_%s.fq.gz" % (smp, m)))) for m in ("1", "2") for smp in ("a", "b", "c", "a")]
    rc, out = _stream(stream_harness, str(tmp_path / "l"), f, per_mate=4)
    assert rc == 0 and "ERROR" not in out and "not eligible" not in out, out
    assert open(str(tmp_path / "l_whole.tsv"), "rb").read() == open(str(tmp_path / "l_stream.tsv"), "rb").read()
    assert "72352 fragments" in out.split("stream:")[1], out   # 30000 + 12345 + 7 + 30000
    rc, out = _stream(stream_harness, str(tmp_path / "u"), f[:4] + f[4:6], per_mate=4)   # mates with different numbers of files: not this reader's case
    assert rc == 0 and "not eligible" in out, out


def test_streamed_gz_with_a_barcode_file(stream_harness, read_sets, tmp_path):
    """a barcode file (two-line FASTA as fastq-extractor writes it, or FASTQ) streamed beside the mates: fragments are the records whose
    barcode is not "missing_barcode" (Genotyper.cpp:376-381), numbered by a thread of their own behind the three indexers"""
    import gzip
    import random
    rnd = random.Random(9)
    n = 30000
    lines = []
    for i in range(n):
        b = "missing_barcode" if rnd.random() < 0.2 else "".join(rnd.choice("ACGT") for _ in range(16))
        lines.append(">r%d\n%s\n" % (i, b))
    f = [_gz(os.path.join(read_sets, "a_%s.fq" % m), str(tmp_path / ("a_%s.fq.gz" % m))) for m in ("1", "2")]
    for kind in ("fa", "fq"):
        text = "".join(lines) if kind == "fa" else "".join(l.replace(">", "@", 1) + "+\n" + "I" * (len(l.split("\n")[1])) + "\n" for l in lines)
        bcp = str(tmp_path / ("bc.%s.gz" % kind))
        with gzip.open(bcp, "wb") as g:
            g.write(text.encode())
        rc, out = _stream(stream_harness, str(tmp_path / ("b" + kind)), f, barcode=bcp)
        assert rc == 0 and "ERROR" not in out and "not eligible" not in out, out
        assert open(str(tmp_path / ("b%s_whole.tsv" % kind)), "rb").read() == open(str(tmp_path / ("b%s_stream.tsv" % kind)), "rb").read()
        kept = sum(1 for l in lines if "missing_barcode" not in l)
        assert ("%d fragments" % kept) in out.split("stream:")[1], out
    # a barcode file with fewer records than the mates: an error at the end of the stream, as for files opened whole
    with gzip.open(str(tmp_path / "short.fa.gz"), "wb") as g:
        g.write("".join(lines[:n - 5]).encode())
    rc, out = _stream(stream_harness, str(tmp_path / "bs"), f, barcode=str(tmp_path / "short.fa.gz"))
    assert rc == 1 and "different numbers of records" in out, out
    # a plain (not compressed) barcode file beside .gz mates: opened whole
    open(str(tmp_path / "plain.fa"), "w").write("".join(lines))
    rc, out = _stream(stream_harness, str(tmp_path / "bp"), f, barcode=str(tmp_path / "plain.fa"))
    assert rc == 0 and "not eligible" in out, out


def test_streamed_gz_trimmed_reads_and_trailing_blank_lines(stream_harness, tmp_path):
    import random
    rnd = random.Random(3)
    recs = []
    for i in range(40000):
        n = 150 if i < 5000 else rnd.choice((31, 40, 75, 150))   # the head of the file says 150, the rest is trimmed
        s = "".join(rnd.choice("ACGTN") for _ in range(n))
        recs.append("@q%d extra words\n%s\n+\n%s\n" % (i, s, "".join(rnd.choice("@+FF:,") for _ in range(n))))  # quality lines that start with '@' and '+'
    open(str(tmp_path / "t.fq"), "w").write("".join(recs) + "\n\n")
    g = _gz(str(tmp_path / "t.fq"), str(tmp_path / "t.fq.gz"))
    rc, out = _stream(stream_harness, str(tmp_path / "t"), [g])
    assert rc == 0 and "ERROR" not in out and "not eligible" not in out, out
    assert open(str(tmp_path / "t_whole.tsv"), "rb").read() == open(str(tmp_path / "t_stream.tsv"), "rb").read()


def test_streamed_gz_refuses_what_it_cannot_follow_and_reports_damage(stream_harness, read_sets, tmp_path):
    a1 = _gz(os.path.join(read_sets, "a_1.fq"), str(tmp_path / "a_1.fq.gz"))
    a2 = _gz(os.path.join(read_sets, "a_2.fq"), str(tmp_path / "a_2.fq.gz"))
    c2 = _gz(os.path.join(read_sets, "c_2.fq"), str(tmp_path / "c_2.fq.gz"))
    # below the size from which streaming pays: opened whole (default threshold)
    rc, out = _stream(stream_harness, str(tmp_path / "n"), [a1, a2], min_mb="32")
    assert rc == 0 and "not eligible" in out, out
    # a plain file, a FASTA file, two members are fine, wrapped records are not the four-line layout
    rc, out = _stream(stream_harness, str(tmp_path / "n"), [os.path.join(read_sets, "a_1.fq")])
    assert rc == 0 and "not eligible" in out, out
    open(str(tmp_path / "w.fa"), "w").write("".join(">s%d\nACGTACGT\nACGT\n" % i for i in range(5000)))
    rc, out = _stream(stream_harness, str(tmp_path / "n"), [_gz(str(tmp_path / "w.fa"), str(tmp_path / "w.fa.gz"))])
    assert rc == 0 and "not eligible" in out, out
    # mates of different length: an error at the end of the stream, as for files opened whole
    rc, out = _stream(stream_harness, str(tmp_path / "m"), [a1, c2])
    assert rc == 1 and "different numbers of reads" in out, out
    # damage: a flipped bit in the trailer's CRC, a flipped byte in the compressed data, a truncated file
    blob = bytearray(open(a1, "rb").read())
    bad = bytearray(blob); bad[-6] ^= 0x10
    open(str(tmp_path / "crc.fq.gz"), "wb").write(bad)
    rc, out = _stream(stream_harness, str(tmp_path / "d"), [str(tmp_path / "crc.fq.gz")])
    assert rc == 1 and "damaged" in out, out
    bad = bytearray(blob); bad[len(bad) // 2] ^= 0x55
    open(str(tmp_path / "mid.fq.gz"), "wb").write(bad)
    rc, out = _stream(stream_harness, str(tmp_path / "d"), [str(tmp_path / "mid.fq.gz")])
    assert rc == 1 and "stream: ERROR" in out, out
    open(str(tmp_path / "cut.fq.gz"), "wb").write(blob[:len(blob) * 2 // 3])
    rc, out = _stream(stream_harness, str(tmp_path / "d"), [str(tmp_path / "cut.fq.gz")])
    assert rc == 1 and "stream: ERROR" in out, out


def test_streamed_gz_gives_up_on_text_the_whole_reader_takes(stream_harness, read_sets, tmp_path):
    """a blank line between two records / a last record without quality lines, behind the head the eligibility check looks at: the
    stream ends with the flag that makes t1k_job_run open the files whole (the whole reader indexes both texts)"""
    import gzip
    lines = open(os.path.join(read_sets, "a_2.fq")).read().split("\n")
    n = (len(lines) - 1) // 4
    texts = {"blank": "\n".join(lines[:4 * (n // 2)]) + "\n\n" + "\n".join(lines[4 * (n // 2):]), "tail": "\n".join(lines[:4 * n - 2]) + "\n"}
    for tag, t in texts.items():
        p = str(tmp_path / (tag + ".fq.gz"))
        with gzip.open(p, "wb") as g:
            g.write(t.encode())
        env = dict(os.environ, T1K_STREAM_GZ_MIN_MB="0.0001", T1K_STREAM_HEAD_MB="0.01")
        r = subprocess.run([stream_harness, str(tmp_path / tag), "1", p], stdout=subprocess.PIPE, text=True, env=env)
        assert r.returncode == 1 and "whole: %d fragments" % n in r.stdout and "gave up" in r.stdout, r.stdout
        r = subprocess.run([stream_harness, str(tmp_path / tag), "1", p], stdout=subprocess.PIPE, text=True, env=dict(env, T1K_STREAM_HEAD_MB="64"))
        # (a head that covers the file sees the blank line and declines at once; the short last record is not a whole record of the head)
        assert (r.returncode == 0 and "not eligible" in r.stdout) if tag == "blank" else (r.returncode == 1 and "gave up" in r.stdout), r.stdout
    blob = bytearray(open(str(tmp_path / "blank.fq.gz"), "rb").read())
    blob[len(blob) // 3] ^= 0x55
    open(str(tmp_path / "bad.fq.gz"), "wb").write(blob)
    r = subprocess.run([stream_harness, str(tmp_path / "bad"), "1", str(tmp_path / "bad.fq.gz")], stdout=subprocess.PIPE, text=True,
                       env=dict(os.environ, T1K_STREAM_GZ_MIN_MB="0.0001", T1K_STREAM_HEAD_MB="0.01"))
    # (damage turns into odd text before the decoder or the CRC notice it: the stream may give up first -- the whole reader the job then
    # opens refuses the file, which is the error the caller sees)
    assert r.returncode == 1 and "stream: ERROR" in r.stdout and "whole: ERROR" in r.stdout, r.stdout


def test_streamed_gz_reader_under_thread_sanitizer(read_sets, tmp_path):
    """the streamed reader is three to seven threads around shared tables (decoder, indexer and CRC checker per file role, the fragment
    numbering for a barcode file, the consumer): the same harness built with -fsanitize=thread must report nothing on the plain, the lane
    and the barcode layouts, and on a damaged file (the error paths end the threads early)"""
    import gzip
    import random
    exe = str(tmp_path / "reads_stream_tsan")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-o", exe, STREAM_SRC, os.path.join(HOST, "reads.cpp"), os.path.join(HOST, "refset.cpp"),
                        os.path.join(HOST, "inflate.cpp"), "-lz", "-lpthread", "-ldl"], stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        pytest.skip("no thread sanitizer runtime for g++ here: " + r.stderr[-200:])
    env = dict(os.environ, T1K_STREAM_GZ_MIN_MB="0.0001", TSAN_OPTIONS="halt_on_error=0 exitcode=0")

    def run(prefix, per_mate, files, barcode=None):
        e = dict(env, HARNESS_BARCODE=barcode) if barcode else env
        p = subprocess.run([exe, str(tmp_path / prefix), str(per_mate)] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
        if "FATAL: ThreadSanitizer" in p.stderr:   # (the sanitizer cannot map its shadow memory in some containers)
            pytest.skip(p.stderr[-200:])
        assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[:4000]
        return p

    f = [_gz(os.path.join(read_sets, "a_%s.fq" % m), str(tmp_path / ("a_%s.fq.gz" % m))) for m in ("1", "2")]
    p = run("p", 1, f)
    assert p.returncode == 0 and open(str(tmp_path / "p_whole.tsv"), "rb").read() == open(str(tmp_path / "p_stream.tsv"), "rb").read(), p.stdout
    lanes = [_gz(os.path.join(read_sets, "%s_%s.fq" % (s, m)), str(tmp_path / ("%s_%s.l.fq.gz" % (s, m)))) for m in ("1", "2") for s in ("a", "b")]
    p = run("l", 2, lanes)
    assert p.returncode == 0 and open(str(tmp_path / "l_whole.tsv"), "rb").read() == open(str(tmp_path / "l_stream.tsv"), "rb").read(), p.stdout
    rnd = random.Random(9)
    with gzip.open(str(tmp_path / "bc.fa.gz"), "wb") as g:
        g.write("".join(">r%d\n%s\n" % (i, "missing_barcode" if rnd.random() < 0.2 else "".join(rnd.choice("ACGT") for _ in range(16))) for i in range(30000)).encode())
    p = run("b", 1, f, barcode=str(tmp_path / "bc.fa.gz"))
    assert p.returncode == 0 and open(str(tmp_path / "b_whole.tsv"), "rb").read() == open(str(tmp_path / "b_stream.tsv"), "rb").read(), p.stdout
    blob = bytearray(open(f[0], "rb").read())
    blob[len(blob) // 2] ^= 0x55
    open(str(tmp_path / "mid.fq.gz"), "wb").write(blob)
    p = run("d", 1, [str(tmp_path / "mid.fq.gz"), f[1]])
    assert p.returncode == 1 and "stream: ERROR" in p.stdout, p.stdout


# ---- fastq-extractor's streaming record reader (host/extract.cpp) against the reference's reader ----
def test_extractor_record_reader_equals_the_reference_reader_on_odd_and_damaged_files(tmp_path):
    """fastq-extractor reads its inputs as streams (host/extract.cpp: RecordReader, an in-place four-line path and the general rules of
    kseq.h:185-224 taking turns over one buffer).  600 seeded odd / damaged files (tests/damaged_reads.py), every third case two files
    back to back, each read with a buffer of 4 MiB, 300 and 64 bytes (records cut by the buffer end take the general path): ids and
    sequences must be the ones the reference's ReadFiles::Next hands out (oracle/_ref/reads_harness); tests/harness/extract_reader_harness.cpp"""
    import damaged_reads
    util.need(util.REF_READS)
    exe = str(tmp_path / "extract_reader_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(util.ROOT, "include"), "-o", exe, os.path.join(util.ROOT, "tests", "harness", "extract_reader_harness.cpp"),
                    os.path.join(HOST, "reads.cpp"), os.path.join(HOST, "refset.cpp"), os.path.join(HOST, "inflate.cpp"), "-lz", "-lpthread", "-ldl"], check=True)
    for v in range(600):
        data, what = damaged_reads.variant(1000003 * 17 + v)
        files = [str(tmp_path / "x.fq")]
        open(files[0], "wb").write(data)
        if v % 3 == 2:
            data2, what2 = damaged_reads.variant(1000003 * 19 + v, undamaged_share=0.7)
            files.append(str(tmp_path / "y.fq"))
            open(files[1], "wb").write(data2)
        want = _ref_records(files)
        for buf in ("4194304", "300", "64"):
            r = subprocess.run([exe, buf] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0 and r.stdout == want, (v, buf, what)


# ---- the allele reference's records (RefSet::load) against the reference's reader, comments included ----
def test_reference_fasta_records_equal_the_reference_reader_on_odd_and_damaged_files(tmp_path):
    """SeqSet::InputRefFa reads the allele FASTA through the same ReadFiles::Next (SeqSet.hpp:872-904) and takes the exon coordinates from the
    header's comment.  900 seeded files, four in five FASTA (one-line and wrapped, comments, CRLF, damage of tests/damaged_reads.py):
    readReferenceRecords -- plain '>' records parsed by all host threads, everything else by the general reader -- must give the
    reference reader's ids, sequences, comments and has-a-comment flags (oracle/_ref/reads_harness -c; tests/harness/ref_records_harness.cpp)"""
    import random
    import damaged_reads
    util.need(util.REF_READS)
    exe = str(tmp_path / "ref_records_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(util.ROOT, "tests", "harness", "ref_records_harness.cpp"),
                    os.path.join(HOST, "refset.cpp"), os.path.join(HOST, "reads.cpp"), os.path.join(HOST, "inflate.cpp"), "-lz", "-lpthread", "-ldl"], check=True)
    p = str(tmp_path / "r.fa")
    for v in range(900):
        rng = random.Random(1000003 * 23 + v)
        lines = damaged_reads.base_records(rng, fasta=rng.random() < 0.8, wrap=rng.choice([0, 60, 60, 7]))
        kinds = []
        if rng.random() < 0.85:
            lines, kinds = damaged_reads.damage(rng, lines)
        open(p, "wb").write(damaged_reads.render(rng, lines))
        want = subprocess.run([util.REF_READS, "-c", p], stdout=subprocess.PIPE, check=True).stdout
        r = subprocess.run([exe, p], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == want, (v, kinds)
