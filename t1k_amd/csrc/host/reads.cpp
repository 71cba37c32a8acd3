// t1k_amd/csrc/host/reads.cpp -- the read files of a job, whole in memory with a record index.
//
// The reference streams records through kseq one at a time on one thread (ReadFiles::Next, ReadFiles.hpp:155-204) and strdup()s
// every id and sequence (Genotyper.cpp:365-440).  Here a plain file is mmap()ed (a gz file is inflated once into memory), cut into
// byte ranges, and every host thread indexes its range in place: a record is two pointers and two lengths into the mapping, nothing
// is copied, and ids / sequences / barcodes are read from the mapping again when the *_aligned*.fa files are written.  The in-place
// indexer accepts the two layouts the pipeline produces (four-line FASTQ, two-line FASTA; LF or CRLF); anything else -- wrapped
// sequences, blank lines -- goes through the general record reader (readSeqFile, the kseq rules) into owned storage.
// Several files per mate are read back to back, as ReadFiles does with currentFpInd.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include "t1k_host.h"

namespace t1k {

namespace {

inline const char *lineEnd(const char *p, const char *end) {
  const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
  return nl ? nl : end;
}
inline bool isBlank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

struct Piece {
  std::vector<const char *> seqP, idP;
  std::vector<uint32_t> seqL;
  std::vector<uint16_t> idL;
  bool ok = true;
};

// one record starting at p (a line start); returns the start of the next record, or nullptr if the layout is not the strict one
struct RecFields { const char *id, *seq; size_t idLen, seqLen; };
inline const char *strictRecordFields(const char *p, const char *end, bool fastq, RecFields &f);
inline const char *strictRecord(const char *p, const char *end, bool fastq, Piece &out) {
  RecFields f;
  const char *next = strictRecordFields(p, end, fastq, f);
  if (next) { out.seqP.push_back(f.seq); out.seqL.push_back((uint32_t)f.seqLen); out.idP.push_back(f.id); out.idL.push_back((uint16_t)f.idLen); }
  return next;
}
inline const char *strictRecordFields(const char *p, const char *end, bool fastq, RecFields &f) {
  if (p >= end || *p != (fastq ? '@' : '>')) return nullptr;
  const char *e1 = lineEnd(p, end);
  const char *id = p + 1, *ie = id;
  while (ie < e1 && !isBlank(*ie)) ++ie;
  size_t il = (size_t)(ie - id);
  if (il >= 2 && id[il - 2] == '/' && (id[il - 1] == '1' || id[il - 1] == '2')) il -= 2;  // ReadFiles.hpp:185-189
  if (il > 0xFFFF) return nullptr;
  if (e1 >= end) return nullptr;
  const char *s = e1 + 1;
  const char *e2 = lineEnd(s, end);
  size_t sl = (size_t)(e2 - s);
  if (sl == 1 && s[0] == '\r') return nullptr;  // a line of a lone CR: the reference's reader keeps that character (kseq.h:142 drops a CR only from more than one character)
  if (sl && s[sl - 1] == '\r') --sl;
  if (sl && (s[0] == '>' || s[0] == '@' || s[0] == '+')) return nullptr;
  const char *next = e2 < end ? e2 + 1 : end;
  if (fastq) {
    if (next >= end || *next != '+') return nullptr;
    const char *e3 = lineEnd(next, end);
    if (e3 >= end) return nullptr;
    const char *q = e3 + 1;
    const char *e4 = lineEnd(q, end);
    size_t ql = (size_t)(e4 - q);
    if (ql == 1 && q[0] == '\r') return nullptr;
    if (ql && q[ql - 1] == '\r') --ql;
    if (ql != sl) return nullptr;
    next = e4 < end ? e4 + 1 : end;
  } else {
    if (next < end && *next != '>') return nullptr;  // wrapped sequence or a blank line: not the strict layout
  }
  f.id = id; f.idLen = il; f.seq = s; f.seqLen = sl;
  return next;
}

// first record start at or after byte `from` (from > 0)
inline const char *findRecord(const char *base, const char *from, const char *end, bool fastq) {
  const char *p = lineEnd(from, end);
  p = p < end ? p + 1 : end;
  while (p < end) {
    if (!fastq) { if (*p == '>') return p; }
    else if (*p == '@') {
      // a header is followed two lines later by a '+' line; a quality line that starts with '@' is followed by a header and then a sequence
      const char *e1 = lineEnd(p, end);
      if (e1 < end) {
        const char *e2 = lineEnd(e1 + 1, end);
        if (e2 < end && e2 + 1 < end && e2[1] == '+') return p;
      }
    }
    const char *e = lineEnd(p, end);
    p = e < end ? e + 1 : end;
  }
  (void)base;
  return end;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Streaming .gz input (ReadFiles.hpp:13, 95 and kseq.h:94-150 hand the reference its first read as soon as the file is open).
// Per mate: the compressed file mapped, the text inflated into one reserved range of anonymous memory by host/inflate.cpp (thread 1), the
// four-line FASTQ records indexed in place behind the decoder's published progress (thread 2), the CRC-32 of every member checked behind it
// (thread 3; libdeflate's routine when the image has it -- 10 GB/s -- zlib's otherwise).  Tables sized to an upper bound of the record count.
// ------------------------------------------------------------------------------------------------------------------
struct ReadInput::Stream {
  struct Src { void *map = nullptr; size_t len = 0; std::string path; };
  struct Mate {
    std::vector<Src> srcs;        // the mate's files, read back to back (ReadFiles::currentFpInd)
    std::mutex m;
    std::vector<uint64_t> fileEnds;  // offsets in the text where a file's text ends (pushed before the next file's first byte is published)
    char *text = nullptr; size_t cap = 0;
    GzProgress pg;
    std::thread inflater, indexer, checker;
    std::atomic<uint64_t> records{0};
    std::atomic<int> state{0};  // indexer: 0 running, 1 done, -1 failed
    std::atomic<int> crcState{0};
    std::atomic<uint64_t> crcDone{0};  // text the checker has been over (nothing beyond it may be dropped: ReadInput::release)
    std::string err, errIndex, path;  // (the decoder's message / the indexer's: two threads, two strings)
    size_t textLen = 0;
    bool fastq = true;   // four-line FASTQ (two-line FASTA otherwise: the barcode file fastq-extractor writes)
    Side *dst = nullptr;
  } mate[3];             // the mates, then the barcode file (ReadInput::bc) when there is one
  int nMates = 1;        // text streams in all
  bool withBarcode = false;
  // fragments = records whose barcode is not "missing_barcode" (Genotyper.cpp:376-381): with a barcode file a thread of its own follows the
  // three indexers and numbers them (frag[] and this counter are what the window loop reads)
  std::thread fragger;
  std::atomic<uint64_t> fragments{0};
  std::atomic<int> fragState{0};
  size_t capRecords = 0;
  bool joined = false;
  ~Stream() {
    if (fragger.joinable()) fragger.join();
    for (int m = 0; m < nMates; ++m) {
      Mate &M = mate[m];
      if (M.inflater.joinable()) M.inflater.join();
      if (M.indexer.joinable()) M.indexer.join();
      if (M.checker.joinable()) M.checker.join();
      for (Src &f : M.srcs) if (f.map) munmap(f.map, f.len);
    }
  }
};

ReadInput::ReadInput() {}
ReadInput::~ReadInput() {
  stream_.reset();  // its threads write into the blobs
  for (auto &b : blobs_)
    if (b.map) munmap(b.map, b.len);
}

bool ReadInput::addBuffer(const char *p, size_t n, int threads, Side &dst, std::string &err, const std::string &what) {
  const char *end = p + n;
  while (n && (end[-1] == '\n' || end[-1] == '\r')) { --end; --n; }  // trailing empty lines (blanks at the end of the last line belong to that line: a quality string they make longer ends the file in the reference's reader)
  const char *b = p;
  while (b < end && isBlank(*b)) ++b;
  if (b >= end) return true;  // empty file: no records
  const bool fastq = *b == '@';
  if (!fastq && *b != '>') { (void)what; return false; }  // text in front of the first header: the general reader skips it as the reference's does (kseq.h:189-193)
  return addRange(b, end, end, fastq, threads, dst);  // false with err empty: the caller falls back to the general reader
}

// records of [b, stop) -- b a record start, stop a record start or the end of the text -- indexed in place by `threads` threads;
// false = not the strict layout
bool ReadInput::addRange(const char *b, const char *stop, const char *end, bool fastq, int threads, Side &dst) {
  if (b >= stop) return true;
  const size_t n = (size_t)(stop - b);
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n / (1u << 20) + 1));
  std::vector<const char *> start(T + 1, stop);
  start[0] = b;
  std::vector<Piece> piece(T);
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { start[t] = findRecord(b, b + n / T * t, stop, fastq); });
    for (auto &x : th) x.join();
    for (int t = 1; t <= T; ++t) start[t] = std::max(start[t], start[t - 1]);
    start[T] = stop;
  }
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        Piece &pc = piece[t];
        const char *q = start[t], *pstop = start[t + 1];
        const size_t guess = (size_t)(pstop - q) / 200 + 16;
        pc.seqP.reserve(guess); pc.seqL.reserve(guess); pc.idP.reserve(guess); pc.idL.reserve(guess);
        while (q < pstop) {
          q = strictRecord(q, end, fastq, pc);
          if (!q) { pc.ok = false; return; }
        }
        if (q != pstop) pc.ok = false;
      });
    for (auto &x : th) x.join();
  }
  bool strict = true;
  for (auto &pc : piece) strict = strict && pc.ok;
  if (!strict) return false;
  size_t tot = 0;
  std::vector<size_t> at(T);
  for (int t = 0; t < T; ++t) { at[t] = dst.seqP.size() + tot; tot += piece[t].seqP.size(); }
  const size_t old = dst.seqP.size();
  dst.seqP.resize(old + tot); dst.seqL.resize(old + tot); dst.idP.resize(old + tot); dst.idL.resize(old + tot);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        const Piece &pc = piece[t];
        const size_t m = pc.seqP.size();
        if (!m) return;
        memcpy(&dst.seqP[at[t]], pc.seqP.data(), m * sizeof(char *)); memcpy(&dst.seqL[at[t]], pc.seqL.data(), m * 4);
        memcpy(&dst.idP[at[t]], pc.idP.data(), m * sizeof(char *)); memcpy(&dst.idL[at[t]], pc.idL.data(), m * 2);
      });
    for (auto &x : th) x.join();
  }
  return true;
}

// the general reader's records, copied into owned storage
bool ReadInput::addGeneral(const std::string &path, Side &dst, std::string &err) {
  std::vector<SeqRec> recs;
  inPlace = false;
  if (!readSeqFile(path, recs, err)) return false;
  size_t bytes = 0;
  for (auto &r : recs) bytes += r.id.size() + r.seq.size();
  Blob &b = newBlob();
  b.owned.reset(new std::vector<char>(bytes + 1));
  char *w = b.owned->data();
  for (auto &r : recs) {
    if (r.id.size() > 0xFFFF) { err = path + ": record name longer than 65535 characters"; return false; }
    memcpy(w, r.id.data(), r.id.size());
    dst.idP.push_back(w); dst.idL.push_back((uint16_t)r.id.size());
    w += r.id.size();
    memcpy(w, r.seq.data(), r.seq.size());
    dst.seqP.push_back(w); dst.seqL.push_back((uint32_t)r.seq.size());
    w += r.seq.size();
  }
  return true;
}

// A .gz file written by bgzip (BGZF: independent gzip members of at most 64 KiB, each carrying its compressed size in a "BC" extra
// field) can be inflated block-parallel; ordinary gzip is one dependent stream and goes through gzread.  False = not BGZF (or damaged):
// the caller falls back to gzread, which reports real damage.
bool ReadInput::bgzfInflate(int fd, size_t fileSize, int threads, Blob &blob, const char *&data, size_t &size) {
  if (fileSize < 28) return false;
  void *m = mmap(nullptr, fileSize, PROT_READ, MAP_PRIVATE, fd, 0);
  if (m == MAP_FAILED) return false;
  const uint8_t *map = (const uint8_t *)m;
  struct Blk { size_t in, inLen, out, outLen; };
  std::vector<Blk> blks;
  size_t pos = 0, out = 0;
  bool ok = true;
  while (pos < fileSize) {
    if (fileSize - pos < 28 || map[pos] != 31 || map[pos + 1] != 139 || map[pos + 2] != 8 || !(map[pos + 3] & 4)) { ok = false; break; }
    const size_t xlen = map[pos + 10] | (map[pos + 11] << 8);
    size_t bsize = 0;
    for (size_t x = pos + 12; x + 4 <= pos + 12 + xlen && x + 6 <= fileSize;) {
      const size_t slen = map[x + 2] | (map[x + 3] << 8);
      if (map[x] == 'B' && map[x + 1] == 'C' && slen == 2) bsize = (size_t)(map[x + 4] | (map[x + 5] << 8)) + 1;
      x += 4 + slen;
    }
    if (!bsize || pos + bsize > fileSize || bsize < xlen + 20) { ok = false; break; }
    const uint8_t *tail = map + pos + bsize - 4;
    const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
    blks.push_back({pos + 12 + xlen, bsize - xlen - 20, out, isize});
    out += isize; pos += bsize;
  }
  if (ok && !blks.empty()) {
    blob.owned.reset(new std::vector<char>());
    std::vector<char> &v = *blob.owned;
    v.resize(out);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&] {
      for (size_t i = next.fetch_add(16); i < blks.size(); i = next.fetch_add(16))
        for (size_t j = i; j < std::min(blks.size(), i + 16); ++j) {
          if (!blks[j].outLen) continue;
          z_stream zs;
          memset(&zs, 0, sizeof(zs));
          if (inflateInit2(&zs, -15) != Z_OK) { bad = true; continue; }
          zs.next_in = (Bytef *)(map + blks[j].in); zs.avail_in = (uInt)blks[j].inLen;
          zs.next_out = (Bytef *)v.data() + blks[j].out; zs.avail_out = (uInt)blks[j].outLen;
          if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) bad = true;
          inflateEnd(&zs);
        }
    };
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), blks.size() / 16 + 1));
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    ok = !bad;
    if (ok && getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k job] bgzip-framed read file: %zu blocks inflated by %d threads\n", blks.size(), T);
    if (ok) { data = v.data(); size = v.size(); }
    else blob.owned.reset();
  } else ok = false;
  munmap(m, fileSize);
  return ok;
}

// An ordinary gzip file is one dependent stream: nothing inflates it block-parallel.  What can be had is a faster inflater and no
// copies: the compressed file is mapped, the text goes straight into one reserved range of anonymous memory (only the pages that are
// written get backed), member after member (concatenated .gz files are legal), through libdeflate's whole-buffer decoder -- 2 - 3 x
// zlib's inflate on FASTQ text -- which this image ships as a runtime library (libdeflate.so.0, bound lazily like librccl; no header is
// needed for three entry points).  False = not available / the text does not fit the reservation / damaged: the caller goes through
// gzread, which also reports real damage.  (The reference reads the same files through gzopen + kseq, ReadFiles.hpp:23-282.)
namespace {
struct Deflate {
  void *(*alloc)() = nullptr;
  int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
  void (*release)(void *) = nullptr;
  static const Deflate *get() {
    static const Deflate d = [] {
      Deflate x;
      if (getenv("T1K_NO_LIBDEFLATE")) return x;
      void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
      if (!h) return x;
      x.alloc = (decltype(x.alloc))dlsym(h, "libdeflate_alloc_decompressor");
      x.gunzip = (decltype(x.gunzip))dlsym(h, "libdeflate_gzip_decompress_ex");
      x.release = (decltype(x.release))dlsym(h, "libdeflate_free_decompressor");
      if (!x.alloc || !x.gunzip || !x.release) x = Deflate();
      return x;
    }();
    return d.gunzip ? &d : nullptr;
  }
};
}  // namespace

bool ReadInput::gzipInflate(int fd, size_t fileSize, Blob &blob, const char *&data, size_t &size) {
  const Deflate *z = Deflate::get();
  if (!z || fileSize < 18) return false;
  void *m = mmap(nullptr, fileSize, PROT_READ, MAP_PRIVATE, fd, 0);
  if (m == MAP_FAILED) return false;
  (void)madvise(m, fileSize, MADV_SEQUENTIAL);
  // room for the text: 48 x the compressed size (FASTQ deflates 3 - 6 x; a file that beats 48 x takes the gzread path), address space only
  const size_t cap = ((fileSize * 48 + (64u << 20)) + 4095) & ~(size_t)4095;
  void *out = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (out == MAP_FAILED) { munmap(m, fileSize); return false; }
  bigBlockAdvise(out, cap);
  void *d = z->alloc();
  bool ok = d != nullptr;
  size_t in = 0, used = 0;
  const uint8_t *src = (const uint8_t *)m;
  while (ok && in < fileSize) {
    if (fileSize - in < 18 || src[in] != 0x1f || src[in + 1] != 0x8b) {  // trailing garbage after the last member: gzread decides what it means
      bool zeros = true;
      for (size_t i = in; i < fileSize && zeros; ++i) zeros = src[i] == 0;  // (zero padding is legal and ignored, as gzip does)
      ok = zeros;
      break;
    }
    size_t ate = 0, made = 0;
    const int r = z->gunzip(d, src + in, fileSize - in, (char *)out + used, cap - used, &ate, &made);
    if (r != 0 || ate == 0) { ok = false; break; }
    in += ate; used += made;
  }
  if (d) z->release(d);
  munmap(m, fileSize);
  if (!ok) { munmap(out, cap); return false; }
  // give the unused tail of the reservation back; the blob owns the rest like a mapped file
  const size_t keep = std::max<size_t>(4096, (used + 4095) & ~(size_t)4095);
  if (keep < cap) munmap((char *)out + keep, cap - keep);
  blob.map = out; blob.len = keep; blob.anon = true;
  data = (const char *)out; size = used;
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k job] gzip read file: %zu -> %zu bytes through libdeflate\n", fileSize, used);
  return true;
}

bool ReadInput::addFile(const std::string &path, int threads, Side &dst, std::string &err) {
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) { err = "cannot open " + path; return false; }
  struct stat st;
  if (fstat(fd, &st) != 0) { ::close(fd); err = "cannot stat " + path; return false; }
  unsigned char magic[2] = {0, 0};
  const bool gz = st.st_size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
  const char *data = nullptr;
  size_t size = 0;
  if (!gz && S_ISREG(st.st_mode)) {
    if (st.st_size == 0) { ::close(fd); return true; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { err = "cannot map " + path; return false; }
    (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
    Blob &b = newBlob();
    b.map = m; b.len = (size_t)st.st_size;
    data = (const char *)m; size = (size_t)st.st_size;
  } else if (gz && S_ISREG(st.st_mode) && [&] { Blob &b = newBlob(); if (bgzfInflate(fd, (size_t)st.st_size, threads, b, data, size)) return true; dropBlob(b); return false; }()) {
    // (a bgzip-framed file: its 64 KiB blocks were inflated side by side by the host threads)
    ::close(fd);
  } else if (gz && S_ISREG(st.st_mode) && [&] { Blob &b = newBlob(); if (gzipInflate(fd, (size_t)st.st_size, b, data, size)) return true; dropBlob(b); return false; }()) {
    // (ordinary gzip: one stream, inflated by libdeflate straight into reserved memory)
    ::close(fd);
  } else {
    ::close(fd);
    gzFile fp = gzopen(path.c_str(), "rb");  // also reads a plain stream (a pipe) transparently
    if (!fp) { err = "cannot open " + path; return false; }
    gzbuffer(fp, 1 << 20);
    Blob &b = newBlob();
    b.owned.reset(new std::vector<char>());
    std::vector<char> &v = *b.owned;
    size_t used = 0;
    v.resize(64u << 20);
    for (;;) {
      if (v.size() - used < (16u << 20)) v.resize(v.size() * 2);
      int got = gzread(fp, v.data() + used, (unsigned)std::min<size_t>(v.size() - used, 1u << 30));
      if (got < 0) { gzclose(fp); err = "cannot read " + path; return false; }
      if (got == 0) break;
      used += (size_t)got;
    }
    gzclose(fp);
    v.resize(used);
    data = v.data(); size = used;
  }
  const size_t before = dst.seqP.size();
  std::string e2;
  if (addBuffer(data, size, threads, dst, e2, path)) return true;
  if (!e2.empty()) { err = e2; return false; }
  dst.seqP.resize(before); dst.seqL.resize(before); dst.idP.resize(before); dst.idL.resize(before);
  return addGeneral(path, dst, err);
}

namespace {
// CRC-32 of a piece of text: libdeflate's (carry-less multiplication, ~10 GB/s) when the image has the library, zlib's otherwise
uint32_t crcStep(uint32_t crc, const void *p, size_t n) {
  static uint32_t (*fast)(uint32_t, const void *, size_t) = [] {
    void *h = getenv("T1K_NO_LIBDEFLATE") ? nullptr : dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    return h ? (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32") : nullptr;
  }();
  if (fast) return fast(crc, p, n);
  const unsigned char *q = (const unsigned char *)p;
  while (n) { const size_t k = std::min<size_t>(n, 1u << 30); crc = (uint32_t)crc32(crc, q, (uInt)k); q += k; n -= k; }
  return crc;
}
}  // namespace

size_t ReadInput::streamAvail() const {
  if (!stream_) return frag.size();
  if (stream_->withBarcode) return (size_t)stream_->fragments.load(std::memory_order_acquire);
  uint64_t n = stream_->mate[0].records.load(std::memory_order_acquire);
  if (stream_->nMates > 1) n = std::min<uint64_t>(n, stream_->mate[1].records.load(std::memory_order_acquire));
  return (size_t)n;
}
int ReadInput::streamState() const {
  if (!stream_) return 1;
  int all = 1;
  for (int m = 0; m < stream_->nMates; ++m) {
    const int a = stream_->mate[m].state.load(std::memory_order_acquire), c = stream_->mate[m].crcState.load(std::memory_order_acquire);
    if (a < 0 || c < 0) return -1;
    if (a == 0 || c == 0) all = 0;
  }
  if (stream_->withBarcode && stream_->fragState.load(std::memory_order_acquire) == 0) all = 0;
  return all;
}
void ReadInput::streamWait(size_t records) const {
  while (streamAvail() < records && streamState() == 0) std::this_thread::sleep_for(std::chrono::microseconds(300));
}

bool ReadInput::openStreaming(const std::vector<std::string> &files1, const std::vector<std::string> &files2, const std::string &barcodeFile, std::string &err) {
  err.clear();
  if (files1.empty() || (!files2.empty() && files2.size() != files1.size())) return false;
  streamFiles1 = files1; streamFiles2 = files2; streamBarcodeFile = barcodeFile;
  static const size_t minBytes = [] { const char *e = getenv("T1K_STREAM_GZ_MIN_MB"); return (size_t)((e ? atof(e) : 32.0) * 1048576.0); }();
  std::unique_ptr<Stream> S(new Stream());
  const int nReadMates = files2.empty() ? 1 : 2;
  S->withBarcode = !barcodeFile.empty();
  S->nMates = nReadMates + (S->withBarcode ? 1 : 0);
  const std::vector<std::string> bcFiles{barcodeFile};
  size_t minRec = ~(size_t)0, estText[3] = {0, 0, 0};
  for (int m = 0; m < S->nMates; ++m) {
    Stream::Mate &M = S->mate[m];
    const bool isBc = m >= nReadMates;
    M.dst = isBc ? &bc : &side[m];
    size_t compressed = 0;
    for (const std::string &path : isBc ? bcFiles : (m ? files2 : files1)) {
      int fd = ::open(path.c_str(), O_RDONLY);
      if (fd < 0) return false;  // (the whole-file path reports it)
      struct stat st;
      if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 64) { ::close(fd); return false; }
      Stream::Src f;
      f.path = path; f.len = (size_t)st.st_size;
      f.map = mmap(nullptr, f.len, PROT_READ, MAP_PRIVATE, fd, 0);
      ::close(fd);
      if (f.map == MAP_FAILED) return false;
      M.srcs.push_back(f);  // (unmapped by the Stream's destructor from here on)
      const uint8_t *z = (const uint8_t *)f.map;
      if (z[0] != 0x1f || z[1] != 0x8b || z[2] != 8) return false;
      if ((z[3] & 4) && f.len > 16 && z[12] == 'B' && z[13] == 'C') return false;  // bgzip: its blocks are inflated side by side (bgzfInflate)
      (void)madvise(f.map, f.len, MADV_SEQUENTIAL);
      compressed += f.len;
      // the text's length: the trailer's length field when it can be the whole file's (one member below 4 GB: the usual case), else 48 x the
      // compressed size as the whole-file path reserves
      const uint32_t isize = (uint32_t)z[f.len - 4] | ((uint32_t)z[f.len - 3] << 8) | ((uint32_t)z[f.len - 2] << 16) | ((uint32_t)z[f.len - 1] << 24);
      const bool plausible = (size_t)isize >= f.len && (size_t)isize <= f.len * 48;
      // The field is the length mod 2^32: a mate with 4.3 - 8.6 GB of text (14 - 27 M reads of 150 bp) leaves a wrapped value that still lies between
      // the file's size and 48 x it (6.4 GB in a 1.4 GB file: 2.1 GB).  FASTQ deflates 3.5 - 5 x, so a value below 3 x the file's size gets 2^32s added
      // until it is not (an estimate too high costs table pages nobody touches twice, one too low a restart of the job: ADVICE round 5).
      size_t est = plausible ? (size_t)isize : f.len * 48;
      if (plausible) while (est < f.len * 3 && est + (1ull << 32) <= f.len * 48) est += 1ull << 32;
      estText[m] += est;
      M.cap += std::max<size_t>(plausible ? est : 0, f.len * 48) + 4096;
    }
    if (!isBc && compressed < minBytes) return false;
    M.path = M.srcs[0].path;
    // a look at the head of the first file's text: four-line FASTQ?  how short can a record be?  (decoded again by the stream: 4 MB are nothing)
    {
      const uint8_t *z = (const uint8_t *)M.srcs[0].map;
      const char *headEnv = getenv("T1K_STREAM_HEAD_MB");
      const size_t headBytes = std::max<size_t>(4096, (size_t)((headEnv ? atof(headEnv) : 4.0) * 1048576.0));  // (tests put odd text right behind a small head)
      std::vector<uint8_t> head(headBytes);
      GzProgress pg;
      std::string e;
      size_t n = 0;
      (void)gzInflateAll(z, M.srcs[0].len, head.data(), head.size(), &pg, &n, nullptr, nullptr, e);  // (ends with "more text than the range holds" for any real file)
      n = (size_t)pg.produced.load();
      const char *p = (const char *)head.data(), *end = p + n;
      if (n < 16 || (*p != '@' && !(isBc && *p == '>'))) return false;   // reads: four-line FASTQ; the barcode file may be two-line FASTA
      M.fastq = *p == '@';
      const int per = M.fastq ? 4 : 2;
      size_t recs = 0;
      while (p < end) {
        // a whole record has all its line ends inside what was decoded
        const char *q = p;
        int lines = 0;
        while (lines < per && q < end) { const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q)); if (!nl) { q = end + 1; break; } q = nl + 1; ++lines; }
        if (lines < per || q > end) break;
        RecFields f;
        if (strictRecordFields(p, q, M.fastq, f) != q) return false;
        if (!isBc) minRec = std::min(minRec, (size_t)(q - p));
        ++recs;
        p = q;
      }
      if (recs < 4) return false;
    }
    M.cap = ((M.cap + ((size_t)64 << 20)) + 4095) & ~(size_t)4095;
  }
  // both files are eligible: room for the text (address space only; the pages that are written get backed)
  for (int m = 0; m < S->nMates; ++m) {
    Stream::Mate &M = S->mate[m];
    void *out = mmap(nullptr, M.cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (out == MAP_FAILED) {
      for (int e = 0; e < m; ++e)
        for (auto it = blobs_.begin(); it != blobs_.end(); ++it) if (it->map == (void *)S->mate[e].text) { munmap(it->map, it->len); blobs_.erase(it); break; }
      return false;
    }
    bigBlockAdvise(out, M.cap);  // huge pages: 3 GB of text are 0.5 s of first touches and 0.25 s of exit in 4 KB pages (t1k_host.h)
    M.text = (char *)out;
    Blob &b = newBlob();
    b.map = out; b.len = M.cap; b.anon = true;
  }
  // tables for twice as many records as the shortest record of the heads would make of the text
  size_t capRecords = 0;
  for (int m = 0; m < nReadMates; ++m) capRecords = std::max(capRecords, estText[m] / std::max<size_t>(8, minRec / 2) + 4096);
  if (capRecords > 0xFFFFFF00ull) return false;  // (fragment numbers are 32 bits)
  S->capRecords = capRecords;
  paired = nReadMates == 2;
  hasBarcode = S->withBarcode;
  for (int m = 0; m < S->nMates; ++m) { Side &d = *S->mate[m].dst; d.seqP.resize(capRecords); d.idP.resize(capRecords); d.seqL.resize(capRecords); d.idL.resize(capRecords); }
  frag.resize(capRecords);
  maxLen = 0;
  streaming = true;
  Stream *sp = S.get();
  for (int m = 0; m < S->nMates; ++m) {
    Stream::Mate *M = &S->mate[m];
    Side *sd = M->dst;
    M->inflater = std::thread([M] {
      size_t at = 0;
      for (size_t i = 0; i < M->srcs.size(); ++i) {
        size_t n = 0;
        const bool last = i + 1 == M->srcs.size();
        if (gzInflateAll((const uint8_t *)M->srcs[i].map, M->srcs[i].len, (uint8_t *)M->text + at, M->cap - at, &M->pg, &n, nullptr, nullptr, M->err, at, last) != 0) {
          M->err = M->srcs[i].path + ": " + M->err;  // (the decoder has set the progress to "failed")
          return;
        }
        at += n;
        if (!last) { std::lock_guard<std::mutex> g(M->m); M->fileEnds.push_back(at); }  // before the next file's first byte is published
      }
      M->textLen = at;
    });
    M->checker = std::thread([M] {
      uint64_t done = 0;
      uint32_t crc = 0;
      size_t member = 0;
      for (;;) {
        const int st = M->pg.state.load(std::memory_order_acquire);
        const uint64_t have = M->pg.produced.load(std::memory_order_acquire);
        uint64_t end = 0;
        uint32_t want = 0;
        bool closes = false;
        { std::lock_guard<std::mutex> g(M->pg.m); if (member < M->pg.members.size()) { end = M->pg.members[member].first; want = M->pg.members[member].second; closes = true; } }
        const uint64_t upto = closes ? end : have;
        if (upto > done) { crc = crcStep(crc, M->text + done, (size_t)(upto - done)); done = upto; M->crcDone.store(done, std::memory_order_release); }
        if (closes) {
          if (crc != want) { M->crcState.store(-1, std::memory_order_release); return; }
          crc = 0; ++member;
          continue;
        }
        if (st != 0 && done >= M->pg.produced.load(std::memory_order_acquire)) {
          bool more;
          { std::lock_guard<std::mutex> g(M->pg.m); more = member < M->pg.members.size(); }
          if (more) continue;
          M->crcState.store(1, std::memory_order_release);  // (a failed decoder is the indexer's to report)
          return;
        }
        if (upto == done) std::this_thread::sleep_for(std::chrono::microseconds(500));
      }
    });
    const bool first = m == 0 && !S->withBarcode;   // (with a barcode file the fragments are numbered by the thread below)
    const bool isRead = m < nReadMates;
    M->indexer = std::thread([this, sp, M, sd, first, isRead] {
      const size_t capR = sp->capRecords;
      uint64_t n = 0;
      size_t recStart = 0, scan = 0;
      int lines = 0, mx = 0;
      auto fail = [&](const std::string &why) { M->errIndex = M->path + ": " + why; M->state.store(-1, std::memory_order_release); };
      auto emit = [&](const char *b, const char *e) -> bool {
        RecFields f;
        const bool laidOut = strictRecordFields(b, e, M->fastq, f) == e;
        if (!laidOut || n >= capR) streamGaveUp.store(true);  // (not damage: the whole-file reader takes such text)
        if (!laidOut) { fail("a record is not in the four-line FASTQ (barcodes: or two-line FASTA) layout the streaming reader follows (T1K_STREAM_GZ=0 reads the file whole)"); return false; }
        if (n >= capR) { fail("more records than the streaming reader sized its tables for (T1K_STREAM_GZ=0 reads the file whole)"); return false; }
        sd->seqP[n] = f.seq; sd->seqL[n] = (uint32_t)f.seqLen; sd->idP[n] = f.id; sd->idL[n] = (uint16_t)f.idLen;
        if (first) frag[n] = (uint32_t)n;
        if (isRead) mx = std::max(mx, (int)f.seqLen);
        ++n;
        return true;
      };
      size_t fileNo = 0;
      // what is left behind the last complete record of a file: blank lines, or a last record whose last line has no line end
      auto closeFile = [&](size_t endAt) -> bool {
        const char *b = M->text + recStart, *e = M->text + endAt;
        while (e > b && (e[-1] == '\n' || e[-1] == '\r')) --e;  // (empty lines only: see addBuffer)
        while (b < e && isBlank(*b)) ++b;
        if (e > b && !emit(b, e)) return false;
        recStart = scan = endAt; lines = 0;
        return true;
      };
      for (;;) {
        const int st = M->pg.state.load(std::memory_order_acquire);
        size_t have = (size_t)M->pg.produced.load(std::memory_order_acquire);
        // (the text of the next file is published only behind this file's end mark: read after `have`, the mark is seen whenever it matters)
        size_t boundary = ~(size_t)0;
        { std::lock_guard<std::mutex> g(M->m); if (fileNo < M->fileEnds.size()) boundary = (size_t)M->fileEnds[fileNo]; }
        if (boundary != ~(size_t)0 && have > boundary) have = boundary;
        const uint64_t before = n;
        while (scan < have) {
          const char *nl = (const char *)memchr(M->text + scan, '\n', have - scan);
          if (!nl) { scan = have; break; }
          scan = (size_t)(nl - M->text) + 1;
          if (++lines == (M->fastq ? 4 : 2)) {
            if (!emit(M->text + recStart, M->text + scan)) return;
            recStart = scan; lines = 0;
            if ((n & 4095) == 0) {
              int old = streamMaxLen.load(std::memory_order_relaxed);
              while (mx > old && !streamMaxLen.compare_exchange_weak(old, mx)) {}
              M->records.store(n, std::memory_order_release);
            }
          }
        }
        if (boundary != ~(size_t)0 && scan >= boundary) {  // this file's text is complete: its tail, then the next file
          if (!closeFile(boundary)) return;
          ++fileNo;
        }
        if (n != before) {
          int old = streamMaxLen.load(std::memory_order_relaxed);
          while (mx > old && !streamMaxLen.compare_exchange_weak(old, mx)) {}
          M->records.store(n, std::memory_order_release);
        }
        if (boundary == ~(size_t)0 && st != 0 && scan >= (size_t)M->pg.produced.load(std::memory_order_acquire)) {
          bool more;
          { std::lock_guard<std::mutex> g(M->m); more = fileNo < M->fileEnds.size(); }
          if (more) continue;  // (a file's end mark arrived between the two looks)
          if (st < 0) { fail("the file is damaged"); return; }
          if (!closeFile(scan)) return;
          int old = streamMaxLen.load(std::memory_order_relaxed);
          while (mx > old && !streamMaxLen.compare_exchange_weak(old, mx)) {}
          M->records.store(n, std::memory_order_release);
          M->state.store(1, std::memory_order_release);
          return;
        }
        if (n == before && !(boundary != ~(size_t)0 && scan >= boundary)) std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    });
  }
  if (sp->withBarcode)
    sp->fragger = std::thread([this, sp] {
      uint64_t done = 0, nf = 0;
      for (;;) {
        int st = 1;
        uint64_t have = ~0ull;
        for (int m = 0; m < sp->nMates; ++m) {
          const int a = sp->mate[m].state.load(std::memory_order_acquire);  // (before the count: a finished indexer's count is final)
          have = std::min<uint64_t>(have, sp->mate[m].records.load(std::memory_order_acquire));
          if (a < 0) { sp->fragState.store(-1, std::memory_order_release); return; }
          if (a == 0) st = 0;
        }
        const uint64_t before = nf;
        for (; done < have; ++done)
          if (!(bc.seqL[done] == 15 && !memcmp(bc.seqP[done], "missing_barcode", 15))) frag[nf++] = (uint32_t)done;
        if (nf != before) sp->fragments.store(nf, std::memory_order_release);
        if (st == 1) {
          uint64_t all = ~0ull;
          for (int m = 0; m < sp->nMates; ++m) all = std::min<uint64_t>(all, sp->mate[m].records.load(std::memory_order_acquire));
          if (done >= all) { sp->fragState.store(1, std::memory_order_release); return; }
          continue;
        }
        if (nf == before) std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    });
  stream_ = std::move(S);
  return true;
}

bool ReadInput::streamFinish(std::string &err) {
  if (!stream_) return true;
  Stream &S = *stream_;
  for (int m = 0; m < S.nMates; ++m) {
    Stream::Mate &M = S.mate[m];
    if (M.inflater.joinable()) M.inflater.join();
    if (M.indexer.joinable()) M.indexer.join();
    if (M.checker.joinable()) M.checker.join();
  }
  streaming = false;
  for (int m = 0; m < S.nMates; ++m) {
    Stream::Mate &M = S.mate[m];
    if (M.state.load() < 0 || M.pg.state.load() < 0) {  // (every thread has been joined: the strings are at rest; the decoder's message says more than "damaged")
      err = !M.err.empty() ? M.err : !M.errIndex.empty() ? M.errIndex : M.path + ": cannot read the file";
      return false;
    }
    if (M.crcState.load() < 0) { err = M.path + ": the file is damaged (CRC check of the inflated text failed)"; return false; }
  }
  if (S.fragger.joinable()) S.fragger.join();
  const size_t n = (size_t)S.mate[0].records.load();
  const int nReadMates = S.nMates - (S.withBarcode ? 1 : 0);
  if (nReadMates == 2 && (size_t)S.mate[1].records.load() != n) { err = "mate files hold different numbers of reads"; return false; }
  if (S.withBarcode && (size_t)S.mate[S.nMates - 1].records.load() != n) { err = "barcode file and read file hold different numbers of records"; return false; }
  for (int m = 0; m < S.nMates; ++m) { Side &d = *S.mate[m].dst; d.seqP.resize(n); d.idP.resize(n); d.seqL.resize(n); d.idL.resize(n); }
  frag.resize(S.withBarcode ? (size_t)S.fragments.load() : n);
  maxLen = streamMaxLen.load();
  // the unused tail of the text reservations goes back (the blobs keep what holds text)
  for (int m = 0; m < S.nMates; ++m) {
    Stream::Mate &M = S.mate[m];
    const size_t keep = std::max<size_t>(4096, (M.textLen + 4095) & ~(size_t)4095);
    for (Blob &b : blobs_)
      if (b.map == (void *)M.text && keep < b.len) { munmap((char *)b.map + keep, b.len - keep); b.len = keep; }
  }
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k job] gzip read files streamed: %zu records per mate, %zu + %zu bytes of text%s, tables sized for %zu records\n", n, S.mate[0].textLen, nReadMates == 2 ? S.mate[1].textLen : (size_t)0,
            S.withBarcode ? " + the barcode file" : "", S.capRecords);
  return true;
}

bool ReadInput::open(const std::vector<std::string> &files1, const std::vector<std::string> &files2, const std::string &barcodeFile, int threads, std::string &err) {
  paired = !files2.empty();
  hasBarcode = !barcodeFile.empty();
  // the mates (and the barcode file) are independent: read them side by side
  std::string e1, e2, e3;
  bool ok1 = true, ok2 = true, ok3 = true;
  const int per = std::max(1, threads / (1 + (paired ? 1 : 0)));
  std::thread t2, t3;
  if (paired) t2 = std::thread([&] { for (auto &f : files2) if (!(ok2 = addFile(f, per, side[1], e2))) break; });
  if (hasBarcode) t3 = std::thread([&] { ok3 = addFile(barcodeFile, std::max(1, per / 2), bc, e3); });
  for (auto &f : files1) if (!(ok1 = addFile(f, per, side[0], e1))) break;
  if (t2.joinable()) t2.join();
  if (t3.joinable()) t3.join();
  if (!ok1) { err = e1; return false; }
  if (!ok2) { err = e2; return false; }
  if (!ok3) { err = e3; return false; }
  if (paired && side[1].seqP.size() != side[0].seqP.size()) { err = "mate files hold different numbers of reads"; return false; }
  if (hasBarcode && bc.seqP.size() != side[0].seqP.size()) { err = "barcode file and read file hold different numbers of records"; return false; }
  finish();
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// One process per GPU: a rank indexes only its own fragments.
//
// The fragments of a sharded job are contiguous slices of the input (rank r: [F r / N, F (r + 1) / N)), so a rank has to find
// the bytes of records it has never seen.  In the strict layouts a record is a fixed number of lines (4: FASTQ, 2: FASTA), so
// counting newlines is enough: the text of all files is cut into 1 MiB blocks, every rank counts the newlines of its share of
// the blocks (1 / N of the bytes), the counts are all-gathered (4 bytes per MiB of input), and every rank then knows how many
// records each file holds and in which block any record starts; it scans that one block for the exact byte and indexes its own
// byte range with the same in-place indexer as the single-process path (which also verifies the layout record by record).
// Anything else -- gz, a pipe, a barcode file, a layout that is not strict on ANY rank -- returns 0 on every rank and the caller
// indexes everything as before.
// ------------------------------------------------------------------------------------------------------------------
int ReadInput::openSharded(const std::vector<std::string> &files1, const std::vector<std::string> &files2, int threads, const ShardComm &c, std::string &err) {
  constexpr size_t BLK = 1u << 20;
  struct View { const char *b = nullptr, *end = nullptr; bool fastq = false; size_t nBlocks = 0, firstBlock = 0; uint64_t lines = 0, records = 0; int L = 4; };
  paired = !files2.empty();
  hasBarcode = false;
  std::vector<View> views[2];
  size_t NB = 0;
  for (int m = 0; m < (paired ? 2 : 1); ++m) {
    for (const std::string &path : m ? files2 : files1) {
      int fd = ::open(path.c_str(), O_RDONLY);
      if (fd < 0) { err = "cannot open " + path; return -1; }
      struct stat st;
      if (fstat(fd, &st) != 0) { ::close(fd); err = "cannot stat " + path; return -1; }
      unsigned char magic[2] = {0, 0};
      const bool gz = st.st_size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
      if (gz || !S_ISREG(st.st_mode)) { ::close(fd); return 0; }
      View v;
      if (st.st_size > 0) {
        void *mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (mp == MAP_FAILED) { ::close(fd); err = "cannot map " + path; return -1; }
        Blob &b = newBlob();
        b.map = mp; b.len = (size_t)st.st_size;
        v.b = (const char *)mp; v.end = v.b + st.st_size;
        while (v.end > v.b && (v.end[-1] == '\n' || v.end[-1] == '\r')) --v.end;  // (empty lines only: see addBuffer)
        while (v.b < v.end && isBlank(*v.b)) ++v.b;
        if (v.b < v.end) {
          v.fastq = *v.b == '@';
          if (!v.fastq && *v.b != '>') { ::close(fd); return 0; }  // the whole-file path reports it
          v.L = v.fastq ? 4 : 2;
          v.nBlocks = ((size_t)(v.end - v.b) + BLK - 1) / BLK;
        }
      }
      ::close(fd);
      v.firstBlock = NB;
      NB += v.nBlocks;
      views[m].push_back(v);
    }
  }
  // newline counts per block: this rank's share, then everybody's
  std::vector<uint32_t> nl(NB + 1, 0);
  const int N = c.nRanks;
  const size_t k0 = NB * (size_t)c.rank / N, k1 = NB * ((size_t)c.rank + 1) / N;
  auto blockText = [&](size_t k, const char *&p, const char *&e) {
    for (int m = 0; m < 2; ++m)
      for (const View &v : views[m])
        if (k >= v.firstBlock && k < v.firstBlock + v.nBlocks) {
          p = v.b + (k - v.firstBlock) * BLK;
          e = std::min(v.end, p + BLK);
          return;
        }
    p = e = nullptr;
  };
  {
    std::atomic<size_t> next{k0};
    auto work = [&] {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= k1) return;
        const char *p, *e;
        blockText(k, p, e);
        uint32_t n = 0;
        while (p < e) {
          const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
          if (!q) break;
          ++n; p = q + 1;
        }
        nl[k] = n;
      }
    };
    std::vector<std::thread> th;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, k1 - k0));
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
  }
  {
    std::vector<uint64_t> bytes(N), displ(N);
    for (int r = 0; r < N; ++r) { displ[r] = 4 * (NB * (size_t)r / N); bytes[r] = 4 * (NB * ((size_t)r + 1) / N) - displ[r]; }
    if (NB && !c.allgatherv(nl.data(), bytes.data(), displ.data(), 4 * NB)) { err = "read input: exchange of the line counts failed"; return -1; }
  }
  uint64_t total[2] = {0, 0};
  for (int m = 0; m < (paired ? 2 : 1); ++m)
    for (View &v : views[m]) {
      if (v.b >= v.end) continue;
      uint64_t n = 0;
      for (size_t k = 0; k < v.nBlocks; ++k) n += nl[v.firstBlock + k];
      v.lines = n + 1;  // the trimmed text does not end in a newline
      if (v.lines % v.L) return 0;  // not the strict layout (every rank sees the same table)
      v.records = v.lines / v.L;
      total[m] += v.records;
    }
  if (paired && total[0] != total[1]) { err = "mate files hold different numbers of reads"; return -1; }
  if (total[0] > 0xFFFFFFF0ull) { err = "too many fragments"; return -1; }
  const uint64_t Fall = total[0];
  const uint64_t fBeg = Fall * (uint64_t)c.rank / N, fEnd = Fall * ((uint64_t)c.rank + 1) / N;
  // byte of the first character of line `line` (0-based) of a file
  auto lineStart = [&](const View &v, uint64_t line) -> const char * {
    if (line == 0) return v.b;
    uint64_t before = 0;  // newlines in the blocks before k
    size_t k = 0;
    while (k < v.nBlocks && before + nl[v.firstBlock + k] < line) { before += nl[v.firstBlock + k]; ++k; }
    if (k >= v.nBlocks) return v.end;
    const char *p = v.b + k * BLK, *e = std::min(v.end, p + BLK);
    for (uint64_t need = line - before; need > 0; --need) {
      const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
      if (!q) return v.end;  // cannot happen: the block holds that many newlines
      p = q + 1;
    }
    return p;
  };
  bool okSide[2] = {true, true};
  auto indexSide = [&](int m, int T) {
    uint64_t R = 0;
    for (const View &v : views[m]) {
      const uint64_t lo = std::max(fBeg, R), hi = std::min(fEnd, R + v.records);
      if (lo < hi) {
        const char *p0 = lineStart(v, (lo - R) * v.L);
        const char *p1 = hi - R == v.records ? v.end : lineStart(v, (hi - R) * v.L);
        const size_t before = side[m].seqP.size();
        if (!addRange(p0, p1, v.end, v.fastq, T, side[m]) || side[m].seqP.size() - before != hi - lo) okSide[m] = false;
      }
      R += v.records;
    }
  };
  {
    const int per = std::max(1, threads / (paired ? 2 : 1));
    std::thread t2;
    if (paired) t2 = std::thread([&] { indexSide(1, per); });
    indexSide(0, per);
    if (t2.joinable()) t2.join();
  }
  // agreement: the layout is strict everywhere (else every rank falls back); the longest read of the whole input (Genotyper.cpp:443)
  uint64_t localMax = 0;
  for (int m = 0; m < (paired ? 2 : 1); ++m)
    for (uint32_t l : side[m].seqL) localMax = std::max<uint64_t>(localMax, l);
  {
    std::vector<uint64_t> flags(2 * (size_t)N, 0), bytes(N, 16), displ(N);
    for (int r = 0; r < N; ++r) displ[r] = 16 * (uint64_t)r;
    flags[2 * c.rank] = okSide[0] && okSide[1] ? 1 : 0;
    flags[2 * c.rank + 1] = localMax;
    if (!c.allgatherv(flags.data(), bytes.data(), displ.data(), 16 * (uint64_t)N)) { err = "read input: exchange of the layout flags failed"; return -1; }
    maxLen = 0;
    for (int r = 0; r < N; ++r) {
      if (!flags[2 * r]) return 0;
      maxLen = std::max<int>(maxLen, (int)flags[2 * r + 1]);
    }
  }
  const size_t n = side[0].seqP.size();
  frag.resize(n);
  for (size_t i = 0; i < n; ++i) frag[i] = (uint32_t)i;
  sharded = true; base = (uint32_t)fBeg; nAll_ = (uint32_t)Fall; shardRank = c.rank; shardRanks = N;
  return 1;
}

void ReadInput::release(size_t recLo, size_t recHi) {
  if (recLo >= recHi) return;
  const long page = sysconf(_SC_PAGESIZE);
  Side *sides[3] = {&side[0], paired ? &side[1] : nullptr, hasBarcode ? &bc : nullptr};
  for (Side *sd : sides) {
    if (!sd || recHi > sd->idP.size() || !sd->idP[recLo]) continue;
    auto blobOf = [&](const char *p) -> const Blob * {
      for (const Blob &b : blobs_)
        if (b.map && p >= (const char *)b.map && p < (const char *)b.map + b.len) return &b;
      return nullptr;
    };
    const char *first = sd->idP[recLo], *last = sd->idP[recHi - 1];
    const Blob *b0 = blobOf(first), *b1 = blobOf(last);
    if (!b0 || !b1) continue;
    const bool anon = b0->anon || b1->anon;
    if (anon && !dropInflatedText) continue;  // owned storage (gz, general reader): a rerun of the job reads the text again, and dropped anonymous pages come back as zeros
    // (inflated text sits in huge pages: whole 2 MB units only, or every call would split two of them)
    const uintptr_t unit = anon ? ((uintptr_t)2 << 20) : (uintptr_t)page;
    auto drop = [&](const char *a, const char *b) {  // whole units inside [a, b)
      uintptr_t lo = ((uintptr_t)a + unit - 1) & ~(unit - 1), hi = (uintptr_t)b & ~(unit - 1);
      if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_DONTNEED);
    };
    // (a stream's tables are sized to a bound: entries beyond what has been published are not records yet, and the text range has room behind
    // the text that exists -- without a next record the range ends at the last record's start)
    const size_t known = streaming ? std::min(sd->idP.size(), streamAvail()) : sd->idP.size();
    const char *end1 = recHi < known && blobOf(sd->idP[recHi]) == b1 ? sd->idP[recHi] - 1 : (anon ? last : (const char *)b1->map + b1->len);
    if (stream_ && anon) {  // ... and the CRC checker must have been over it (it runs far ahead of the loop; with zlib's routine it may not)
      int mate = 0;
      for (int m = 0; m < stream_->nMates; ++m) if (stream_->mate[m].dst == sd) mate = m;
      Stream::Mate &SM = stream_->mate[mate];
      const char *checked = SM.text + SM.crcDone.load(std::memory_order_acquire);
      if (end1 > checked) end1 = checked;
      // ... and the DECODER must be done with it: a match copies from up to 32 KB behind its write position, inside the member it is decoding.  A
      // writer that has caught up with a stalled decoder would drop a unit under that window, the decoder would copy zeros into later text and the
      // run would end with a CRC error nobody can reproduce.  The write position is at or beyond what is published: while the
      // decoder runs (state read first), nothing within 32 KB of the published length is dropped.
      if (SM.pg.state.load(std::memory_order_acquire) == 0) {
        const uint64_t produced = SM.pg.produced.load(std::memory_order_acquire);
        const char *safe = SM.text + (produced > 32768 ? produced - 32768 : 0);
        if (end1 > safe) end1 = safe;
      }
      if (end1 <= first) continue;
    }
    if (b0 == b1) drop(first - 1, end1);
    else { drop(first - 1, (const char *)b0->map + b0->len); drop((const char *)b1->map, end1); }  // (files wholly inside the range wait for the unmapping)
  }
}

void ReadInput::setMemory(const char *seq1, const uint64_t *off1, const char *seq2, const uint64_t *off2, uint32_t n) {
  paired = seq2 != nullptr;
  hasBarcode = false;
  noIds = true;
  for (int m = 0; m < (paired ? 2 : 1); ++m) {
    const char *s = m ? seq2 : seq1;
    const uint64_t *o = m ? off2 : off1;
    Blob &b = newBlob();
    b.owned.reset(new std::vector<char>(s + o[0], s + o[n]));
    const char *base = b.owned->data();
    Side &d = side[m];
    d.seqP.resize(n); d.seqL.resize(n); d.idP.assign(n, nullptr); d.idL.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) { d.seqP[i] = base + (o[i] - o[0]); d.seqL[i] = (uint32_t)(o[i + 1] - o[i]); }
  }
  finish();
}

// fragments = records whose barcode is not "missing_barcode" (dropped with their mates, Genotyper.cpp:376-381)
void ReadInput::finish() {
  const size_t n = side[0].seqP.size();
  const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency())), n / 65536 + 1));
  auto pieces = [&](auto fn) {  // fn(t, begin, end) over contiguous pieces of [0, n)
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { fn(t, n * t / T, n * (t + 1) / T); });
    fn(0u, (size_t)0, n / T);
    for (auto &x : th) x.join();
  };
  frag.clear();
  if (!hasBarcode) {
    frag.resize(n);
    pieces([&](unsigned, size_t b, size_t e) { for (size_t i = b; i < e; ++i) frag[i] = (uint32_t)i; });
  } else {
    frag.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      if (bc.seqL[i] == 15 && !memcmp(bc.seqP[i], "missing_barcode", 15)) continue;
      frag.push_back((uint32_t)i);
    }
  }
  // the longest read (of the fragments that are kept)
  std::vector<int> mx(T, 0);
  const size_t nf = frag.size();
  {
    std::vector<std::thread> th;
    auto work = [&](unsigned t) {
      int m = 0;
      for (int s2 = 0; s2 < (paired ? 2 : 1); ++s2)
        for (size_t i = nf * t / T; i < nf * (t + 1) / T; ++i) m = std::max<int>(m, (int)side[s2].seqL[frag[i]]);
      mx[t] = m;
    };
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  maxLen = 0;
  for (int m : mx) maxLen = std::max(maxLen, m);
}

}  // namespace t1k
