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
inline const char *strictRecord(const char *p, const char *end, bool fastq, Piece &out) {
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
    if (ql && q[ql - 1] == '\r') --ql;
    if (ql != sl) return nullptr;
    next = e4 < end ? e4 + 1 : end;
  } else {
    if (next < end && *next != '>') return nullptr;  // wrapped sequence or a blank line: not the strict layout
  }
  out.seqP.push_back(s); out.seqL.push_back((uint32_t)sl); out.idP.push_back(id); out.idL.push_back((uint16_t)il);
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

ReadInput::~ReadInput() {
  for (auto &b : blobs_)
    if (b.map) munmap(b.map, b.len);
}

bool ReadInput::addBuffer(const char *p, size_t n, int threads, Side &dst, std::string &err, const std::string &what) {
  const char *end = p + n;
  while (n && isBlank(end[-1])) { --end; --n; }  // trailing blank lines
  const char *b = p;
  while (b < end && isBlank(*b)) ++b;
  if (b >= end) return true;  // empty file: no records
  const bool fastq = *b == '@';
  if (!fastq && *b != '>') { err = what + ": neither FASTA nor FASTQ"; return false; }
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
        while (v.end > v.b && isBlank(v.end[-1])) --v.end;
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
    auto drop = [&](const char *a, const char *b) {  // whole pages inside [a, b)
      uintptr_t lo = ((uintptr_t)a + (uintptr_t)page - 1) & ~((uintptr_t)page - 1), hi = (uintptr_t)b & ~((uintptr_t)page - 1);
      if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_DONTNEED);
    };
    const char *first = sd->idP[recLo], *last = sd->idP[recHi - 1];
    const Blob *b0 = blobOf(first), *b1 = blobOf(last);
    if (!b0 || !b1 || b0->anon || b1->anon) continue;  // owned storage (gz, general reader): a rerun of the job reads the text again, and dropped anonymous pages come back as zeros
    const char *end1 = recHi < sd->idP.size() && blobOf(sd->idP[recHi]) == b1 ? sd->idP[recHi] - 1 : (const char *)b1->map + b1->len;
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
