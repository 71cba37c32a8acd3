// t1k_amd/csrc/host/reads.cpp -- the read files of a job, whole in memory with a record index.
//
// The reference streams records through kseq one at a time on one thread (ReadFiles::Next, ReadFiles.hpp:155-204) and strdup()s
// every id and sequence (Genotyper.cpp:365-440).  Here a plain file is mmap()ed (a gz file is inflated once into memory), cut into
// byte ranges, and every host thread indexes its range in place: a record is two pointers and two lengths into the mapping, nothing
// is copied, and ids / sequences / barcodes are read from the mapping again when the *_aligned*.fa files are written.  The in-place
// indexer accepts the two layouts the pipeline produces (four-line FASTQ, two-line FASTA; LF or CRLF); anything else -- wrapped
// sequences, blank lines -- goes through the general record reader (readSeqFile, the kseq rules) into owned storage.
// Several files per mate are read back to back, as ReadFiles does with currentFpInd.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
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
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n / (1u << 20) + 1));
  std::vector<const char *> start(T + 1, end);
  start[0] = b;
  std::vector<Piece> piece(T);
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { start[t] = findRecord(p, b + (size_t)(end - b) / T * t, end, fastq); });
    for (auto &x : th) x.join();
    for (int t = 1; t <= T; ++t) start[t] = std::max(start[t], start[t - 1]);
    start[T] = end;
  }
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        Piece &pc = piece[t];
        const char *q = start[t], *stop = start[t + 1];
        const size_t guess = (size_t)(stop - q) / 200 + 16;
        pc.seqP.reserve(guess); pc.seqL.reserve(guess); pc.idP.reserve(guess); pc.idL.reserve(guess);
        while (q < stop) {
          q = strictRecord(q, end, fastq, pc);
          if (!q) { pc.ok = false; return; }
        }
        if (q != stop) pc.ok = false;
      });
    for (auto &x : th) x.join();
  }
  bool strict = true;
  for (auto &pc : piece) strict = strict && pc.ok;
  if (!strict) return false;  // caller falls back to the general reader (err left empty)
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
  frag.clear();
  frag.reserve(n);
  maxLen = 0;
  for (size_t i = 0; i < n; ++i) {
    if (hasBarcode && bc.seqL[i] == 15 && !memcmp(bc.seqP[i], "missing_barcode", 15)) continue;
    frag.push_back((uint32_t)i);
  }
  for (int m = 0; m < (paired ? 2 : 1); ++m)
    for (uint32_t i : frag) maxLen = std::max<int>(maxLen, (int)side[m].seqL[i]);
}

}  // namespace t1k
