// t1k_amd/csrc/host/refset.cpp -- sequence-file input and the allele reference set of the genotyper stage.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstring>
#include <thread>
#include "t1k_host.h"

namespace t1k {

namespace {
// A record reader with the rules of the reader the reference vendors (kseq.h:185-224, driven by ReadFiles::Next, ReadFiles.hpp:155-204),
// restated as a state machine over a byte source: what it hands out must equal the reference's reader byte for byte on ANY file -- odd
// but legal ones (wrapped sequences, blank lines, CRLF, comments, records of both kinds in one file) and damaged ones, where the rules
// have consequences a line-by-line reading misses:
//   * after a FASTQ record the next record starts at the next '@' or '>' WHEREVER it stands (not only at a line start);
//   * quality lines are joined until they are at least as long as the sequence; a different length ends the FILE (kseq_read returns -2,
//     ReadFiles::Next goes on with the next file), and so does a '+' line with nothing behind it;
//   * a trailing CR is dropped from a line only when what has been gathered so far is longer than one character (kseq.h:142);
//   * the name ends at the first isspace() character, the comment is the rest of that line behind this one character.
// The source is filled 16 384 bytes at a time through zlib, as the reference's stream is (kseq.h:234; plain files are read as they are):
// an end of file on a buffer boundary is seen one call later there, and here.
class RecordSource {
 public:
  explicit RecordSource(const std::string &path) : buf_(kBuf) { fp_ = gzopen(path.c_str(), "rb"); }
  ~RecordSource() { if (fp_) gzclose(fp_); }
  bool ok() const { return fp_ != nullptr; }
  // >= 0: a record (its sequence length); -1: end of the file; -2: a quality string of another length, or none -- the reader of the
  // reference treats both as the end of this file
  long next(std::string &name, std::string &comment, std::string &seq, std::string &qual) {
    int c;
    if (last_ == 0) {  // to the next header character
      while ((c = getc()) != -1 && c != '>' && c != '@') {}
      if (c == -1) return -1;
      last_ = c;
    }
    comment.clear(); seq.clear(); qual.clear();
    if (until(false, name, &c, false) < 0) return -1;
    if (c != '\n') until(true, comment, nullptr, false);
    while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
      if (c == '\n') continue;  // empty lines
      seq.push_back((char)c);
      until(true, seq, nullptr, true);
    }
    if (c == '>' || c == '@') last_ = c;  // the next record's first character is taken
    if (c != '+') return (long)seq.size();
    while ((c = getc()) != -1 && c != '\n') {}
    if (c == -1) return -2;
    while (until(true, qual, nullptr, true) >= 0 && qual.size() < seq.size()) {}
    last_ = 0;
    if (qual.size() != seq.size()) return -2;
    return (long)seq.size();
  }

 private:
  static constexpr size_t kBuf = 16384;
  void fill() {
    const int n = gzread(fp_, buf_.data(), (unsigned)kBuf);
    begin_ = 0; end_ = n > 0 ? (size_t)n : 0;
    if (end_ < kBuf) eof_ = true;
  }
  int getc() {
    if (begin_ >= end_) {
      if (eof_) return -1;
      fill();
      if (end_ == 0) return -1;
    }
    return (unsigned char)buf_[begin_++];
  }
  // gathers up to the end of the line (line) or the first isspace() character; the delimiter is consumed and reported in *delim (0: none)
  long until(bool line, std::string &str, int *delim, bool append) {
    if (delim) *delim = 0;
    if (!append) str.clear();
    if (begin_ >= end_ && eof_) return -1;
    for (;;) {
      if (begin_ >= end_) {
        if (eof_) break;
        fill();
        if (end_ == 0) break;
      }
      size_t i = begin_;
      if (line) { const char *nl = (const char *)memchr(buf_.data() + begin_, '\n', end_ - begin_); i = nl ? (size_t)(nl - buf_.data()) : end_; }
      else while (i < end_ && !isspace((unsigned char)buf_[i])) ++i;
      str.append(buf_.data() + begin_, i - begin_);
      begin_ = i + 1;
      if (i < end_) { if (delim) *delim = (unsigned char)buf_[i]; break; }
    }
    if (line && str.size() > 1 && str.back() == '\r') str.pop_back();
    return (long)str.size();
  }
  gzFile fp_ = nullptr;
  std::vector<char> buf_;
  size_t begin_ = 0, end_ = 0;
  bool eof_ = false;
  int last_ = 0;
};
}  // namespace

bool readSeqFile(const std::string &path, std::vector<SeqRec> &out, std::string &err) {
  RecordSource in(path);
  if (!in.ok()) { err = "cannot open " + path; return false; }
  std::string name, comment, seq, qual;
  while (in.next(name, comment, seq, qual) >= 0) {  // ReadFiles::Next: any negative return ends the file
    SeqRec r;
    // the reference copies name, sequence and comment as C strings (strdup, ReadFiles.hpp:183-198): a NUL byte ends them
    r.id.assign(name.c_str());
    size_t n = r.id.size();
    if (n >= 2 && r.id[n - 2] == '/' && (r.id[n - 1] == '1' || r.id[n - 1] == '2')) r.id.resize(n - 2);  // ReadFiles.hpp:185-189
    r.seq.assign(seq.c_str());
    r.hasComment = !comment.empty();
    if (r.hasComment) r.comment.assign(comment.c_str());
    out.push_back(std::move(r));
  }
  return true;
}

namespace {
// The reference FASTA (35 MB for an HLA set) read by all host threads: the file is mapped, cut at record starts, and every thread
// parses its records with the rules of readSeqFile (name up to the first blank, "/1" "/2" stripped, the rest of the header line is
// the comment, sequence = the following lines joined, CR dropped).  Anything that is not plain '>' records -- gz, a line starting
// with '@' or '+' -- returns false and the caller takes the general reader.
bool readFastaParallel(const std::string &path, std::vector<SeqRec> &out) {
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  unsigned char magic[2] = {0, 0};
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2 || pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) { ::close(fd); return false; }
  const size_t n = (size_t)st.st_size;
  void *mp = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (mp == MAP_FAILED) return false;
  const char *d = (const char *)mp, *end = d + n;
  // two things only the general reader restates: a NUL byte (the reference copies every field as a C string) and a header character as
  // the file's last byte (kseq.h:195: no record -- unless the stream's buffer ends there too)
  // (the NUL test runs on the threads below, each over its own piece)
  if (d[n - 1] == '>' && (n == 1 || d[n - 2] == '\n')) { munmap(mp, n); return false; }
  auto recordStart = [&](const char *from) -> const char * {  // first '>' at a line start at or after `from`
    const char *p = from;
    if (p == d && *p == '>') return p;
    while (p < end) {
      const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
      if (!nl || nl + 1 >= end) return end;
      if (nl[1] == '>') return nl + 1;
      p = nl + 1;
    }
    return end;
  };
  const int T = (int)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), n / (1u << 20) + 1));
  std::vector<const char *> start(T + 1, end);
  start[0] = recordStart(d);
  for (int t = 1; t < T; ++t) start[t] = recordStart(d + n / T * t);
  for (int t = 1; t <= T; ++t) start[t] = std::max(start[t], start[t - 1]);
  start[T] = end;
  // lines before the first record must be blank or carry no header (the general reader skips them); '@' / '+' lines: not our case
  bool plain = true;
  for (const char *p = d; p < start[0];) {
    const char *nl = (const char *)memchr(p, '\n', (size_t)(start[0] - p));
    if (*p == '@' || *p == '+') plain = false;
    if (!nl) break;
    p = nl + 1;
  }
  std::vector<std::vector<SeqRec>> part(T);
  std::atomic<bool> ok{plain};
  auto work = [&](int t) {
    const char *p = start[t], *stop = start[t + 1];
    if (memchr(t == 0 ? d : p, 0, (size_t)(stop - (t == 0 ? d : p)))) { ok = false; return; }
    std::vector<SeqRec> &recs = part[t];
    while (p < stop && ok) {
      // header line
      const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
      const char *le = nl ? nl : end;
      SeqRec r;
      const char *q = p + 1;
      while (q < le && !isspace((unsigned char)*q)) ++q;  // the name ends at the first isspace() character (a CR included), kseq.h:195
      r.id.assign(p + 1, q);
      if (q < le) {  // the comment is the rest of the line behind that one character; a CR goes from more than one character (kseq.h:142, 196)
        r.comment.assign(q + 1, le);
        if (r.comment.size() > 1 && r.comment.back() == '\r') r.comment.pop_back();
        r.hasComment = !r.comment.empty();
      }
      const size_t il = r.id.size();
      if (il >= 2 && r.id[il - 2] == '/' && (r.id[il - 1] == '1' || r.id[il - 1] == '2')) r.id.resize(il - 2);
      p = nl ? nl + 1 : end;
      // sequence lines up to the next record
      while (p < end && *p != '>') {
        if (*p == '@' || *p == '+') { ok = false; return; }
        const char *nl2 = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *e2 = nl2 ? nl2 : end;
        const char *se = e2;
        if (se == p + 1 && *p == '\r' && r.seq.empty()) { ok = false; return; }  // a lone CR as a sequence's first character is kept by the reference's reader (kseq.h:142): the general reader's case
        if (se > p && se[-1] == '\r') --se;
        r.seq.append(p, se);
        p = nl2 ? nl2 + 1 : end;
      }
      recs.push_back(std::move(r));
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  munmap(mp, n);
  if (!ok) return false;
  size_t total = 0;
  for (auto &v : part) total += v.size();
  out.reserve(out.size() + total);
  for (auto &v : part)
    for (auto &r : v) out.push_back(std::move(r));
  return true;
}
}  // namespace

// the records of the allele reference: plain '>' records by all host threads, anything else through the general reader
bool readReferenceRecords(const std::string &path, std::vector<SeqRec> &out, std::string &err) {
  if (readFastaParallel(path, out)) return true;
  out.clear();
  return readSeqFile(path, out, err);
}

// Genotyper::ParseAlleleName (Genotyper.hpp:63-131)
void RefSet::splitName(const std::string &allele, std::string &gene, std::string &major, int fieldsType) const {
  int mode = 1, fields = digitUnits;
  char delim = 0;
  if (fields == -1) {
    fields = 3;
    if (allele.find(':') != std::string::npos) { delim = ':'; mode = 2; }
    if (fieldsType >= 1) fields = mode == 1 ? 5 : 3;
  }
  if (delimiter != 0) { delim = delimiter; mode = 2; }
  size_t star = allele.find('*');
  size_t i = star == std::string::npos ? allele.size() : star;
  gene = allele.substr(0, i);
  if (mode == 1) {
    size_t j = 0;
    while ((int)j <= fields && i + j < allele.size()) ++j;
    major = allele.substr(0, i + j);
  } else {
    int seen = 0;
    size_t j = i;
    for (; j < allele.size(); ++j)
      if (allele[j] == delim && ++seen >= fields) break;
    major = allele.substr(0, j);
  }
}

namespace {
// canonical 31-mer multiset of one sequence (KmerCount::AddCount, KmerCount.hpp:53-81); which of the two strands
// represents a pair is irrelevant for the similarity below, only that both map to one key
void kmerProfile(const std::string &s, std::unordered_map<uint64_t, int> &prof) {
  const int K = 31;
  if ((int)s.size() < K) return;
  const uint64_t mask = (1ull << (2 * K)) - 1;
  uint64_t fw = 0, rv = 0;
  int invalid = -1;
  for (size_t i = 0; i < s.size(); ++i) {
    int c = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : s[i] == 'T' ? 3 : -1;
    if (invalid != -1) ++invalid;
    int b = c < 0 ? 3 : c;
    fw = ((fw << 2) | (uint64_t)b) & mask;
    rv = (rv >> 2) | ((uint64_t)(3 - b) << (2 * (K - 1)));
    if (s[i] == 'N') invalid = 0;
    if (invalid >= K) invalid = -1;
    if ((int)i < K - 1) continue;
    if (invalid == -1) ++prof[fw < rv ? fw : rv];
  }
}
// fn(g) for every gene g on a bounded pool (a custom reference may split into thousands of "genes": one thread each could hit the
// process thread limit, and the resulting std::system_error would end the process)
template <class F>
void overGenes(int Gn, F fn) {
  const int T = std::max(1, std::min({16, (int)std::thread::hardware_concurrency(), Gn}));
  std::atomic<int> next{0};
  auto work = [&] { for (int g = next.fetch_add(1); g < Gn; g = next.fetch_add(1)) fn(g); };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work);
  work();
  for (auto &x : th) x.join();
}
}  // namespace

bool RefSet::load(const std::string &fasta, int digitUnitsArg, char delimiterArg, std::string &err, const std::set<std::string> *selected) {
  digitUnits = digitUnitsArg;
  delimiter = delimiterArg;
  std::vector<SeqRec> recs;
  const auto tA = std::chrono::steady_clock::now();
  if (!readReferenceRecords(fasta, recs, err)) return false;
  const auto tB = std::chrono::steady_clock::now();
  // per record, on the host threads: hash of the sequence, effective length, exon mask
  struct Pre { uint64_t hash = 0; int effLen = 0; bool gap = false; std::vector<uint8_t> mask; };
  std::vector<Pre> pre(recs.size());
  {
    const size_t R = recs.size();
    const int T = (int)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), R / 256 + 1));
    std::vector<std::thread> th;
    auto work = [&](int t) {
      for (size_t i = R * t / T; i < R * (t + 1) / T; ++i) {
        const SeqRec &r = recs[i];
        if (selected && !selected->count(r.id)) continue;  // Genotyper.hpp:742-743
        Pre &q = pre[i];
        q.hash = std::hash<std::string>()(r.seq);
        const int L = (int)r.seq.size();
        {
          // SeqSet::ComputeEffectiveLen (747-758).  (Counted in a local: through q.effLen every base cost a load-add-store of the counter,
          // since a char access may alias it -- 7 ns a base, a third of the reference's load time)
          const char *sp = r.seq.data();
          int eff = 0;
          for (int k = 0; k < L; ++k) eff += (sp[k] != 'N' || (k > 0 && sp[k - 1] != 'N')) ? 1 : 0;
          q.effLen = eff;
        }
        // exon intervals from the header comment (SeqSet.hpp:933-976): numbers[0] ignored, then (start, end) pairs
        std::vector<std::pair<int, int>> ex;
        if (r.hasComment) {
          std::vector<int> nums;
          int v = 0;
          for (char c : r.comment) {
            if (c >= '0' && c <= '9') v = v * 10 + (c - '0');
            else { nums.push_back(v); v = 0; }
          }
          if (v) nums.push_back(v);
          if (!nums.empty()) {
            for (size_t k = 1; k < nums.size(); k += 2) ex.push_back({nums[k], k + 1 < nums.size() ? nums[k + 1] : 0});
          } else ex.push_back({0, L - 1});
        } else ex.push_back({0, L - 1});
        q.mask.assign(L, 0);  // SetSeqExonInfo (638-723)
        for (auto &e : ex) {
          const int j0 = std::max(e.first, 0), j1 = std::min(e.second, L - 1);
          if (j1 >= j0) memset(q.mask.data() + j0, 1, (size_t)(j1 - j0 + 1));
        }
        for (size_t k = 1; k < ex.size(); ++k)
          if (ex[k].first > ex[k - 1].second + 1) { q.gap = true; break; }
      }
    };
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  // identical sequences merge into the first one (Genotyper.hpp:718-721): found through the hash, confirmed by comparing
  std::unordered_multimap<uint64_t, int> firstWithSeq;
  firstWithSeq.reserve(recs.size() * 2);
  for (size_t i = 0; i < recs.size(); ++i) {
    SeqRec &r = recs[i];
    if (selected && !selected->count(r.id)) continue;
    Pre &q = pre[i];
    bool merged = false;
    auto range = firstWithSeq.equal_range(q.hash);
    for (auto it = range.first; it != range.second; ++it)
      if (seqs[it->second] == r.seq) { al[it->second].weight += 1; merged = true; break; }
    if (merged) continue;
    firstWithSeq.emplace(q.hash, (int)al.size());
    AlleleMeta m;
    m.name = r.id;
    m.seqLen = (int)r.seq.size();
    m.effLen = q.effLen;
    if (q.gap) rnaData = false;
    al.push_back(m);
    seqs.push_back(std::move(r.seq));
    exon.push_back(std::move(q.mask));
  }
  const int A = (int)al.size();
  const auto tC = std::chrono::steady_clock::now();
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k host] reference: file parse %.1f ms, alleles %.1f ms\n", std::chrono::duration<double, std::milli>(tB - tA).count(), std::chrono::duration<double, std::milli>(tC - tB).count());
  if (A == 0) { err = "no sequences in " + fasta; return false; }
  if (!rnaData) {  // SeqSet::UpdateDnaSeqWeight (1008-1029): weight = multiplicity of the exon-only sequence
    std::unordered_map<std::string, int> w;
    std::vector<std::string> exonSeq(A);
    for (int a = 0; a < A; ++a)
      for (int p = 0; p < al[a].seqLen; ++p)
        if (exon[a][p]) exonSeq[a] += seqs[a][p];
    for (int a = 0; a < A; ++a) w[exonSeq[a]] += al[a].weight;
    for (int a = 0; a < A; ++a) al[a].weight = w[exonSeq[a]];
  }
  // Genotyper::InitAlleleInfo (559-682)
  std::unordered_map<std::string, int> geneId, majorId;
  for (int a = 0; a < A; ++a) {
    std::string g, m;
    splitName(al[a].name, g, m, 0);
    auto gi = geneId.find(g);
    if (gi == geneId.end()) { gi = geneId.emplace(g, (int)geneName.size()).first; geneName.push_back(g); }
    auto mi = majorId.find(m);
    if (mi == majorId.end()) { mi = majorId.emplace(m, (int)majorName.size()).first; majorName.push_back(m); }
    al[a].gene = gi->second;
    al[a].major = mi->second;
  }
  const int Gn = (int)geneName.size();
  // gene similarity from the lexicographically smallest allele sequence of each gene (598-638)
  std::vector<std::unordered_map<uint64_t, int>> prof(Gn);
  {
    // (the comparisons run over long common prefixes: the host threads each take a piece of the alleles, and the pieces' picks are
    // combined in allele order with the same strict comparison, so the first of equal sequences still wins)
    const int T = (int)std::max(1u, std::min(16u, std::min(std::thread::hardware_concurrency(), (unsigned)(A / 1024 + 1))));
    std::vector<std::vector<int>> part(T, std::vector<int>(Gn, -1));
    auto scan = [&](int t) {
      const int a0 = (int)((int64_t)A * t / T), a1 = (int)((int64_t)A * (t + 1) / T);
      std::vector<int> &pk = part[t];
      for (int a = a0; a < a1; ++a) {
        const int g = al[a].gene;
        if (pk[g] == -1 || strcmp(seqs[a].c_str(), seqs[pk[g]].c_str()) < 0) pk[g] = a;
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(scan, t);
      scan(0);
      for (auto &x : th) x.join();
    }
    std::vector<int> pick(Gn, -1);
    for (int t = 0; t < T; ++t)
      for (int g = 0; g < Gn; ++g)
        if (part[t][g] != -1 && (pick[g] == -1 || strcmp(seqs[part[t][g]].c_str(), seqs[pick[g]].c_str()) < 0)) pick[g] = part[t][g];
    overGenes(Gn, [&](int g) { kmerProfile(seqs[pick[g]], prof[g]); });
  }
  geneSim.assign(Gn, std::vector<double>(Gn, 0));
  {
    auto row = [&](int i) {
      for (int j = 0; j < Gn; ++j) {
        if (i == j) { geneSim[i][j] = 1.0; continue; }
        int total = 0, shared = 0;
        for (auto &kv : prof[i]) {
          total += kv.second;
          if (prof[j].count(kv.first)) shared += kv.second;
        }
        geneSim[i][j] = (double)shared / (double)total;
      }
    };
    overGenes(Gn, row);
  }
  // effective-length repair (641-681): alleles > 500 shorter than their gene's modal length inherit the mode
  for (int g = 0; g < Gn; ++g) {
    std::vector<int> lens;
    for (int a = 0; a < A; ++a)
      if (al[a].gene == g) lens.push_back(al[a].effLen);
    std::sort(lens.begin(), lens.end());
    int mode = 0, run = 0;
    for (size_t j = 0; j < lens.size();) {
      size_t k = j;
      while (k < lens.size() && lens[k] == lens[j]) ++k;
      if ((int)(k - j) > run) { run = (int)(k - j); mode = lens[j]; }
      j = k;
    }
    for (int a = 0; a < A; ++a)
      if (al[a].gene == g && al[a].effLen < mode - 500) al[a].effLen = mode;
  }
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k host] reference: naming + gene similarity + effective lengths %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tC).count());
  return true;
}

}  // namespace t1k

// ------------------------------------------------------------------------------------------------------------------
// large host blocks (t1k_host.h)
// ------------------------------------------------------------------------------------------------------------------
namespace t1k {
void bigBlockAdvise(void *p, size_t bytes) {
  static const bool thp = getenv("T1K_NO_THP") == nullptr;
  if (!thp) return;
  const uintptr_t two = (uintptr_t)2 << 20;
  const uintptr_t lo = ((uintptr_t)p + two - 1) & ~(two - 1), hi = ((uintptr_t)p + bytes) & ~(two - 1);
  if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);  // (refused on hosts without the feature: nothing is lost)
}
}  // namespace t1k
