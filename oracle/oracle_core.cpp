// oracle/oracle_core.cpp -- TEST INFRASTRUCTURE ONLY (see oracle_core.hpp header).
// CPU restatement of the T1K genotyper hot path; every routine cites the reference lines it follows.
#include "oracle_core.hpp"
#include <cctype>

#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace t1k_oracle {

// ---------------------------------------------------------------------------------------------------
// sequence helpers
// ---------------------------------------------------------------------------------------------------
static inline int baseCode(char c) {  // Genotyper.cpp:37-42 nucToNum; anything else is -1 there
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

std::string reverseComplement(const std::string &s) {  // SeqSet.hpp:2103-2114
  std::string o(s.size(), 'N');
  int n = (int)s.size();
  for (int i = 0; i < n; ++i) {
    char c = s[n - 1 - i];
    int b = baseCode(c);
    o[i] = (c != 'N' && b >= 0) ? "ACGT"[3 - b] : 'N';
  }
  return o;
}

static inline bool baseEq(char a, char b) { return a == b || a == 'N' || b == 'N'; }  // AlignAlgo.hpp:304-305

// ---------------------------------------------------------------------------------------------------
// AlignAlgo::GlobalAlignment (AlignAlgo.hpp:215-421): banded affine-gap global alignment + traceback.
// Scores +2/-2, gap open -4, extend -1 (AlignAlgo.hpp:12-15).  Matrices are indexed [row i over p][col j over t].
// ---------------------------------------------------------------------------------------------------
int globalAlignment(const char *t, int lent, const char *p, int lenp, std::vector<int8_t> &ops) {
  ops.clear();
  if (lent == 0 || lenp == 0) return 0;  // 217-221
  if (lent == 1 && lenp == 1) {          // 222-236
    if (baseEq(t[0], p[0])) { ops.push_back(OP_MATCH); return 2; }
    ops.push_back(OP_MISMATCH);
    return -2;
  }
  const int band = 5;
  int leftBand = band, rightBand = band;  // 240-245
  if (lent > lenp) rightBand += lent - lenp;
  else if (lent < lenp) leftBand += lenp - lent;
  const int W = lent + 1;
  const int negInf = (lent + 1) * (lenp + 1) * -4;  // 248
  static thread_local std::vector<int> M, E, F;
  size_t cells = (size_t)(lenp + 1) * W;
  if (M.size() < cells) { M.resize(cells); E.resize(cells); F.resize(cells); }
  int *m = M.data(), *e = E.data(), *f = F.data();
  m[0] = e[0] = f[0] = 0;
  for (int i = 1; i <= lenp; ++i) {  // 256-262
    e[i * W] = -4 + i * -1;
    f[i * W] = -4 + i * -4;
    m[i * W] = -4 + i * -4;
  }
  for (int j = 1; j <= lent; ++j) {  // 264-270; e[0][j] uses the stale loop variable i == lenp+1
    f[j] = -4 + j * -1;
    e[j] = -4 + (lenp + 1) * -4;
    m[j] = -4 + j * -4;
  }
  for (int i = 1; i <= lenp; ++i) {  // 272-311
    int start = (i - leftBand < 1) ? 1 : (i - leftBand);
    int end = (i + rightBand > lent) ? lent : (i + rightBand);
    if (start > 1) e[i * W + start - 1] = f[i * W + start - 1] = m[i * W + start - 1] = negInf;
    if (end < lent) e[i * W + end + 1] = f[i * W + end + 1] = m[i * W + end + 1] = negInf;
    for (int j = start; j <= end; ++j) {
      int s = std::max(e[(i - 1) * W + j] - 1, m[(i - 1) * W + j] - 5);
      e[i * W + j] = s;
      s = std::max(f[i * W + j - 1] - 1, m[i * W + j - 1] - 5);
      f[i * W + j] = s;
      s = m[(i - 1) * W + j - 1] + (baseEq(t[j - 1], p[i - 1]) ? 2 : -2);
      s = std::max(s, e[i * W + j]);
      s = std::max(s, f[i * W + j]);
      m[i * W + j] = s;
    }
  }
  int ret = m[lenp * W + lent];
  int ti = lenp, tj = lent, mat = 0;  // 323-408
  while (ti > 0 || tj > 0) {
    if (mat == 0) {
      int a = OP_INSERT;
      if (f[ti * W + tj] >= e[ti * W + tj]) a = OP_DELETE;
      if (ti > 0 && tj > 0) {
        bool eq = baseEq(t[tj - 1], p[ti - 1]);
        if (m[(ti - 1) * W + tj - 1] + (eq ? 2 : -2) == m[ti * W + tj]) a = eq ? OP_MATCH : OP_MISMATCH;
      }
      if (a == OP_MATCH || a == OP_MISMATCH) { ops.push_back((int8_t)a); --ti; --tj; }
      else if (a == OP_INSERT) mat = 1;
      else mat = 2;
    } else if (mat == 1) {
      ops.push_back(OP_INSERT);
      if (ti > 0) {
        if (m[(ti - 1) * W + tj] - 5 == e[ti * W + tj]) mat = 0;
        --ti;
      } else mat = 2;
    } else {
      ops.push_back(OP_DELETE);
      if (tj > 0) {
        if (m[ti * W + tj - 1] - 5 == f[ti * W + tj]) mat = 0;
        --tj;
      } else mat = 1;
    }
  }
  std::reverse(ops.begin(), ops.end());
  return ret;
}

static int countMatches(const std::vector<int8_t> &ops) {  // SeqSet::GetAlignStats (SeqSet.hpp:438-455), match count only
  int c = 0;
  for (int8_t o : ops) c += (o == OP_MATCH);
  return c;
}

// ---------------------------------------------------------------------------------------------------
// ordering of overlaps (SeqSet.hpp:103-127)
// ---------------------------------------------------------------------------------------------------
bool overlapBefore(const Overlap &a, const Overlap &b) {
  if (a.matchCnt != b.matchCnt) return a.matchCnt > b.matchCnt;
  if (a.similarity != b.similarity) return a.similarity > b.similarity;
  if (a.readEnd - a.readStart != b.readEnd - b.readStart) return a.readEnd - a.readStart > b.readEnd - b.readStart;
  if (a.seqIdx != b.seqIdx) return a.seqIdx < b.seqIdx;
  if (a.strand != b.strand) return a.strand < b.strand;
  if (a.readStart != b.readStart) return a.readStart < b.readStart;
  if (a.readEnd != b.readEnd) return a.readEnd < b.readEnd;
  if (a.seqStart != b.seqStart) return a.seqStart < b.seqStart;
  return a.seqEnd < b.seqEnd;
}

// ---------------------------------------------------------------------------------------------------
// file reading (ReadFiles.hpp:155-204 + kseq.h record rules)
// ---------------------------------------------------------------------------------------------------
// kseq.h:185-224 (kseq_read) over kseq.h:93-141 (ks_getuntil2) and ks_getc, followed step by step: the stream is filled 16 384 bytes at
// a time (kseq.h:234), `last` is kseq_t::last_char.  Pinned against the reference's own reader on odd and damaged files through the
// product's reader tests (tests/test_host_reads_cpu.py, oracle/_ref/reads_harness).
namespace {
struct KStream {
  gzFile fp;
  std::vector<char> buf;
  int begin = 0, end = 0;
  bool isEof = false;
  int last = 0;
  explicit KStream(const std::string &path) : buf(16384) { fp = gzopen(path.c_str(), "r"); }
  ~KStream() { if (fp) gzclose(fp); }
  void fill() {
    begin = 0;
    end = gzread(fp, buf.data(), (unsigned)buf.size());
    if (end < (int)buf.size()) isEof = true;
    if (end < 0) end = 0;
  }
  int getc() {
    if (isEof && begin >= end) return -1;
    if (begin >= end) { fill(); if (end == 0) return -1; }
    return (unsigned char)buf[begin++];
  }
  int getuntil(bool line, std::string &str, int *dret, bool append) {
    if (dret) *dret = 0;
    if (!append) str.clear();
    if (begin >= end && isEof) return -1;
    for (;;) {
      if (begin >= end) {
        if (isEof) break;
        fill();
        if (end == 0) break;
      }
      int i = begin;
      if (line) { while (i < end && buf[i] != '\n') ++i; }
      else { while (i < end && !isspace((unsigned char)buf[i])) ++i; }
      str.append(buf.data() + begin, (size_t)(i - begin));
      begin = i + 1;
      if (i < end) { if (dret) *dret = (unsigned char)buf[i]; break; }
    }
    if (line && str.size() > 1 && str.back() == '\r') str.pop_back();
    return (int)str.size();
  }
  int read(std::string &name, std::string &comment, std::string &seq, std::string &qual) {
    int c;
    if (last == 0) {
      while ((c = getc()) != -1 && c != '>' && c != '@') {}
      if (c == -1) return -1;
      last = c;
    }
    comment.clear(); seq.clear(); qual.clear();
    if (getuntil(false, name, &c, false) < 0) return -1;
    if (c != '\n') getuntil(true, comment, nullptr, false);
    while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
      if (c == '\n') continue;
      seq.push_back((char)c);
      getuntil(true, seq, nullptr, true);
    }
    if (c == '>' || c == '@') last = c;
    if (c != '+') return (int)seq.size();
    while ((c = getc()) != -1 && c != '\n') {}
    if (c == -1) return -2;
    while (getuntil(true, qual, nullptr, true) >= 0 && qual.size() < seq.size()) {}
    last = 0;
    if (seq.size() != qual.size()) return -2;
    return (int)seq.size();
  }
};
}  // namespace

bool readAllRecords(const std::string &path, std::vector<SeqRecord> &out) {
  KStream in(path);
  if (!in.fp) return false;
  std::string name, comment, seq, qual;
  while (in.read(name, comment, seq, qual) >= 0) {  // ReadFiles::Next (ReadFiles.hpp:161-164): a negative return ends this file
    SeqRecord r;
    r.id = name.c_str();  // strdup (ReadFiles.hpp:183-198)
    r.rawId = r.id;
    int n = (int)r.id.size();  // ReadFiles.hpp:185-189: strip trailing /1 or /2
    if (n >= 2 && (r.id[n - 1] == '1' || r.id[n - 1] == '2') && r.id[n - 2] == '/') r.id.resize(n - 2);
    r.seq = seq.c_str();
    r.qual = qual.c_str();
    r.hasComment = !comment.empty();
    if (r.hasComment) r.comment = comment.c_str();
    out.push_back(std::move(r));
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// reference loading
// ---------------------------------------------------------------------------------------------------
void Oracle::addAllele(const std::string &name, const std::string &comment, const std::string &seq, bool hasComment) {
  // SeqSet::InputRefSeq (SeqSet.hpp:906-982) with initExonInfo = true
  AlleleRec a;
  a.name = name;
  a.seq = seq;
  int L = (int)seq.size();
  a.effectiveLen = 0;  // ComputeEffectiveLen (747-758): runs of N count once
  for (int i = 0; i < L; ++i)
    if (seq[i] != 'N' || (i > 0 && seq[i - 1] != 'N')) ++a.effectiveLen;
  a.separators.push_back(-1);  // 924-928
  for (int i = 0; i < L; ++i)
    if (seq[i] == 'N') a.separators.push_back(i);
  a.separators.push_back(L);
  std::vector<std::pair<int, int>> exons;
  if (hasComment) {  // 934-969
    std::vector<int> nums;
    int n = 0;
    for (char c : comment) {
      if (c >= '0' && c <= '9') n = n * 10 + (c - '0');
      else { nums.push_back(n); n = 0; }
    }
    if (n) nums.push_back(n);
    int size = (int)nums.size();
    if (size > 0) {
      for (int i = 1; i < size; i += 2) {
        int b = (i + 1 < size) ? nums[i + 1] : 0;  // the reference reads past the end here; well-formed headers never do
        exons.push_back({nums[i], b});
      }
    } else exons.push_back({0, L - 1});
  } else exons.push_back({0, L - 1});
  // SetSeqExonInfo (638-723)
  a.exon.assign(L, 0);
  for (auto &ex : exons)
    for (int j = ex.first; j <= ex.second && j < L; ++j)
      if (j >= 0) a.exon[j] = 1;
  for (size_t i = 1; i < exons.size(); ++i)
    if (exons[i].first > exons[i - 1].second + 1) { rnaData = false; break; }
  a.cov.assign((size_t)L * 4, 0);
  alleles.push_back(std::move(a));
}

int Oracle::loadReference(const std::string &fasta) {
  // Genotyper::InitRefSet (Genotyper.hpp:707-730): identical sequences collapse onto the first name, weight++
  std::vector<SeqRecord> recs;
  if (!readAllRecords(fasta, recs)) return -1;
  std::map<std::string, int> used;
  for (auto &r : recs) {
    auto it = used.find(r.seq);
    if (it != used.end()) alleles[it->second].weight += 1;
    else {
      used[r.seq] = (int)alleles.size();
      addAllele(r.id, r.comment, r.seq, r.hasComment);
    }
  }
  finishReference();
  return (int)alleles.size();
}

void Oracle::parseAlleleName(const std::string &allele, std::string &gene, std::string &major) const {
  // Genotyper::ParseAlleleName (Genotyper.hpp:63-131), fieldsType = 0
  int parseType = 1;
  int fields = prm.alleleDigitUnits;
  char delim = 0;
  if (fields == -1) {
    fields = 3;
    if (allele.find(':') != std::string::npos) { delim = ':'; parseType = 2; }
  }
  if (prm.alleleDelimiter != 0) { delim = prm.alleleDelimiter; parseType = 2; }
  size_t star = allele.find('*');
  size_t i = star == std::string::npos ? allele.size() : star;
  gene = allele.substr(0, i);
  if (parseType == 1) {
    size_t j = 0;
    while ((int)j <= fields && i + j < allele.size()) ++j;
    major = allele.substr(0, i + j);
  } else {
    int k = 0;
    size_t j = i;
    for (; j < allele.size(); ++j)
      if (allele[j] == delim) { ++k; if (k >= fields) break; }
    major = allele.substr(0, j);
  }
}

void Oracle::finishReference() {
  int A = (int)alleles.size();
  // SeqSet::UpdateDnaSeqWeight (SeqSet.hpp:1008-1029)
  if (!rnaData) {
    std::map<std::string, int> w;
    std::vector<std::string> ex(A);
    for (int i = 0; i < A; ++i) {
      for (size_t p = 0; p < alleles[i].seq.size(); ++p)
        if (alleles[i].exon[p]) ex[i] += alleles[i].seq[p];
    }
    for (int i = 0; i < A; ++i) w[ex[i]] += alleles[i].weight;
    for (int i = 0; i < A; ++i) alleles[i].weight = w[ex[i]];
  }
  // Genotyper::InitAlleleInfo (Genotyper.hpp:559-682): gene / major-allele ids in first-appearance order
  std::map<std::string, int> g2i, m2i;
  for (int i = 0; i < A; ++i) {
    std::string g, m;
    parseAlleleName(alleles[i].name, g, m);
    if (!g2i.count(g)) { g2i[g] = (int)geneNames.size(); geneNames.push_back(g); }
    if (!m2i.count(m)) { m2i[m] = (int)majorNames.size(); majorNames.push_back(m); }
    alleles[i].gene = g2i[g];
    alleles[i].majorAllele = m2i[m];
  }
  // effective-length fix (641-681): alleles > 500 shorter than the gene's modal effective length get the mode
  for (int g = 0; g < (int)geneNames.size(); ++g) {
    std::vector<int> ids, lens;
    for (int i = 0; i < A; ++i)
      if (alleles[i].gene == g) { ids.push_back(i); lens.push_back(alleles[i].effectiveLen); }
    std::sort(lens.begin(), lens.end());
    int mode = 0, best = 0;
    for (size_t j = 0; j < lens.size();) {
      size_t k = j;
      while (k < lens.size() && lens[k] == lens[j]) ++k;
      if ((int)(k - j) > best) { best = (int)(k - j); mode = lens[j]; }
      j = k;
    }
    for (int id : ids)
      if (alleles[id].effectiveLen < mode - 500) alleles[id].effectiveLen = mode;
  }
  buildIndex();
}

// KmerIndex::BuildIndexFromRead (KmerIndex.hpp:107-130) + KmerCode::Append (KmerCode.hpp:93-108), flattened into a
// direct-address table; postings keep the reference's insertion order (allele, then offset).
void Oracle::buildIndex() {
  const int k = prm.k;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  size_t nKeys = (size_t)1 << (2 * k);
  std::vector<uint32_t> cnt(nKeys + 1, 0);
  std::vector<std::pair<uint32_t, Posting>> tmp;
  for (int a = 0; a < (int)alleles.size(); ++a) {
    const std::string &s = alleles[a].seq;
    int len = (int)s.size();
    if (len < k) continue;
    uint64_t code = 0, prev = 0;  // prevKmerCode starts as code 0 (KmerIndex.hpp:115)
    int invalid = -1;
    for (int i = 0; i < len; ++i) {
      if (invalid != -1) ++invalid;
      code = ((code << 2) & mask) | (uint64_t)(s[i] == 'N' ? prm.nBaseCode : (baseCode(s[i]) & 3));
      if (s[i] == 'N') invalid = 0;
      if (invalid >= k) invalid = -1;
      if (i < k - 1) continue;
      if (invalid == -1 && (i == k || code != prev)) {
        tmp.push_back({(uint32_t)code, Posting{(uint32_t)a, (uint32_t)(i - k + 1)}});
        ++cnt[code + 1];
      }
      prev = code;
    }
  }
  idxStart.assign(nKeys + 1, 0);
  for (size_t i = 0; i < nKeys; ++i) idxStart[i + 1] = idxStart[i] + cnt[i + 1];
  idxPost.resize(tmp.size());
  std::vector<uint32_t> cur(idxStart.begin(), idxStart.end() - 1);
  for (auto &e : tmp) idxPost[cur[e.first]++] = e.second;
}

// SeqSet::GetHitsFromRead (SeqSet.hpp:1071-1229) with strand=0, barcode=-1, puse=NULL, allowTotalSkip=false
void Oracle::seedHits(const std::string &read, std::vector<int> &strand, std::vector<int> &readOff, std::vector<Posting> &post) {
  const int k = prm.k;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  const int skipLimit = k / 2;  // 1081
  int len = (int)read.size();
  std::string rc = reverseComplement(read);
  uint64_t prev = 0;  // prevKmerCode is NOT reset between strands
  for (int pass = 0; pass < 2; ++pass) {
    const std::string &s = pass == 0 ? read : rc;
    uint64_t code = 0;
    int invalid = -1, skipCnt = 0;
    for (int i = 0; i < len; ++i) {
      if (invalid != -1) ++invalid;
      code = ((code << 2) & mask) | (uint64_t)(s[i] == 'N' ? prm.nBaseCode : (baseCode(s[i]) & 3));
      if (s[i] == 'N') invalid = 0;
      if (invalid >= k) invalid = -1;
      if (i < k - 1) continue;
      if (i == k - 1 || code != prev) {  // 1104
        uint32_t b = 0, e = 0;
        if (invalid == -1) { b = idxStart[code]; e = idxStart[code + 1]; }
        int size = (int)(e - b);
        ++stats.lookups;
        if (size >= 100 && i != k - 1 && i != len - 1 && skipCnt < skipLimit) {  // 1109-1116, SURVEY H2: prev is not updated
          ++skipCnt;
          continue;
        }
        skipCnt = 0;
        stats.postings += size;
        for (uint32_t q = b; q < e; ++q) {
          strand.push_back(pass == 0 ? 1 : -1);
          readOff.push_back(i - k + 1);
          post.push_back(idxPost[q]);
        }
      }
      prev = code;
    }
  }
}

// SeqSet::LongestIncreasingSubsequence (SeqSet.hpp:352-436) on pairs (a = read offset, b = allele offset)
static void lisChain(const std::vector<std::pair<int, int>> &hits, std::vector<std::pair<int, int>> &out) {
  out.clear();
  int n = (int)hits.size();
  if (n == 0) return;
  std::vector<int> top(n), link(n);
  top[0] = 0; link[0] = -1;
  int ret = 1;
  for (int i = 1; i < n; ++i) {
    int tag;
    if (hits[top[ret - 1]].first <= hits[i].first) tag = ret - 1;
    else {  // BinarySearch_LIS (327-348)
      int l = 0, r = ret - 1;
      tag = -2;
      while (l <= r) {
        int m = (l + r) / 2;
        if (hits[i].first == hits[top[m]].first) { tag = m; break; }
        if (hits[i].first < hits[top[m]].first) r = m - 1; else l = m + 1;
      }
      if (tag == -2) tag = l - 1;
    }
    if (tag == -1) { top[0] = i; link[i] = -1; }
    else if (hits[i].first > hits[top[tag]].first) {
      if (tag == ret - 1) { top[ret] = i; ++ret; link[i] = top[tag]; }
      else if (hits[i].first < hits[top[tag + 1]].first) { top[tag + 1] = i; link[i] = top[tag]; }
    }
  }
  std::vector<std::pair<int, int>> lis(ret);
  int kx = top[ret - 1];
  for (int i = ret - 1; i >= 0; --i) { lis[i] = hits[kx]; kx = link[kx]; }
  out.push_back(lis[0]);  // 418-429: drop entries repeating the previous kept allele offset
  for (int i = 1; i < ret; ++i)
    if (lis[i].second != out.back().second) out.push_back(lis[i]);
}

static int hitLength(const std::vector<std::pair<int, int>> &c, bool onRead, int k) {  // SeqSet.hpp:1032-1069
  int n = (int)c.size(), ret = 0;
  for (int i = 0; i < n;) {
    int j = i + 1;
    for (; j < n; ++j) {
      int cur = onRead ? c[j].first : c[j].second, pre = onRead ? c[j - 1].first : c[j - 1].second;
      if (cur > pre + k - 1) break;
    }
    int last = onRead ? c[j - 1].first : c[j - 1].second, first = onRead ? c[i].first : c[i].second;
    ret += last - first + k;
    i = j;
  }
  return ret;
}

// SeqSet::SortHits + GetOverlapsFromHits (SeqSet.hpp:1558-1590, 1232-1556), filter = 0, all sequences isRef
void Oracle::candidatesFromHits(const std::vector<int> &strand, const std::vector<int> &readOff, const std::vector<Posting> &post,
                                std::vector<Cand> &cands) {
  const int k = prm.k;
  size_t n = strand.size();
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    if (strand[x] != strand[y]) return strand[x] < strand[y];
    return post[x].allele < post[y].allele;
  });
  struct T3 { int a, b, c; };
  std::vector<T3> diff;
  std::vector<std::pair<int, int>> conc, chain;
  std::vector<int> used;
  for (size_t i = 0; i < n;) {
    size_t j = i + 1;
    while (j < n && strand[order[j]] == strand[order[i]] && post[order[j]].allele == post[order[i]].allele) ++j;
    if (j - i < 3) { i = j; continue; }  // refMinHitRequired = 3 (1253, 1314)
    diff.clear();
    int maxOff = 0;
    for (size_t q = i; q < j; ++q) {
      int a = readOff[order[q]], b = (int)post[order[q]].offset;
      diff.push_back({a, b, a - b});
      maxOff = std::max(maxOff, a);
    }
    std::sort(diff.begin(), diff.end(), [](const T3 &x, const T3 &y) {  // CompSortHitCoordDiff (266-274)
      if (x.c != y.c) return x.c < y.c;
      if (x.b != y.b) return x.b < y.b;
      return x.a < y.a;
    });
    used.assign(maxOff + 1, -1);
    int m = (int)diff.size();
    int dominant = 0;
    for (int s = 0; s < m;) {  // 1360-1551
      int curDiff = diff[s].c, curCnt = 1, domCnt = 0;
      used[diff[s].a] = -1;
      int e = s + 1;
      for (; e < m; ++e) {
        int d = std::abs(diff[e].c - diff[e - 1].c);
        if (d > prm.radius) break;
        if (d == 0) ++curCnt;
        else {
          if (curCnt > domCnt) { dominant = curDiff; domCnt = curCnt; }
          curDiff = diff[e].c; curCnt = 1;
        }
        used[diff[e].a] = -1;
      }
      if (curCnt > domCnt) dominant = curDiff;  // 1393-1397
      if (e - s < 3 || (e - s) * k < prm.hitLenRequired) { s = e; continue; }
      conc.clear();
      for (int q = s; q < e; ++q) conc.push_back({diff[q].a, diff[q].b});
      // keep, per read offset, the hits nearest to the dominant diagonal (1437-1456)
      for (auto &h : conc) {
        int d = std::abs(h.first - h.second - dominant);
        if (used[h.first] == -1 || used[h.first] > d) used[h.first] = d;
      }
      size_t l = 0;
      for (auto &h : conc)
        if (std::abs(h.first - h.second - dominant) == used[h.first]) conc[l++] = h;
      conc.resize(l);
      std::sort(conc.begin(), conc.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) {  // CompSortPairBInc
        if (x.second != y.second) return x.second < y.second;
        return x.first < y.first;
      });
      lisChain(conc, chain);
      int lisSize = (int)chain.size();
      if (lisSize * k < prm.hitLenRequired) { s = e; continue; }
      int hitLen = hitLength(chain, true, k);
      if (hitLen < prm.hitLenRequired || hitLength(chain, false, k) < prm.hitLenRequired) { s = e; continue; }
      Cand c;
      c.o.seqIdx = (int)post[order[i]].allele;
      c.o.readStart = chain[0].first;
      c.o.readEnd = chain[lisSize - 1].first + k - 1;
      c.o.strand = strand[order[i]];
      c.o.seqStart = chain[0].second;
      c.o.seqEnd = chain[lisSize - 1].second + k - 1;
      c.o.matchCnt = 2 * hitLen;
      c.o.similarity = 0;
      c.chain = chain;
      cands.push_back(std::move(c));
      s = e;
    }
    i = j;
  }
}

bool Oracle::separatorInRange(int s, int e, int seqIdx) const {  // SeqSet.hpp:487-498
  for (int p : alleles[seqIdx].separators)
    if (p >= s && p <= e) return true;
  return false;
}

static bool lowComplexity(const std::string &r, int rs, int re) {  // SeqSet::IsOverlapLowComplex (458-485)
  int cnt[4] = {0, 0, 0, 0};
  for (int i = rs; i <= re; ++i) {
    int b = baseCode(r[i]);
    if (r[i] == 'N' || b < 0) continue;
    ++cnt[b];
  }
  int low = 0, lowTotal = 0;
  for (int i = 0; i < 4; ++i)
    if (cnt[i] <= 2) { ++low; lowTotal += cnt[i]; }
  if (lowTotal * 7 >= re - rs + 1) return false;
  return low >= 2;
}

// SeqSet::GetOverlapsFromRead (SeqSet.hpp:1594-1912)
int Oracle::overlapsFromRead(const std::string &read, const std::string &rc, std::vector<Overlap> &out) {
  const int k = prm.k;
  out.clear();
  if ((int)read.size() < k) return -1;
  std::vector<int> strand, readOff;
  std::vector<Posting> post;
  seedHits(read, strand, readOff, post);
  std::vector<Cand> cands;
  candidatesFromHits(strand, readOff, post, cands);
  if (cands.empty()) return 0;
  size_t best = 0;  // 1619-1648: similarity is still 0 for everyone here
  for (size_t i = 1; i < cands.size(); ++i)
    if (overlapBefore(cands[i].o, cands[best].o)) best = i;
  int keepStrand = cands[best].o.strand;
  std::vector<int8_t> ops;
  for (auto &c : cands) {
    if (c.o.strand != keepStrand) continue;
    ++stats.candidates;
    const std::string &r = c.o.strand == 1 ? read : rc;
    const std::string &ref = alleles[c.o.seqIdx].seq;
    int matchCnt = 2 * k;  // 1697-1833
    for (size_t j = 1; j < c.chain.size(); ++j) {
      auto p = c.chain[j - 1], q = c.chain[j];
      bool sameDiag = (p.second - p.first) == (q.second - q.first);
      bool readOv = p.first + k - 1 >= q.first, seqOv = p.second + k - 1 >= q.second;
      if (sameDiag) {
        if (readOv) matchCnt += 2 * (q.first - p.first);
        else {
          matchCnt += 2 * k;
          int lt = q.second - (p.second + k), lp = q.first - (p.first + k);
          ++stats.gaCalls; stats.gaCells += (uint64_t)lt * lp;
          globalAlignment(ref.data() + p.second + k, lt, r.data() + p.first + k, lp, ops);
          matchCnt += 2 * countMatches(ops);
        }
      } else {  // 1761-1832
        if (readOv && !seqOv) matchCnt += 2 * (q.first - p.first);
        else if (!readOv && seqOv) matchCnt += 2 * (q.second - p.second);
        else if (readOv && seqOv) matchCnt += 2 * std::min(q.first - p.first, q.second - p.second);
        else {
          matchCnt += 2 * k;
          int lt = q.second - (p.second + k), lp = q.first - (p.first + k);
          ++stats.gaCalls; stats.gaCells += (uint64_t)lt * lp;
          globalAlignment(ref.data() + p.second + k, lt, r.data() + p.first + k, lp, ops);
          matchCnt += 2 * countMatches(ops);
        }
      }
    }
    Overlap o = c.o;
    o.matchCnt = matchCnt;
    o.similarity = (double)matchCnt / (o.seqEnd - o.seqStart + 1 + o.readEnd - o.readStart + 1);  // 1838-1840
    if (lowComplexity(r, o.readStart, o.readEnd)) o.similarity = 0;                                // 1844-1845
    if (o.similarity < prm.refSeqSimilarity) continue;                                             // 1894-1908
    out.push_back(o);
  }
  return (int)out.size();
}

// SeqSet::ExtendOverlap (SeqSet.hpp:1994-2100)
bool Oracle::extendOverlap(const std::string &r, const Overlap &o, Overlap &e) {
  const std::string &ref = alleles[o.seqIdx].seq;
  int len = (int)r.size(), refLen = (int)ref.size();
  std::vector<int8_t> ops;
  int lo = std::min(o.readStart, o.seqStart);
  int leftClip = 0, rightClip = 0;
  if (o.readStart > o.seqStart) leftClip = o.readStart - o.seqStart;
  for (int i = 0; i < lo; ++i)
    if (ref[o.seqStart - i - 1] == 'N') { leftClip = lo - i; lo = i; break; }
  ++stats.gaCalls; stats.gaCells += (uint64_t)lo * lo;
  globalAlignment(ref.data() + o.seqStart - lo, lo, r.data() + o.readStart - lo, lo, ops);
  int match = countMatches(ops);
  int ro = std::min(len - 1 - o.readEnd, refLen - 1 - o.seqEnd);
  if (len - 1 - o.readEnd > refLen - 1 - o.seqEnd) rightClip = len - 1 - o.readEnd - (refLen - 1 - o.seqEnd);
  for (int i = 0; i < ro; ++i)
    if (ref[o.seqEnd + 1 + i] == 'N') { rightClip = ro - i; ro = i; break; }
  ++stats.gaCalls; stats.gaCells += (uint64_t)ro * ro;
  globalAlignment(ref.data() + o.seqEnd + 1, ro, r.data() + o.readEnd + 1, ro, ops);
  match += countMatches(ops);
  e = o;
  e.readStart = o.readStart - lo; e.readEnd = o.readEnd + ro;
  e.seqStart = o.seqStart - lo; e.seqEnd = o.seqEnd + ro;
  e.matchCnt = 2 * match + o.matchCnt;
  e.similarity = (double)e.matchCnt / (e.readEnd - e.readStart + 1 + e.seqEnd - e.seqStart + 1);
  e.relaxedMatchCnt = e.matchCnt;
  e.leftClip = leftClip; e.rightClip = rightClip;
  bool ok = !(e.similarity < prm.refSeqSimilarity);  // 2074 (before clip credit)
  if (leftClip > 0 || rightClip > 0) {               // 2078-2087
    e.matchCnt += 2 * leftClip + 2 * rightClip;
    e.similarity = double(e.matchCnt) / (e.readEnd - e.readStart + 1 + e.seqEnd - e.seqStart + 1 + 2 * leftClip + 2 * rightClip);
  }
  return ok;
}

// SeqSet::AssignRead (SeqSet.hpp:2119-2303), barcode = -1
int Oracle::assignRead(const std::string &read, int weight, std::vector<Overlap> &out) {
  out.clear();
  ++stats.readEnds;
  std::string rc = reverseComplement(read);
  std::vector<Overlap> ov;
  int cnt = overlapsFromRead(read, rc, ov);
  if (cnt <= 0 || alleles.empty()) return -1;
  std::sort(ov.begin(), ov.end(), overlapBefore);  // 2140
  int len = (int)read.size();
  const std::string &r = ov[0].strand == -1 ? rc : read;
  std::vector<Overlap> ext;
  bool onlyConsiderClip = false;  // 2156-2186
  int goodMatchCnt = -1;
  for (auto &o : ov) {
    if (separatorInRange(o.seqStart, o.seqEnd, o.seqIdx)) continue;
    bool needClip = separatorInRange(o.seqStart - o.readStart, o.seqEnd + (len - o.readEnd - 1), o.seqIdx);
    if (onlyConsiderClip && o.matchCnt < goodMatchCnt && (!needClip || o.similarity < 0.95)) continue;
    Overlap e;
    if (extendOverlap(r, o, e)) {
      ext.push_back(e);
      if (!onlyConsiderClip && (goodMatchCnt == -1 || o.matchCnt > goodMatchCnt)) goodMatchCnt = o.matchCnt;
    } else onlyConsiderClip = true;
  }
  stats.extended += ext.size();
  if (!ext.empty() && weight >= 0) {  // 2188-2285
    int bestMatch = ext[0].matchCnt;
    {
      Overlap best = ext[0];
      for (auto &e : ext) if (overlapBefore(e, best)) best = e;
      bestMatch = best.matchCnt;
    }
    std::vector<int8_t> ops;
    for (auto &e : ext) {
      if (e.matchCnt >= bestMatch - 10) {
        ++stats.nearBest;
        AlleleRec &al = alleles[e.seqIdx];
        int lt = e.seqEnd - e.seqStart + 1, lp = e.readEnd - e.readStart + 1;
        ++stats.gaCalls; stats.gaCells += (uint64_t)lt * lp;
        globalAlignment(al.seq.data() + e.seqStart, lt, r.data() + e.readStart, lp, ops);
        if (prm.relaxIntronAlign) {  // 2215-2246
          int m = 0, refPos = e.seqStart;
          for (int8_t op : ops) {
            bool ex = refPos < (int)al.exon.size() ? al.exon[refPos] != 0 : false;
            if (ex) { if (op == OP_MATCH) ++m; }
            else ++m;
            if (op != OP_INSERT) ++refPos;
          }
          e.relaxedMatchCnt = 2 * m;
        } else e.relaxedMatchCnt = e.matchCnt;
        if (weight > 0) {  // 2253-2274
          int refPos = e.seqStart, readPos = e.readStart;
          for (int8_t op : ops) {
            if (op == OP_MATCH) {
              int b = baseCode(r[readPos]);
              if (r[readPos] != 'N' && b >= 0) al.cov[(size_t)refPos * 4 + b] += weight;
            }
            if (op != OP_INSERT) ++refPos;
            if (op != OP_DELETE) ++readPos;
          }
        }
      } else e.relaxedMatchCnt = 0;  // 2282
    }
  }
  if (ext.size() > 1000) {  // 2290-2298
    std::sort(ext.begin(), ext.end(), overlapBefore);
    size_t j = 1;
    for (; j < ext.size(); ++j)
      if (ext[j].similarity < ext[0].similarity - 0.1) break;
    ext.resize(j);
  }
  out = ext;
  return (int)out.size();
}

// ---------------------------------------------------------------------------------------------------
// SeqSet::ReadAssignmentToFragmentAssignment (SeqSet.hpp:2310-2655)
// ---------------------------------------------------------------------------------------------------
static bool fragBefore(const FragmentOverlap &a, const FragmentOverlap &b) {  // SeqSet.hpp:164-171
  if (a.matchCnt != b.matchCnt) return a.matchCnt > b.matchCnt;
  if (a.similarity != b.similarity) return a.similarity > b.similarity;
  return overlapBefore(a.o1, b.o1);
}

int Oracle::pairFragments(const std::vector<Overlap> &ov1, const std::vector<Overlap> *pov2, bool hasN, std::vector<FragmentOverlap> &assign) {
  assign.clear();
  std::vector<std::pair<int, int>> frags;
  int n1 = (int)ov1.size();
  if (!pov2) {
    for (int i = 0; i < n1; ++i) frags.push_back({i, -1});
  } else if (n1 == 0 || pov2->empty()) {  // dangling candidates (2330-2347)
    for (int i = 0; i < n1; ++i) frags.push_back({i, -1});
    for (int i = 0; i < (int)pov2->size(); ++i) frags.push_back({-1, i});
  } else {
    const std::vector<Overlap> &ov2 = *pov2;
    std::map<int, std::vector<int>> byAllele;
    for (int i = 0; i < (int)ov2.size(); ++i) byAllele[ov2[i].seqIdx].push_back(i);
    for (int i = 0; i < n1; ++i) {
      auto it = byAllele.find(ov1[i].seqIdx);
      if (it == byAllele.end()) continue;
      for (int j : it->second) {
        if (ov1[i].strand == ov2[j].strand) continue;  // 2369-2371
        if ((ov1[i].strand == 1 && ov1[i].seqStart < ov2[j].seqStart) || (ov1[i].strand == -1 && ov1[i].seqStart > ov2[j].seqStart))
          frags.push_back({i, j});
      }
    }
  }
  std::map<int, int> slotOfAllele;  // seqIdxToOverlapIdx: best fragment per allele (2385-2455)
  for (auto &fr : frags) {
    FragmentOverlap f;
    if (fr.first >= 0) {
      const Overlap &o = ov1[fr.first];
      f.matchCnt = o.matchCnt; f.similarity = o.similarity; f.seqIdx = o.seqIdx;
      f.seqStart = o.seqStart; f.seqEnd = o.seqEnd; f.hasMatePair = false; f.hasN = hasN; f.o1FromR2 = false;
      f.o1 = o; f.relaxedMatchCnt = o.relaxedMatchCnt;
      if (fr.second >= 0) {
        const Overlap &o2 = (*pov2)[fr.second];
        f.matchCnt += o2.matchCnt;
        f.relaxedMatchCnt += o2.relaxedMatchCnt;
        if (o.strand == 1) f.seqEnd = o2.seqEnd; else f.seqStart = o2.seqStart;
        f.similarity = (double)f.matchCnt / (o.readEnd - o.readStart + 1 + o2.readEnd - o2.readStart + 1 + o.seqEnd - o.seqStart + 1 +
                                             o2.seqEnd - o2.seqStart + 1 + 2 * o.leftClip + 2 * o.rightClip + 2 * o2.leftClip + 2 * o2.rightClip);
        f.hasMatePair = true;
        f.o2 = o2;
      }
    } else {
      const Overlap &o = (*pov2)[fr.second];
      f.matchCnt = o.matchCnt; f.similarity = o.similarity; f.seqIdx = o.seqIdx;
      f.seqStart = o.seqStart; f.seqEnd = o.seqEnd; f.hasMatePair = false; f.hasN = hasN; f.o1FromR2 = true;
      f.relaxedMatchCnt = o.relaxedMatchCnt; f.o1 = o;
    }
    auto it = slotOfAllele.find(f.seqIdx);
    if (it != slotOfAllele.end()) { if (fragBefore(f, assign[it->second])) assign[it->second] = f; }
    else { slotOfAllele[f.seqIdx] = (int)assign.size(); assign.push_back(f); }
  }
  int bestMatch = -1, bestRelaxed = 0;  // 2474-2487
  double bestSim = 0;
  for (auto &f : assign)
    if (f.matchCnt > bestMatch || (f.matchCnt == bestMatch && f.similarity > bestSim)) { bestMatch = f.matchCnt; bestSim = f.similarity; bestRelaxed = f.relaxedMatchCnt; }
  size_t kx = 0;
  for (size_t i = 0; i < assign.size(); ++i) {  // 2488-2545
    int relax = 2;
    FragmentOverlap &f = assign[i];
    if (prm.relaxIntronAlign && f.hasMatePair && f.o1.seqIdx == f.o2.seqIdx &&
        ((f.o1.seqStart <= f.o2.seqStart && f.o1.seqEnd >= f.o2.seqStart) || (f.o2.seqStart <= f.o1.seqStart && f.o2.seqEnd >= f.o1.seqStart))) {
      if (f.o1.matchCnt < f.o1.relaxedMatchCnt && f.o2.matchCnt < f.o2.relaxedMatchCnt) relax = 4;
    }
    bool keep = (f.matchCnt == bestMatch && f.similarity == bestSim) ||
                (prm.relaxIntronAlign && f.matchCnt >= bestMatch - relax && f.relaxedMatchCnt == bestRelaxed);
    if (keep) { FragmentOverlap t = f; t.qual = 1; assign[kx++] = t; }
  }
  assign.resize(kx);
  if (!assign.empty() && pov2 && !assign[0].hasMatePair) {  // dangling rule (2553-2578)
    size_t i = 0;
    for (; i < assign.size(); ++i) {
      const FragmentOverlap &f = assign[i];
      if (f.similarity < 1 || separatorInRange(f.seqStart, f.seqEnd, f.seqIdx) ||
          (f.seqEnd - f.seqStart + 1 + f.o1.readEnd - f.o1.readStart + 1 < 3 * prm.hitLenRequired))
        break;
      const int spanRange = 100;
      if ((f.o1.strand == 1 && f.seqEnd + spanRange < (int)alleles[f.seqIdx].seq.size()) || (f.o1.strand == -1 && f.seqStart - spanRange >= 0)) break;
    }
    if (i < assign.size()) assign.clear();
  }
  if (!assign.empty() && pov2 && assign[0].hasMatePair) {  // truncated-reference rule (2580-2653)
    const FragmentOverlap rep = assign[0];
    auto truncated = [&](const Overlap &o, const Overlap &c1, const Overlap &c2) {  // TruncatedMatePairOverlap (502-523)
      if (o.seqIdx == -1 || c1.seqIdx == -1 || c2.seqIdx == -1) return false;
      if (o.strand == 1) {
        if ((int)alleles[o.seqIdx].seq.size() - 1 < o.seqEnd + c2.seqEnd - c1.seqEnd ||
            separatorInRange(o.seqEnd, o.seqEnd + c2.seqEnd - c1.seqEnd + 1, o.seqIdx))
          return true;
      } else if (o.strand == -1) {
        if (o.seqStart - (c1.seqStart - c2.seqStart) < 0 || separatorInRange(o.seqStart - (c1.seqStart - c2.seqStart) - 1, o.seqStart, o.seqIdx))
          return true;
      }
      return false;
    };
    bool filter = false;
    for (int i = 0; i < n1 && !filter; ++i) {
      const Overlap &o = ov1[i];
      if (o.matchCnt > rep.o1.matchCnt ||
          ((o.matchCnt == rep.o1.matchCnt && o.similarity > rep.o1.similarity) && slotOfAllele.find(o.seqIdx) == slotOfAllele.end())) {
        if (truncated(o, rep.o1, rep.o2)) filter = true;
        else if (o.similarity > rep.o2.similarity + 0.1) filter = true;
      }
    }
    for (int i = 0; i < (int)pov2->size() && !filter; ++i) {
      const Overlap &o = (*pov2)[i];
      if (o.matchCnt > rep.o2.matchCnt ||
          ((o.matchCnt == rep.o2.matchCnt && o.similarity > rep.o2.similarity) && slotOfAllele.find(o.seqIdx) == slotOfAllele.end())) {
        if (truncated(o, rep.o2, rep.o1)) filter = true;
        else if (o.similarity > rep.o1.similarity + 0.1) filter = true;
      }
    }
    if (filter) assign.clear();
  }
  return (int)assign.size();
}

// Genotyper::SetReadAssignments + ReadAssignmentWeight (Genotyper.hpp:778-832, 205-230); whitelist unused
void Oracle::fragmentToRow(const std::vector<FragmentOverlap> &frag, std::vector<RowEntry> &row) {
  row.clear();
  int n = (int)frag.size();
  if (prm.maxAssignCnt > 0 && n > prm.maxAssignCnt) return;
  for (auto &f : frag)
    if (separatorInRange(f.seqStart, f.seqEnd, f.seqIdx)) return;
  double adjust = 1.0, maxSim = 0;
  for (auto &f : frag) maxSim = std::max(maxSim, f.similarity);
  if (maxSim < 1) adjust = 0.25;
  for (auto &f : frag) {
    RowEntry e;
    e.alleleIdx = f.seqIdx; e.start = f.seqStart; e.end = f.seqEnd;
    double w = 1, seg = (1 - prm.refSeqSimilarity) / 4.0;
    if (seg < 0.01) seg = 0.01;
    if (f.similarity < 1 - 3 * seg) w = 0.01;
    else if (f.similarity < 1 - 2 * seg) w = 0.1;
    else if (f.similarity < 1 - seg) w = 0.5;
    if (f.hasN) w /= 10.0;
    e.weight = (float)w;
    e.qual = (float)f.qual;
    e.adjustWeight = (float)(adjust * e.weight);
    row.push_back(e);
  }
}

// Genotyper::CoalesceReadAssignments (Genotyper.hpp:841-908), one fragment row at a time, in fragment order
void Oracle::coalesceRow(std::vector<RowEntry> &row) {
  if (row.empty()) return;
  std::sort(row.begin(), row.end(), [](const RowEntry &a, const RowEntry &b) { return a.alleleIdx < b.alleleIdx; });
  std::vector<int> pat;
  for (auto &e : row) pat.push_back(e.alleleIdx);
  auto it = groupOfPattern.find(pat);
  if (it == groupOfPattern.end()) {
    groupOfPattern[pat] = (int)groups.size();
    groups.push_back(row);
    return;
  }
  std::vector<RowEntry> &g = groups[it->second];
  for (size_t j = 0; j < row.size(); ++j) {  // 887-897
    if (row[j].qual == 1) {
      if (row[j].start < g[j].start) g[j].start = row[j].start;
      if (row[j].end < g[j].end) g[j].end = row[j].start;
    }
    g[j].weight += row[j].weight;
    g[j].adjustWeight += row[j].adjustWeight;
  }
}

int Oracle::missingBaseCoverage(int seqIdx, double ratio) const {  // SeqSet::GetSeqMissingBaseCoverage (2717-2755)
  const AlleleRec &a = alleles[seqIdx];
  std::vector<int> c;
  for (size_t i = 0; i < a.seq.size(); ++i)
    if (a.exon[i]) {
      int b = baseCode(a.seq[i]);
      c.push_back(b >= 0 ? a.cov[i * 4 + b] : 0);
    }
  if (c.empty()) return 0;
  std::sort(c.begin(), c.end());
  double cutoff = c[c.size() / 2] * ratio;
  if (cutoff < 1) cutoff = 1;
  size_t i = 0;
  for (; i < c.size(); ++i)
    if (c[i] >= cutoff) break;
  return (int)i;
}

// Genotyper::FinalizeReadAssignments -> BuildAlleleEquivalentClass (Genotyper.hpp:912-939, 1072-1139)
void Oracle::finalizeGroups() {
  int A = (int)alleles.size(), G = (int)groups.size();
  groupsInAllele.assign(A, {});
  slotInAllele.assign(A, {});
  for (int g = 0; g < G; ++g)
    for (int j = 0; j < (int)groups[g].size(); ++j) {
      groupsInAllele[groups[g][j].alleleIdx].push_back(g);
      slotInAllele[groups[g][j].alleleIdx].push_back(j);
    }
  struct FP { int allele, fp; };
  std::vector<FP> fps;
  for (int i = 0; i < A; ++i) {
    int b = -1;
    alleles[i].ec = -1;
    if (!groupsInAllele[i].empty()) {
      b = 0;
      for (int g : groupsInAllele[i]) b = (int)(((uint32_t)b * (uint32_t)G + (uint32_t)g) % 1000003u);  // 1089
    }
    fps.push_back({i, b});
  }
  std::sort(fps.begin(), fps.end(), [](const FP &x, const FP &y) {  // CompSortPairByBDec
    if (x.fp != y.fp) return y.fp < x.fp;
    return x.allele < y.allele;
  });
  ecAlleles.clear();
  if (A == 0 || fps[0].fp == -1) return;
  for (int i = 0; i < A; ++i) {
    if (fps[i].fp == -1) break;
    int join = -1;
    for (int j = i - 1; j >= 0; --j) {
      if (fps[i].fp != fps[j].fp) break;
      if (groupsInAllele[fps[i].allele] == groupsInAllele[fps[j].allele]) { join = j; break; }  // qual is always 1
    }
    if (join < 0) {
      alleles[fps[i].allele].ec = (int)ecAlleles.size();
      ecAlleles.push_back({fps[i].allele});
    } else {
      int ec = alleles[fps[join].allele].ec;
      alleles[fps[i].allele].ec = ec;
      ecAlleles[ec].push_back(fps[i].allele);
    }
  }
  // RemoveLowMAPQAlleleInEquivalentClass (1330-1368) is the identity here: every qual is 1, and the members of a
  // class share their group list, hence their qual sums.
  for (int i = 0; i < A; ++i) alleles[i].missingCoverage = missingBaseCoverage(i, 0.01);
}

// Genotyper::SetAlleleAbundance (Genotyper.hpp:957-1014)
void Oracle::setAlleleAbundance(const std::vector<double> &n, std::vector<double> &majorAbund, std::vector<double> &geneMax) {
  for (auto &a : alleles) a.abundance = a.ecAbundance = 0;
  for (size_t i = 0; i < ecAlleles.size(); ++i) {
    double abund = 0;
    abund += n[i];
    abund = abund / ecLength[i] * 1000.0;
    int size = (int)ecAlleles[i].size();
    for (int al : ecAlleles[i]) { alleles[al].abundance = abund / size; alleles[al].ecAbundance = abund; }
  }
  majorAbund.assign(majorNames.size(), 0);
  geneMax.assign(geneNames.size(), 0);
  for (auto &a : alleles) majorAbund[a.majorAllele] += a.abundance;
  for (auto &a : alleles)
    if (majorAbund[a.majorAllele] > geneMax[a.gene]) geneMax[a.gene] = majorAbund[a.majorAllele];
}

// Genotyper::EMupdate (Genotyper.hpp:372-421)
double Oracle::emUpdate(const std::vector<double> &x0, std::vector<double> &x1, std::vector<double> &n,
                        const std::vector<std::vector<int>> &rows, const std::vector<double> &count) {
  size_t E = ecAlleles.size();
  std::fill(n.begin(), n.end(), 0.0);
  for (size_t g = 0; g < rows.size(); ++g) {
    double psum = 0;
    for (int ec : rows[g]) psum += x0[ec] * 1;
    if (psum == 0) psum = 1;
    for (int ec : rows[g]) n[ec] += count[g] * (x0[ec] * 1 / psum);
  }
  double diff = 0, norm = 0;
  for (size_t i = 0; i < E; ++i) norm += n[i] / ecLength[i];
  for (size_t i = 0; i < E; ++i) {
    double t = n[i] / ecLength[i] / norm;
    diff += std::fabs(t - x0[i]);
    x1[i] = t;
  }
  return diff;
}

// Genotyper::QuantifyAlleleEquivalentClass (Genotyper.hpp:1142-1328)
int Oracle::quantify(std::vector<double> *trajectory) {
  size_t E = ecAlleles.size(), G = groups.size();
  std::vector<double> count(G);
  std::vector<std::vector<int>> rows(G);
  for (size_t g = 0; g < G; ++g) {
    double c = groups[g][0].weight;
    for (size_t j = 1; j < groups[g].size(); ++j)
      if (groups[g][j].weight > c) c = groups[g][j].weight;
    count[g] = c;
    for (auto &e : groups[g]) {
      int ec = alleles[e.alleleIdx].ec;
      if (std::find(rows[g].begin(), rows[g].end(), ec) == rows[g].end()) rows[g].push_back(ec);
    }
  }
  ecLength.assign(E, 0);
  for (size_t i = 0; i < E; ++i) {
    int len = alleles[ecAlleles[i][0]].effectiveLen;
    for (int al : ecAlleles[i]) len = std::min(len, alleles[al].effectiveLen);
    ecLength[i] = len;
  }
  std::vector<double> x0(E), x1(E), x2(E), x3(E), n(E);
  for (size_t i = 0; i < E; ++i) {
    x0[i] = 0;
    for (int al : ecAlleles[i]) x0[i] += alleles[al].weight;  // 1226-1228
  }
  const int maxIt = 1000, maskRound = 10;
  int ret = 0;
  std::vector<double> majorAbund, geneMax;
  for (int t = 0; t < maxIt; ++t) {
    ++ret;
    emUpdate(x0, x1, n, rows, count);
    emUpdate(x1, x2, n, rows, count);
    double sr = 0, sv = 0;  // SQUAREMalpha (424-437)
    for (size_t i = 0; i < E; ++i) {
      sr += (x1[i] - x0[i]) * (x1[i] - x0[i]);
      sv += (x2[i] - 2 * x1[i] + x0[i]) * (x2[i] - 2 * x1[i] + x0[i]);
    }
    double alpha = sv == 0 ? -1 : -std::sqrt(sr) / std::sqrt(sv);
    if (prm.minSquaremAlpha < 0 && alpha < prm.minSquaremAlpha) alpha = prm.minSquaremAlpha;
    for (size_t i = 0; i < E; ++i)
      x3[i] = x0[i] - 2 * alpha * (x1[i] - x0[i]) + alpha * alpha * (x2[i] - 2 * x1[i] + x0[i]);
    emUpdate(x3, x1, n, rows, count);
    double diff = 0;
    for (size_t i = 0; i < E; ++i) { diff += std::fabs(x1[i] - x0[i]); x0[i] = x1[i]; }
    if (trajectory) { trajectory->insert(trajectory->end(), n.begin(), n.end()); trajectory->insert(trajectory->end(), x0.begin(), x0.end()); }
    if (diff < 1e-5 && t < maxIt - 2) t = maxIt - 2;
    if (t > 0 && t % maskRound == 0) {
      setAlleleAbundance(n, majorAbund, geneMax);
      for (auto &a : alleles)
        if (majorAbund[a.majorAllele] < prm.filterFrac * 0.5 * geneMax[a.gene]) { a.abundance = 0; a.ecAbundance = 0; }
      for (size_t i = 0; i < E; ++i) x0[i] = alleles[ecAlleles[i][0]].ecAbundance;
    }
  }
  setAlleleAbundance(n, majorAbund, geneMax);
  majorAlleleAbundance = majorAbund;
  geneMaxMajorAlleleAbundance = geneMax;
  ecReadCountFinal = n;
  ecAbundanceFinal = x0;
  return ret;
}

// ------------------------------------------------------------------------------------------------------------------
// After the EM: likelihood pruning inside the classes, allele selection, genotype quality, the two tables
// (SURVEY 8a rows 20-22).  Sequential restatements; read groups play the role of the reference's "reads".
// ------------------------------------------------------------------------------------------------------------------

// InitAlleleInfo, Genotyper.hpp:597-638: per gene the 31-mer profile (KmerCount.hpp:53-81: canonical codes, a window holding an N
// is skipped) of its lexicographically smallest consensus; similarity[i][j] = share of i's k-mer occurrences whose k-mer also
// occurs in j (KmerCount::GetCountSimilarity, KmerCount.hpp:196-216: not symmetric)
void Oracle::computeGeneSimilarity() {
  const int G = (int)geneNames.size(), K = 31;
  std::vector<std::map<uint64_t, int>> profile(G);
  for (int g = 0; g < G; ++g) {
    int pick = -1;
    for (int a = 0; a < (int)alleles.size(); ++a)
      if (alleles[a].gene == g && (pick == -1 || strcmp(alleles[a].seq.c_str(), alleles[pick].seq.c_str()) < 0)) pick = a;
    if (pick < 0) continue;
    const std::string &q = alleles[pick].seq;
    if ((int)q.size() < K) continue;
    const uint64_t mask = (1ull << (2 * K)) - 1;
    uint64_t code = 0;
    int sinceN = K;  // positions since the last N, saturating: the window is clean when >= K
    for (int i = 0; i < (int)q.size(); ++i) {
      code = ((code << 2) & mask) | (uint64_t)(baseCode(q[i]) & 3);
      sinceN = q[i] == 'N' ? 0 : std::min(sinceN + 1, K);
      if (i < K - 1 || sinceN < K) continue;
      uint64_t rc = 0;
      for (int b = 0; b < K; ++b) rc = (rc << 2) | (3ull - ((code >> (2 * b)) & 3ull));  // KmerCode::GetCanonicalKmerCode
      ++profile[g][rc < code ? rc : code];
    }
  }
  geneSimilarity.assign(G, std::vector<double>(G, 0.0));
  for (int i = 0; i < G; ++i)
    for (int j = 0; j < G; ++j) {
      if (i == j) { geneSimilarity[i][j] = 1.0; continue; }
      int total = 0, shared = 0;
      for (auto &kv : profile[i]) {
        total += kv.second;
        if (profile[j].count(kv.first)) shared += kv.second;
      }
      geneSimilarity[i][j] = (double)shared / (double)total;  // 0/0 = NaN for an empty profile, as in the reference
    }
}

void Oracle::removeLowLikelihoodAlleles() {
  for (auto &members : ecAlleles) {
    const int size = (int)members.size();
    if (size == 0) continue;
    std::vector<int> lo(size), hi(size, -1);
    std::map<int, int> slotOf;
    for (int j = 0; j < size; ++j) { lo[j] = (int)alleles[members[j]].seq.size(); slotOf[members[j]] = j; }
    // the groups of the class representative, and in each of them the entries of the class members (1398-1416)
    for (int g : groupsInAllele[members[0]])
      for (const RowEntry &e : groups[g]) {
        auto it = slotOf.find(e.alleleIdx);
        if (it == slotOf.end()) continue;
        if (e.start < lo[it->second]) lo[it->second] = e.start;
        if (e.end > hi[it->second]) hi[it->second] = e.end;
      }
    std::vector<double> ll(size);
    double best = -1;
    for (int j = 0; j < size; ++j) {
      const int len = (int)alleles[members[j]].seq.size();
      int span = hi[j] - lo[j] + 1;
      if (span > len) span = len;
      ll[j] = pow((double)span / len, alleles[members[j]].ecAbundance);
      if (ll[j] > best) best = ll[j];
    }
    std::vector<int> kept;
    for (int j = 0; j < size; ++j)
      if (ll[j] / best >= 0.05 || ll[j] == best) kept.push_back(members[j]);
    members = kept;
  }
}

int Oracle::geneAlleleTypes(int gene) const {
  if (selectedAlleles[gene].empty()) return 0;
  int top = 0;
  for (auto &s : selectedAlleles[gene]) top = std::max(top, s.second);
  return top + 1;
}

// upper / lower tail of the standard normal distribution: Hill, Algorithm AS 66, Applied Statistics 22(3) 1973, 424-427 -- the
// routine the reference carries as alnorm (Genotyper.hpp:252-370)
static double normalIntegralAS66(double x, bool upper) {
  const double ltone = 7.0, utzero = 18.66, con = 1.28;
  bool up = upper;
  double z = x;
  if (z < 0.0) { up = !up; z = -z; }
  if (ltone < z && (!up || utzero < z)) return up ? 0.0 : 1.0;
  const double y = 0.5 * z * z;
  double value;
  if (z <= con)
    value = 0.5 - z * (0.398942280444 - 0.39990348504 * y / (y + 5.75885480458 + -29.8213557807 / (y + 2.62433121679 + 48.6959930692 / (y + 5.92885724438))));
  else
    value = 0.398942280385 * exp(-y) /
            (z + -0.000000038052 +
             1.00000615302 / (z + 0.000398064794 + 1.98615381364 / (z + -0.151679116635 + 5.29330324926 / (z + 4.8385912808 + -15.1508972451 / (z + 0.742380924027 + 30.789933034 / (z + 3.99019417011))))));
  return up ? value : 1.0 - value;
}

void Oracle::selectAllelesForGenes() {
  const int nGenes = (int)geneNames.size(), nGroups = (int)groups.size(), nEc = (int)ecAlleles.size();
  const double frac = prm.filterFrac;
  if (geneSimilarity.empty()) computeGeneSimilarity();
  for (auto &a : alleles) { a.genotypeQuality = -1; a.alleleRank = -1; }
  selectedAlleles.assign(nGenes, {});
  std::vector<char> covered(nGroups, 0);
  auto optimal = [&](int allele, int r) { return groups[groupsInAllele[allele][r]][slotInAllele[allele][r]].qual == 1; };  // IsReadsInAlleleIdxOptimal (198-203)
  auto weak = [&](int a) {  // the abundance filter of 1568-1570, applied again at 1656-1658
    const AlleleRec &m = alleles[a];
    return m.ecAbundance < frac * geneMaxMajorAlleleAbundance[m.gene] &&
           (m.ecAbundance * 3 >= majorAlleleAbundance[m.majorAllele] || majorAlleleAbundance[m.majorAllele] < 3 * frac * geneMaxMajorAlleleAbundance[m.gene]);
  };
  // classes by abundance, descending; ties by class id (CompSortPairIntDoubleBDec)
  std::vector<std::pair<int, double>> order;
  for (int e = 0; e < nEc; ++e) order.push_back({e, alleles[ecAlleles[e][0]].ecAbundance});
  std::sort(order.begin(), order.end(), [](const std::pair<int, double> &p, const std::pair<int, double> &q) { return p.second != q.second ? q.second < p.second : p.first < q.first; });
  std::vector<int> filtered;
  for (int i = 0; i < nEc; ++i) {
    const std::vector<int> &members = ecAlleles[order[i].first];
    const int lead = members[0];
    if (alleles[lead].ecAbundance <= 1e-6) break;
    double coveredWeight = 0, totalWeight = 0;  // of the lead's groups: first-entry weights (1529-1539)
    const int nLead = (int)groupsInAllele[lead].size();
    for (int r = 0; r < nLead; ++r) {
      if (!optimal(lead, r)) continue;
      const int g = groupsInAllele[lead][r];
      const double w = groups[g][0].weight;
      if (covered[g]) coveredWeight += w;
      totalWeight += w;
    }
    std::vector<int> genesToAdd, toAdd;
    for (int a : members) {
      const AlleleRec &m = alleles[a];
      bool filter = weak(a);
      if (coveredWeight == totalWeight &&
          (m.ecAbundance < 0.25 * geneMaxMajorAlleleAbundance[m.gene] || selectedAlleles[m.gene].empty() ||
           m.ecAbundance < 0.5 * alleles[selectedAlleles[m.gene].back().first].ecAbundance))
        filter = true;
      if (filter) { filtered.push_back(a); continue; }
      if (std::find(genesToAdd.begin(), genesToAdd.end(), m.gene) == genesToAdd.end()) genesToAdd.push_back(m.gene);
      toAdd.push_back(a);
    }
    const int quality = genesToAdd.size() > 1 ? 0 : 60;
    if (!genesToAdd.empty())
      for (int r = 0; r < nLead; ++r)
        if (optimal(lead, r)) covered[groupsInAllele[lead][r]] = 1;
    std::map<int, int> newRankOfGene;
    for (int a : toAdd) {
      AlleleRec &m = alleles[a];
      int rank = -1;
      for (auto &s : selectedAlleles[m.gene])
        if (alleles[s.first].majorAllele == m.majorAllele) { rank = s.second; break; }
      if (rank == -1) {
        auto it = newRankOfGene.find(m.gene);
        if (it != newRankOfGene.end()) rank = it->second;
        else { rank = geneAlleleTypes(m.gene); newRankOfGene[m.gene] = rank; }
      }
      m.genotypeQuality = weak(a) ? 0 : quality;
      m.alleleRank = rank;
      selectedAlleles[m.gene].push_back({a, rank});
    }
  }
  // filtered alleles whose major allele was selected after all join it (1669-1695)
  for (int a : filtered) {
    const AlleleRec &m = alleles[a];
    int rank = -1;
    for (auto &s : selectedAlleles[m.gene])
      if (alleles[s.first].majorAllele == m.majorAllele) { rank = s.second; break; }
    if (rank != -1) selectedAlleles[m.gene].push_back({a, rank});
  }
  // genes with more than two allele types: the pair of types that explains the most groups not explained elsewhere (1697-1996)
  std::vector<int> use(nGroups, 0);
  auto countTopTwo = [&](int gene, std::map<int, int> &usedEc, int delta) {
    for (auto &s : selectedAlleles[gene]) {
      if (s.second > 1) continue;
      if (usedEc.count(alleles[s.first].ec)) continue;
      usedEc[alleles[s.first].ec] = 1;
      for (int r = 0; r < (int)groupsInAllele[s.first].size(); ++r)
        if (optimal(s.first, r)) use[groupsInAllele[s.first][r]] += delta;
    }
  };
  {
    std::map<int, int> usedEc;  // one map over all genes (1705-1729)
    for (int g = 0; g < nGenes; ++g) countTopTwo(g, usedEc, +1);
  }
  std::vector<std::map<int, double>> missingWeight(nGenes);  // per gene: missing coverage -> largest type abundance with it (1733-1770)
  for (int g = 0; g < nGenes; ++g) {
    const int T = geneAlleleTypes(g);
    std::vector<int> miss(T, -1);
    std::vector<double> ab(T, 0.0);
    for (auto &s : selectedAlleles[g]) {
      ab[s.second] += alleles[s.first].abundance;
      if (miss[s.second] == -1 || alleles[s.first].missingCoverage < miss[s.second]) miss[s.second] = alleles[s.first].missingCoverage;
    }
    for (int t = 0; t < T; ++t)
      if (!missingWeight[g].count(miss[t]) || missingWeight[g][miss[t]] < ab[t]) missingWeight[g][miss[t]] = ab[t];
  }
  for (int iter = 0; iter < 1000; ++iter) {
    int updated = 0;
    for (int g = 0; g < nGenes; ++g) {
      const int T = geneAlleleTypes(g);
      if (T <= 2) continue;
      std::vector<std::pair<int, int>> &sel = selectedAlleles[g];
      const int S = (int)sel.size();
      std::map<int, int> usedEc;
      countTopTwo(g, usedEc, -1);
      double maxCover = 0, maxCoverAbundance = 0;
      std::vector<std::pair<int, int>> bestTypes;
      int alleleJ = 0;
      for (int j = 0; j < T - 1 && j <= 1; ++j) {
        usedEc.clear();
        std::map<int, int> fromJ;
        for (int l = 0; l < S; ++l) {
          if (sel[l].second != j) continue;
          const int a = sel[l].first;
          if (usedEc.count(alleles[a].ec)) continue;
          usedEc[alleles[a].ec] = 1;
          for (int r = 0; r < (int)groupsInAllele[a].size(); ++r)
            if (use[groupsInAllele[a][r]] == 0 && optimal(a, r)) fromJ[groupsInAllele[a][r]] |= 1;
          alleleJ = l;
        }
        for (int k = j + 1; k < T; ++k) {
          std::map<int, int> cover = fromJ;
          for (int l = 0; l < S; ++l) {  // usedEc is not cleared between the k's
            if (sel[l].second != k) continue;
            const int a = sel[l].first;
            if (usedEc.count(alleles[a].ec)) continue;
            usedEc[alleles[a].ec] = 1;
            for (int r = 0; r < (int)groupsInAllele[a].size(); ++r)
              if (use[groupsInAllele[a][r]] == 0 && optimal(a, r)) cover[groupsInAllele[a][r]] |= 2;
          }
          double abJ = 0, abK = 0;
          int missJ = -1, missK = -1;
          for (int l = 0; l < S; ++l) {
            const AlleleRec &m = alleles[sel[l].first];
            if (sel[l].second == j) { abJ += m.abundance; if (missJ == -1 || m.missingCoverage < missJ) missJ = m.missingCoverage; }
            else if (sel[l].second == k) { abK += m.abundance; if (missK == -1 || m.missingCoverage < missK) missK = m.missingCoverage; }
          }
          const double product = abJ * abK;
          double score = 0;
          for (auto &kv : cover) score += groups[kv.first][0].adjustWeight;
          if (T > 3 || missJ >= 10 || missK >= 10) {
            double wJ = missingWeight[g][missJ], wK = missingWeight[g][missK];
            if (T <= 3) {
              if (wJ >= 1) wJ = log(wJ) / log(10.0);
              if (wK >= 1) wK = log(wK) / log(10.0);
            }
            score = score - missJ * wJ * readLength / 150.0 - missK * wK * readLength / 150.0 + (alleles[sel[alleleJ].first].weight);
          }
          if (bestTypes.empty() || score > maxCover || (score == maxCover && product > maxCoverAbundance)) {
            maxCover = score; maxCoverAbundance = product;
            bestTypes.clear();
            bestTypes.push_back({j, k});
          } else if (score == maxCover) bestTypes.push_back({j, k});
        }
      }
      const std::pair<int, int> best = bestTypes[0];
      if (best.first != 0 || best.second != 1) {
        ++updated;
        for (auto &s : sel) {
          int r;
          if (s.second == best.first) r = 0;
          else if (s.second == best.second) r = 1;
          else if (s.second < best.first) r = s.second + 2;
          else if (s.second < best.second) r = s.second + 1;
          else continue;
          s.second = r;
          alleles[s.first].alleleRank = r;
        }
      }
      usedEc.clear();
      countTopTwo(g, usedEc, +1);
    }
    if (!updated) break;
  }
  // genotype quality (2010-2085)
  std::vector<double> geneAbundance(nGenes, 0.0);
  for (int g = 0; g < nGenes; ++g)
    for (auto &s : selectedAlleles[g]) geneAbundance[g] += alleles[s.first].abundance;
  const double crossAlleleRate = 0.01;
  for (int g = 0; g < nGenes; ++g) {
    const int T = geneAlleleTypes(g);
    std::vector<double> rankAbundance(T, 0.0);
    for (auto &s : selectedAlleles[g]) rankAbundance[s.second] += alleles[s.first].abundance;
    double noise = 0;
    for (int o = 0; o < nGenes; ++o)
      if (o != g) noise += prm.crossGeneRate * geneSimilarity[o][g] * geneAbundance[o];
    for (int t = 0; t < T; ++t) {
      const double nullMean = (geneAbundance[g] - rankAbundance[t]) * crossAlleleRate + noise;
      double score = 0;
      if (rankAbundance[t]) score = -log(normalIntegralAS66(2 * (sqrt(rankAbundance[t]) - sqrt(nullMean)), true)) / log(double(10.0));
      if (score > 60) score = 60;
      if (score < 0) score = 0;
      if (rankAbundance[t] < prm.filterCov) score = 0;
      for (auto &s : selectedAlleles[g])
        if (s.second == t && alleles[s.first].genotypeQuality > 0) alleles[s.first].genotypeQuality = (int)score;
    }
  }
}

// one line per gene: name, number of called alleles, allele 1, allele 2, further types
std::string Oracle::genotypeText() const {
  std::string out;
  char num[64];
  for (int g = 0; g < (int)geneNames.size(); ++g) {
    std::vector<char> used(majorNames.size(), 0);
    std::string field[3];
    int called = 0, qualities[2] = {-1, -1};
    const int T = std::max(2, geneAlleleTypes(g));
    char sep = '\t';
    for (int type = 0; type < T; ++type) {
      std::string &buf = field[type > 1 ? 2 : type];
      if (type > 1) sep = ';';
      buf.clear();  // 2132: also for every type beyond the second, so the third column shows the last type only
      double abundance = 0;
      bool added = false;
      int localQual = -1;
      if (type == 1 && qualities[0] == 0) std::fill(used.begin(), used.end(), 0);
      for (auto &s : selectedAlleles[g]) {
        if (s.second != type) continue;
        const AlleleRec &m = alleles[s.first];
        abundance += m.abundance;
        if (used[m.majorAllele]) continue;
        localQual = m.genotypeQuality;
        if (type <= 1) called = type + 1;
        buf += (added ? "," : "") + majorNames[m.majorAllele];  // (the "|" branch of 2159-2160 needs a non-empty buffer: unreachable)
        added = true;
        used[m.majorAllele] = 1;
      }
      if (localQual >= 0) { snprintf(num, sizeof num, "%c%lf%c%d", sep, abundance, sep, localQual); buf += num; }
      else if (type <= 1) buf += ".\t0\t-1";
      if (type <= 1) qualities[type] = localQual;
    }
    snprintf(num, sizeof num, "\t%d", called);
    out += geneNames[g] + num + "\t" + field[0] + "\t" + field[1] + "\t" + field[2] + "\n";
  }
  return out;
}

void Oracle::parseAlleleNameExon(const std::string &allele, std::string &gene, std::string &major) const {
  // Genotyper::ParseAlleleName (Genotyper.hpp:63-131) with fieldsType = 1: five digits without a delimiter, three fields with one
  int parseType = 1, fields = prm.alleleDigitUnits;
  char delim = 0;
  if (fields == -1) {
    if (allele.find(':') != std::string::npos) { delim = ':'; parseType = 2; }
    fields = parseType == 1 ? 5 : 3;
  }
  if (prm.alleleDelimiter != 0) { delim = prm.alleleDelimiter; parseType = 2; }
  const size_t star = allele.find('*');
  const size_t i = star == std::string::npos ? allele.size() : star;
  gene = allele.substr(0, i);
  if (parseType == 1) {
    size_t j = 0;
    while ((int)j <= fields && i + j < allele.size()) ++j;
    major = allele.substr(0, i + j);
  } else {
    int k = 0;
    size_t j = i;
    for (; j < allele.size(); ++j)
      if (allele[j] == delim) { ++k; if (k >= fields) break; }
    major = allele.substr(0, j);
  }
}

// per gene up to two representative alleles "name quality"
std::string Oracle::alleleText() const {
  std::string out;
  for (int g = 0; g < (int)geneNames.size(); ++g) {
    int rep[2] = {-1, -1};
    for (auto &s : selectedAlleles[g]) {
      const int tag = s.second, a = s.first;
      if (tag > 1 || alleles[a].genotypeQuality < 1) continue;
      if (rep[tag] == -1 || alleles[rep[tag]].ecAbundance < alleles[a].ecAbundance || (alleles[rep[tag]].ecAbundance == alleles[a].ecAbundance && rep[tag] > a)) rep[tag] = a;
    }
    if (rep[1] == -1 && rep[0] != -1) {  // two alleles of one major allele: the strongest other class of type 0 that differs in the exons (2200-2222)
      double top = -1;
      int pick = -1;
      for (auto &s : selectedAlleles[g]) {
        const int a = s.first;
        if (s.second != 0 || alleles[a].ec == alleles[rep[0]].ec) continue;
        std::string gA, xA, gB, xB;
        parseAlleleNameExon(alleles[a].name, gA, xA);
        parseAlleleNameExon(alleles[rep[0]].name, gB, xB);
        if (xA == xB) continue;
        if (alleles[a].ecAbundance > top || (alleles[a].ecAbundance == top && a < pick)) { top = alleles[a].ecAbundance; pick = a; }
      }
      if (top != -1) rep[1] = pick;
    }
    for (int j = 0; j < 2; ++j)
      if (rep[j] != -1) out += alleles[rep[j]].name + " " + std::to_string(alleles[rep[j]].genotypeQuality) + "\n";
  }
  return out;
}

}  // namespace t1k_oracle
