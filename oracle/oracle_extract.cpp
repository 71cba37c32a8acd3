// oracle/oracle_extract.cpp -- TEST INFRASTRUCTURE ONLY.
// CPU restatement of the candidate-read test of the reference's fastq-extractor:
// IsLowComplexity / IsGoodCandidate (FastqExtractor.cpp:89-118) and SeqSet::HasHitInSet (SeqSet.hpp:1915-1990), on top of the
// seeding and hit-chaining restatements the genotyper path already uses (seedHits = GetHitsFromRead 1071-1229,
// candidatesFromHits = GetOverlapsFromHits 1232-1556 with filter 0).
// Parity status: PINNED against oracle/_ref/fastq-extractor (the reference's own FastqExtractor.cpp built by oracle/Makefile):
// tests/test_extract_oracle.py compares the extracted files byte for byte.
#include <algorithm>
#include <cstdlib>
#include <map>

#include "oracle_core.hpp"

namespace t1k_oracle {

int Oracle::loadReferenceFa(const std::string &fasta) {
  prm.nBaseCode = 0;  // this program's nucToNum maps 'N' to 0 (FastqExtractor.cpp:51-54)
  std::vector<SeqRecord> recs;
  if (!readAllRecords(fasta, recs)) return -1;
  for (auto &r : recs) {
    AlleleRec a;
    a.name = r.id;
    a.seq = r.seq;
    alleles.push_back(std::move(a));
  }
  buildIndex();
  return (int)alleles.size();
}

int Oracle::inferKmerLength() const {
  int total = 0;  // an int in the reference as well
  for (auto &a : alleles) total += (int)a.seq.size();
  int ret = 0;
  while (total) { ++ret; total /= 4; }
  return ret + 1;
}

void Oracle::setKmerLength(int k) {
  prm.k = k;
  buildIndex();
}

bool Oracle::isLowComplexityRead(const std::string &read) {
  int cnt[5] = {0, 0, 0, 0, 0};
  for (char c : read) {
    if (c == 'N') ++cnt[4];
    else ++cnt[c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3];
  }
  const int n = (int)read.size();
  if (cnt[0] >= n / 2 || cnt[1] >= n / 2 || cnt[2] >= n / 2 || cnt[3] >= n / 2 || cnt[4] >= n / 10) return true;
  int low = 0;
  for (int i = 0; i < 4; ++i) if (cnt[i] <= 2) ++low;
  return low >= 2;
}

bool Oracle::hasHitInSet(const std::string &read) {
  const int k = prm.k, len = (int)read.size();
  if (len < k) return false;
  std::vector<int> strand, readOff;
  std::vector<Posting> post;
  seedHits(read, strand, readOff, post);
  if (strand.empty()) return false;
  // the (strand, sequence) bucket with the most hits; minus strand first, then sequence order, first maximum wins (1934-1957)
  std::map<std::pair<int, uint32_t>, int> cnt;
  for (size_t i = 0; i < strand.size(); ++i) ++cnt[{strand[i] == 1 ? 1 : 0, post[i].allele}];
  int best = -1;
  std::pair<int, uint32_t> bestKey{0, 0};
  for (auto &kv : cnt) if (kv.second > best) { best = kv.second; bestKey = kv.first; }
  if (k * best < prm.hitLenRequired) return false;  // 1959
  std::vector<int> s2, r2;
  std::vector<Posting> p2;
  for (size_t i = 0; i < strand.size(); ++i)
    if ((strand[i] == 1 ? 1 : 0) == bestKey.first && post[i].allele == bestKey.second) { s2.push_back(strand[i]); r2.push_back(readOff[i]); p2.push_back(post[i]); }
  std::vector<Cand> cands;
  candidatesFromHits(s2, r2, p2, cands);
  const int mismatchThreshold = (int)(len * (1 - prm.refSeqSimilarity)) * k;  // 1974
  for (auto &c : cands)
    if (len - c.o.matchCnt / 2 <= mismatchThreshold) return true;
  return false;
}

}  // namespace t1k_oracle
