// tests/harness/variants_san.cpp -- TEST INFRASTRUCTURE.  The host code of the analyzer's variant calling (t1k_amd/csrc/host/variants.cpp)
// built with -fsanitize=address,undefined and run on a dumped case (tests/test_variants_host.py writes it): every array it walks -- cell
// tables, edit strings, reads seen reverse-complemented, candidate lists -- under the sanitizers' eyes; prints the VCF text and one line of
// kept flags per fragment, which the test compares with what the library computed.
//   variants_san <case file> <varMaxGroup>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../t1k_amd/csrc/host/t1k_host.h"

using namespace t1k;

static void readOverlap(std::istream &in, t1k_overlap &o, std::vector<int8_t> &ops, uint64_t &at, uint32_t &n) {
  std::string e;
  in >> o.seq_idx >> o.read_start >> o.read_end >> o.seq_start >> o.seq_end >> o.strand >> o.match_cnt >> o.left_clip >> o.right_clip >> o.relaxed_match_cnt >> o.similarity >> e;
  at = ops.size();
  n = 0;
  if (e != "-") { for (char c : e) ops.push_back((int8_t)(c - '0')); n = (uint32_t)e.size(); }
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  std::ifstream in(argv[1]);
  if (!in) return 2;
  RefSet ref;
  size_t A;
  in >> A;
  std::vector<double> abundance(A);
  for (size_t a = 0; a < A; ++a) {
    AlleleMeta m;
    std::string seq, mask;
    in >> m.name >> m.gene >> abundance[a] >> seq >> mask;
    m.seqLen = (int)seq.size();
    ref.al.push_back(m);
    ref.seqs.push_back(seq);
    std::vector<uint8_t> ex(mask.size());
    for (size_t i = 0; i < mask.size(); ++i) ex[i] = mask[i] == '1';
    ref.exon.push_back(ex);
  }
  size_t F;
  int paired;
  in >> F >> paired;
  // reads in heap blocks of exactly their size (no terminator): a step past either end is seen
  std::vector<std::unique_ptr<char[]>> text;
  std::vector<uint32_t> len[2];
  std::vector<const char *> ptr[2];
  std::vector<uint64_t> asgPtr(F + 1, 0);
  std::vector<t1k_frag_assignment> asg;
  std::vector<int8_t> ops;
  for (size_t f = 0; f < F; ++f) {
    size_t n;
    in >> n;
    for (int m = 0; m < (paired ? 2 : 1); ++m) {
      std::string r;
      in >> r;
      if (r == "-") r.clear();
      text.emplace_back(new char[r.size() ? r.size() : 1]);
      memcpy(text.back().get(), r.data(), r.size());
      ptr[m].push_back(text.back().get());
      len[m].push_back((uint32_t)r.size());
    }
    for (size_t i = 0; i < n; ++i) {
      t1k_frag_assignment a;
      memset(&a, 0, sizeof a);
      in >> a.allele_idx >> a.has_mate_pair >> a.o1_from_r2;
      readOverlap(in, a.o1, ops, a.ops1, a.n_ops1);
      if (a.has_mate_pair) readOverlap(in, a.o2, ops, a.ops2, a.n_ops2);
      asg.push_back(a);
    }
    asgPtr[f + 1] = asg.size();
  }
  if (!in) { fprintf(stderr, "case file cut short\n"); return 2; }
  // the edit strings too: one heap block of exactly their size
  std::unique_ptr<int8_t[]> opsBlock(new int8_t[ops.size() ? ops.size() : 1]);
  memcpy(opsBlock.get(), ops.data(), ops.size());
  std::vector<VariantCaller::Fragment> frags(F);
  for (size_t f = 0; f < F; ++f) {
    frags[f].asg = asg.data() + asgPtr[f];
    frags[f].n = (uint32_t)(asgPtr[f + 1] - asgPtr[f]);
    frags[f].r1 = ptr[0][f]; frags[f].l1 = len[0][f];
    if (paired) { frags[f].r2 = ptr[1][f]; frags[f].l2 = len[1][f]; }
  }
  VariantCaller vc(ref, abundance, atoi(argv[2]));
  vc.compute(frags, opsBlock.get());
  std::string vcf = vc.vcfText();
  printf("%zu\n%s", vc.variants.size(), vcf.c_str());
  std::vector<uint8_t> keep;
  for (size_t f = 0; f < F; ++f) {
    keep.assign(frags[f].n, 0);
    vc.adjust(frags[f], opsBlock.get(), keep.data());
    std::string line;
    for (uint8_t k : keep) line += k ? '1' : '0';
    printf("%s\n", line.empty() ? "-" : line.c_str());
  }
  return 0;
}
