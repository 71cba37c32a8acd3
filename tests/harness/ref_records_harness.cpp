// tests/harness/ref_records_harness.cpp -- TEST INFRASTRUCTURE (CPU): the records of an allele reference as the product's RefSet::load reads
// them (t1k_amd/csrc/host/refset.cpp: readReferenceRecords -- plain '>' records parsed by all host threads, anything else by the general
// reader), one line "id TAB seq TAB comment TAB has-comment" each, to be compared with oracle/_ref/reads_harness -c (the reference's
// SeqSet::InputRefFa reads its FASTA through the same ReadFiles::Next, SeqSet.hpp:872-904).
#include <cstdio>
#include "../../t1k_amd/csrc/host/t1k_host.h"
int main(int argc, char **argv) {
  for (int i = 1; i < argc; ++i) {
    std::vector<t1k::SeqRec> recs;
    std::string err;
    if (!t1k::readReferenceRecords(argv[i], recs, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    for (auto &r : recs) {
      fwrite(r.id.data(), 1, r.id.size(), stdout); fputc('\t', stdout);
      fwrite(r.seq.data(), 1, r.seq.size(), stdout); fputc('\t', stdout);
      fwrite(r.comment.data(), 1, r.comment.size(), stdout);
      printf("\t%d\n", r.hasComment ? 1 : 0);
    }
  }
  return 0;
}
