// oracle/ref_reads_harness.cpp -- TEST INFRASTRUCTURE ONLY.  Tiny driver around the REFERENCE's own read-file reader (ReadFiles.hpp + the
// kseq.h it vendors; compiled with -I/root/reference by oracle/Makefile `ref`, output oracle/_ref/reads_harness; no reference source is
// copied): opens the files named on the command line the way the genotyper does (Genotyper.cpp:282-296: AddReadFile per -u / -1 file) and
// prints one line "id<TAB>seq" per record ReadFiles::Next hands out, so that the product's own indexer (t1k_amd/csrc/host/reads.cpp) can be
// compared with the reference's reader on odd and damaged files without a GPU.
#include <cstdio>
#include "ReadFiles.hpp"

#include <cstring>

int main(int argc, char **argv) {
  ReadFiles reads;
  bool comments = false;  // -c: a third column, the header's comment as SeqSet::InputRefFa sees it ("" and a fourth column 0 when there is none)
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-c")) comments = true;
    else reads.AddReadFile(argv[i], false);
  }
  while (reads.Next()) {
    if (comments) printf("%s\t%s\t%s\t%d\n", reads.id, reads.seq, reads.comment ? reads.comment : "", reads.comment ? 1 : 0);
    else printf("%s\t%s\n", reads.id, reads.seq);
  }
  return 0;
}
