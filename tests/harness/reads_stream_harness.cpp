// tests/harness/reads_stream_harness.cpp -- TEST INFRASTRUCTURE (CPU): the read-file index of the host side (t1k_amd/csrc/host/reads.cpp) opened
// whole and opened as a stream (ReadInput::openStreaming: the .gz files inflated by host/inflate.cpp while the records are indexed behind
// the decoder), consumed the way the job's window loop does: records are read as soon as they are published, in pieces.
//   reads_stream_harness <out prefix> <files per mate> files of mate 1 ... [files of mate 2 ...]
// writes <prefix>_whole.tsv and <prefix>_stream.tsv (id1, seq1[, id2, seq2] per fragment) and prints what happened.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../t1k_amd/csrc/host/t1k_host.h"
using t1k::ReadInput;
static void line(FILE *f, const ReadInput &in, size_t i) {
  const uint32_t r = in.frag[i];
  fprintf(f, "%.*s\t%.*s", (int)in.side[0].idL[r], in.side[0].idP[r], (int)in.side[0].seqL[r], in.side[0].seqP[r]);
  if (in.paired) fprintf(f, "\t%.*s\t%.*s", (int)in.side[1].idL[r], in.side[1].idP[r], (int)in.side[1].seqL[r], in.side[1].seqP[r]);
  if (in.hasBarcode) fprintf(f, "\t%.*s", (int)in.bc.seqL[r], in.bc.seqP[r]);
  fputc('\n', f);
}
int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const std::string out = argv[1];
  const int per = atoi(argv[2]);
  std::vector<std::string> f1, f2;
  for (int i = 3; i < argc; ++i) ((int)f1.size() < per ? f1 : f2).push_back(argv[i]);
  std::string err;
  {
    ReadInput whole;
    const char *bcEnv = getenv("HARNESS_BARCODE");
    if (!whole.open(f1, f2, bcEnv ? bcEnv : "", 4, err)) { printf("whole: ERROR %s\n", err.c_str()); }
    else {
      FILE *f = fopen((out + "_whole.tsv").c_str(), "w");
      for (size_t i = 0; i < whole.nFrag(); ++i) line(f, whole, i);
      fclose(f);
      printf("whole: %zu fragments, longest read %d\n", whole.nFrag(), whole.maxLen);
    }
  }
  ReadInput st;
  err.clear();
  const char *bcEnv = getenv("HARNESS_BARCODE");
  const std::string bcFile = bcEnv ? bcEnv : "";
  if (!st.openStreaming(f1, f2, bcFile, err)) { printf("stream: %s\n", err.empty() ? "not eligible" : ("ERROR " + err).c_str()); return err.empty() ? 0 : 1; }
  const size_t bound = st.nFrag();
  FILE *f = fopen((out + "_stream.tsv").c_str(), "w");
  size_t done = 0, pieces = 0;
  for (;;) {  // the consumer: whatever has been published, at once
    const int state = st.streamState();
    const size_t have = st.streamAvail();
    for (; done < have; ++done) line(f, st, done);
    if (have > 0) ++pieces;
    if (state != 0 && done >= st.streamAvail()) break;
    st.streamWait(done + 1000);
  }
  fclose(f);
  if (!st.streamFinish(err)) { printf("stream: ERROR %s%s\n", err.c_str(), st.streamGaveUp.load() ? " [gave up: the job opens the files whole]" : ""); return 1; }
  if (st.nFrag() != done) { printf("stream: ERROR %zu records read, %zu in the trimmed tables\n", done, st.nFrag()); return 1; }
  printf("stream: %zu fragments, longest read %d, tables sized for %zu, read in %zu pieces\n", st.nFrag(), st.maxLen, bound, pieces);
  return 0;
}
