// tests/harness/reads_shard_harness.cpp -- TEST INFRASTRUCTURE (CPU): the read-file index of the host side (t1k_amd/csrc/host/reads.cpp)
// opened whole, as a single process does, and opened by N ranks that each index only their own fragments (ReadInput::openSharded),
// with the ranks as threads of this process and their all-gather as a shared buffer.  Prints one line per fragment
// (id1, seq1, id2, seq2) for both, so that the test can compare the concatenation of the ranks' slices with the whole index.
//   reads_shard_harness <nRanks> <threads> <out prefix> <n files per mate> files1... [files2...]
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include "../../t1k_amd/csrc/host/t1k_host.h"

using t1k::ReadInput;

static void dump(FILE *f, const ReadInput &in) {
  for (size_t i = 0; i < in.nFrag(); ++i) {
    const uint32_t r = in.frag[i];
    fprintf(f, "%.*s\t%.*s", (int)in.side[0].idL[r], in.side[0].idP[r], (int)in.side[0].seqL[r], in.side[0].seqP[r]);
    if (in.paired) fprintf(f, "\t%.*s\t%.*s", (int)in.side[1].idL[r], in.side[1].idP[r], (int)in.side[1].seqL[r], in.side[1].seqP[r]);
    fputc('\n', f);
  }
}

// all-gather among threads: every rank copies its piece into a shared buffer, the last one to arrive publishes it
struct Gather {
  std::mutex m;
  std::condition_variable cv;
  std::vector<char> shared;
  int arrived = 0, round = 0, n = 1;
  bool run(int rank, void *buf, const uint64_t *bytes, const uint64_t *displ, uint64_t total) {
    std::unique_lock<std::mutex> g(m);
    if (arrived == 0) shared.assign(total, 0);
    memcpy(shared.data() + displ[rank], (const char *)buf + displ[rank], bytes[rank]);
    const int myRound = round;
    if (++arrived == n) { arrived = 0; ++round; cv.notify_all(); }
    else cv.wait(g, [&] { return round != myRound; });
    memcpy(buf, shared.data(), total);
    // nobody may start the next round (and clear `shared`) before everyone has copied this one out
    static int left = 0;
    if (++left == n) { left = 0; ++round; cv.notify_all(); }
    else { const int r2 = round; cv.wait(g, [&] { return round != r2; }); }
    return true;
  }
};

int main(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "usage: reads_shard_harness nRanks threads outPrefix nFilesPerMate files1... [files2...]\n"); return 2; }
  const int N = atoi(argv[1]), T = atoi(argv[2]);
  const std::string out = argv[3];
  const int per = atoi(argv[4]);
  std::vector<std::string> f1, f2;
  for (int i = 5; i < argc; ++i) ((int)f1.size() < per ? f1 : f2).push_back(argv[i]);
  std::string err;
  {
    ReadInput whole;
    if (!whole.open(f1, f2, "", T, err)) { fprintf(stderr, "open: %s\n", err.c_str()); return 1; }
    FILE *f = fopen((out + "_whole.tsv").c_str(), "w");
    dump(f, whole);
    fclose(f);
    fprintf(stderr, "whole: %zu fragments, longest read %d\n", whole.nFrag(), whole.maxLen);
  }
  Gather G;
  G.n = N;
  std::vector<std::unique_ptr<ReadInput>> part(N);
  std::vector<int> rc(N, 0);
  std::vector<std::string> errs(N);
  std::vector<std::thread> th;
  for (int r = 0; r < N; ++r)
    th.emplace_back([&, r] {
      part[r].reset(new ReadInput);
      ReadInput::ShardComm c;
      c.rank = r; c.nRanks = N;
      c.allgatherv = [&, r](void *buf, const uint64_t *bytes, const uint64_t *displ, uint64_t total) { return G.run(r, buf, bytes, displ, total); };
      rc[r] = part[r]->openSharded(f1, f2, T, c, errs[r]);
    });
  for (auto &t : th) t.join();
  FILE *f = fopen((out + "_sharded.tsv").c_str(), "w");
  size_t expectBase = 0;
  for (int r = 0; r < N; ++r) {
    if (rc[r] != 1) { fprintf(stderr, "rank %d: openSharded returned %d (%s)\n", r, rc[r], errs[r].c_str()); fclose(f); return rc[r] == 0 ? 3 : 1; }
    if (part[r]->base != expectBase) { fprintf(stderr, "rank %d: base %u, expected %zu\n", r, part[r]->base, expectBase); fclose(f); return 1; }
    expectBase += part[r]->nFrag();
    dump(f, *part[r]);
    fprintf(stderr, "rank %d: fragments [%u, %zu) of %zu\n", r, part[r]->base, (size_t)part[r]->base + part[r]->nFrag(), part[r]->nAll());
    if (part[r]->nAll() != part[0]->nAll()) { fprintf(stderr, "ranks disagree on the total\n"); fclose(f); return 1; }
  }
  fclose(f);
  if (expectBase != part[0]->nAll()) { fprintf(stderr, "slices cover %zu of %zu fragments\n", expectBase, part[0]->nAll()); return 1; }
  return 0;
}
