// t1k_amd/csrc/host/job_internal.h -- what the files of the job layer share: the job object behind the t1k_job_* C ABI and the helpers that cross
// file boundaries.  The layer is split by stage (round 6): job.cpp (creation, read input, the window loop = the device half of the stage),
// job_finish.cpp (classes, EM, selection; the group-table and variant-calling entry points), job_output.cpp (the *_aligned*.fa and table writers),
// genotyper_main.cpp (the argv-compatible executable, Genotyper.cpp:194-738) and analyzer.cpp (the post-analysis stage, Analyzer.cpp:236-733).
#pragma once
#include <fcntl.h>
#include <getopt.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <set>
#include "t1k_host.h"
#include <unordered_map>

using namespace t1k;

struct t1k_job {
  t1k_job_params prm;
  std::string err;
  RefSet ref;
  Genotyper gt;
  t1k_ctx *ctx = nullptr;           // owns the reference, pipeline 0, final stages (missing coverage, coalescing, EM)
  std::vector<t1k_ctx *> more;      // further pipelines on the same GPU (own stream and batch arenas): several batches in flight
  t1k_ctx *reader[2] = {nullptr, nullptr};  // the read sets of two consecutive windows of fragments (upload, pack, identical-read-end collapse)
  std::shared_ptr<ReadInput> in;    // the read files, mapped and indexed (the rank threads of one process share one)
  // multi-GPU: this job is rank `rank` of `nRanks`; it owns fragments [F * rank / nRanks, F * (rank + 1) / nRanks) of the input
  int rank = 0, nRanks = 1;
  t1k_comm *comm = nullptr;         // not owned
  // the *_aligned*.fa files only need the fragmentAssigned flags: with an output prefix registered before the run they are written by
  // background threads while the classes are built and the EM runs
  std::string outPrefix;
  struct StreamOut { std::string path; int what = 0, fd = -1; uint64_t offset = 0; };
  std::vector<StreamOut> stream;   // read files being written along the device loop (single-GPU jobs)
  uint32_t streamDone = 0;          // local fragments already appended
  std::thread bgWriter;
  bool bgStarted = false, bgOk = true;
  bool analyzer = false;            // analyzer mode: the rowset keeps the raw fragment assignment lists and is left alive after run_local
  t1k_rowset *rows = nullptr;       // every fragment's row, resident on the GPU until the job is coalesced
  // Per-base coverage is only read for the alleles allele selection puts on its candidate lists (t1k_gpu.h, "per-base coverage only
  // where it is read"): the windows' read sets (distinct read-ends + final overlap lists) stay resident and the coverage of those
  // alleles is added inside select() (covDeferred; T1K_COVERAGE=eager restores the per-range updates for every allele)
  bool covDeferred = false;
  std::vector<t1k_readset *> archive;
  uint64_t coverRecords = 0; double msCover = 0;
  std::vector<uint8_t> fragAssigned;
  bool ran = false, localDone = false;
  std::vector<char> whitelist;      // per allele, empty = everything allowed
  std::string abundanceFile;
  std::string assignText;           // --outputReadAssignment rows
  t1k_stats stats{};
  uint64_t distinctReadEnds = 0, readEnds = 0;
  double msLoad = 0, msDevice = 0, msHost = 0, msEm = 0, msCoalesce = 0, msWrite = 0;
};

static inline double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static inline int jobFail(t1k_job *job, int code, const std::string &msg) {
  if (job) job->err = msg;
  return code;
}

// host threads for parsing, window assembly and the output writers: -t, but never fewer than the machine offers (up to 32) --
// the GPU path is fed by the host, and the reference's -t default of 1 would starve it
static inline int hostThreadsFor(int threads) {  // threads: -t
  if (const char *e = getenv("T1K_HOST_THREADS")) return std::max(1, atoi(e));
  const int hw = (int)std::thread::hardware_concurrency();
  return std::max(1, std::max(threads, std::min(hw, 32)));
}
static inline int hostThreads(const t1k_job *job) { return hostThreadsFor(job->prm.threads); }

template <class F>
static inline void parallelRanges(size_t n, int T, F fn) {  // fn(t, begin, end) over contiguous pieces of [0, n)
  T = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, n / 4096 + 1));
  if (T == 1) { fn(0, (size_t)0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + T - 1) / T;
  for (int t = 0; t < T; ++t) th.emplace_back([=] { fn(t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
  for (auto &x : th) x.join();
}

namespace t1k {
// job.cpp
int jobCreate(const t1k_job_params *p, const char *refFasta, const std::set<std::string> *selected, t1k_job **out);  // selected: the analyzer's allele list (Genotyper.hpp:732-757)
bool loadAbundance(t1k_job *job);
// job_output.cpp: ">id\nSEQ\n" of every assigned fragment (Genotyper.cpp:680-718)
struct AlignedPlan {
  std::string path;
  int what = 0, T = 1;
  std::vector<uint64_t> pieceBytes;  // exclusive prefix over the T pieces of this rank's fragments
  uint64_t baseOffset = 0;           // of this rank's part in the file
  bool create = true;                // this rank truncates / creates the file (done in the plan step when the job's input is sharded)
};
bool streamOpen(t1k_job *job, const std::string &pfx);
void streamClose(t1k_job *job, bool removeFiles);
bool streamAppend(t1k_job *job, uint32_t fLo, uint32_t fHi, bool besideLoop);
bool planAlignedFiles(t1k_job *job, const std::string &pfx, std::vector<AlignedPlan> &plans);
bool writePlannedFiles(t1k_job *job, const std::vector<AlignedPlan> &plans);
bool writesAligned(const t1k_job *job);
// genotyper_main.cpp: same shape as the reference's PrintLog (Genotyper.cpp:113-124): users grep these lines
void logLine(const char *fmt, ...);
}  // namespace t1k
