// t1k_amd/csrc/host/job.cpp -- the whole genotyper stage as a job: the t1k_job_* C ABI and t1k_genotyper_main(), the
// argv-compatible replacement of the reference's genotyper executable (Genotyper.cpp:194-738, invoked by run-t1k:430,434).
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

static double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int jobFail(t1k_job *job, int code, const std::string &msg) {
  if (job) job->err = msg;
  return code;
}

// host threads for parsing, window assembly and the output writers: -t, but never fewer than the machine offers (up to 32) --
// the GPU path is fed by the host, and the reference's -t default of 1 would starve it
static int hostThreadsFor(int threads) {  // threads: -t
  if (const char *e = getenv("T1K_HOST_THREADS")) return std::max(1, atoi(e));
  const int hw = (int)std::thread::hardware_concurrency();
  return std::max(1, std::max(threads, std::min(hw, 32)));
}
static int hostThreads(const t1k_job *job) { return hostThreadsFor(job->prm.threads); }

template <class F>
static void parallelRanges(size_t n, int T, F fn) {  // fn(t, begin, end) over contiguous pieces of [0, n)
  T = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, n / 4096 + 1));
  if (T == 1) { fn(0, (size_t)0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + T - 1) / T;
  for (int t = 0; t < T; ++t) th.emplace_back([=] { fn(t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
  for (auto &x : th) x.join();
}

extern "C" {

static bool streamOpen(t1k_job *job, const std::string &pfx);
static void streamClose(t1k_job *job, bool removeFiles);
static bool streamAppend(t1k_job *job, uint32_t fLo, uint32_t fHi, bool besideLoop);

void t1k_job_params_default(t1k_job_params *p) {
  memset(p, 0, sizeof(*p));
  t1k_params_default(&p->dev);
  p->filter_frac = 0.15;   // Genotyper.cpp:222-225
  p->filter_cov = 1.0;
  p->cross_gene_rate = 0.04;
  p->squarem_min_alpha = 0;
  p->allele_digit_units = -1;
  p->allele_delimiter = 0;
  p->threads = 1;
  p->device = 0;
  p->output_read_assignment = 0;
  p->batch_fragments = 0;
}

static int jobCreate(const t1k_job_params *p, const char *refFasta, const std::set<std::string> *selected, t1k_job **out);
int t1k_job_create(const t1k_job_params *p, const char *refFasta, t1k_job **out) { return jobCreate(p, refFasta, nullptr, out); }

static int jobCreate(const t1k_job_params *p, const char *refFasta, const std::set<std::string> *selected, t1k_job **out) {
  if (!out || !refFasta) return T1K_ERR_ARG;
  *out = nullptr;
  t1k_job *job = new t1k_job();
  if (p) job->prm = *p; else t1k_job_params_default(&job->prm);
  *out = job;  // handed back even on failure so the caller can read the message
  const double t0 = nowMs();
  // the HIP runtime comes up and the contexts (streams, events) are created while the host parses the reference
  const char *pl = getenv("T1K_PIPELINES");
  const int nPipe = pl ? std::max(1, std::min(8, atoi(pl))) : 3;
  int rcCtx = T1K_OK;
  std::thread init;
  if (job->prm.device >= 0)
    init = std::thread([&] {
      rcCtx = t1k_ctx_create(job->prm.device, &job->prm.dev, &job->ctx);
      // Most kernels of the path are latency-bound; further independent pipelines (context, stream, arenas) on the same GPU let the
      // hardware overlap batches and hide the host's share of a batch (counter fetches, launches).  Measured on the HLA-like
      // workload (device loop): 1 M pairs 840 / 723 / 678 ms with 1 / 2 / 4 pipelines, 10 M pairs 6.4 / 6.0 / 5.9 s with 2 / 3 / 4;
      // every pipeline's arenas are device memory the driver may have to zero first (~35 ms per GB).  T1K_PIPELINES overrides.
      // (the other contexts side by side: a stream of its own is a hardware queue with a 177 MB save area that the runtime allocates and
      // touches at creation -- ~10 ms each, on the cold path of a fresh process; T1K_SERIAL_CONTEXTS=1: one after the other)
      if (rcCtx != T1K_OK) return;
      std::vector<t1k_ctx *> extra((size_t)std::max(0, nPipe - 1), nullptr);
      int rcReader[2] = {T1K_OK, T1K_OK};
      {
        const bool serial = getenv("T1K_SERIAL_CONTEXTS") != nullptr;
        std::vector<std::thread> th;
        auto run = [&](std::function<void()> f) { if (serial) f(); else th.emplace_back(f); };
        for (size_t i = 0; i < extra.size(); ++i)
          run([&, i] { t1k_ctx *c = nullptr; if (t1k_ctx_create(job->prm.device, &job->prm.dev, &c) != T1K_OK) { if (c) t1k_ctx_destroy(c); c = nullptr; } extra[i] = c; });
        for (int i = 0; i < 2; ++i)
          run([&, i] { if (t1k_ctx_create(job->prm.device, &job->prm.dev, &job->reader[i]) != T1K_OK) rcReader[i] = T1K_ERR_DEVICE; });
        for (auto &t : th) t.join();
      }
      for (size_t i = 0; i < extra.size(); ++i) {  // (a pipeline that could not be had ends the list: the job runs with the ones before it)
        if (!extra[i]) { for (size_t j = i + 1; j < extra.size(); ++j) if (extra[j]) t1k_ctx_destroy(extra[j]); break; }
        job->more.push_back(extra[i]);
      }
      if (rcReader[0] != T1K_OK || rcReader[1] != T1K_OK) rcCtx = T1K_ERR_DEVICE;
    });
  const bool loaded = job->ref.load(refFasta, job->prm.allele_digit_units, job->prm.allele_delimiter, job->err, selected);
  const double tLoaded = nowMs();
  if (init.joinable()) init.join();
  if (!loaded) return T1K_ERR_IO;
  job->gt.ref = &job->ref;
  job->gt.prm = job->prm;
  if (job->prm.device < 0) return T1K_OK;  // host-only job: group bookkeeping for tests / merging, no device work possible
  const double t1 = nowMs();
  int rc = rcCtx;
  if (rc != T1K_OK) return jobFail(job, rc, "cannot create a GPU context on device " + std::to_string(job->prm.device) + " (this build has no CPU path)");
  // upload the reference
  const RefSet &R = job->ref;
  std::string blob;
  std::vector<uint64_t> off(R.seqs.size() + 1, 0);
  std::vector<uint8_t> ex;
  for (size_t a = 0; a < R.seqs.size(); ++a) off[a + 1] = off[a] + R.seqs[a].size();
  blob.resize(off.back());
  ex.resize(off.back());
  parallelRanges(R.seqs.size(), std::min(16, hostThreads(job)), [&](int, size_t b, size_t e) {  // (28 MB each for the HLA-like reference: appended by one thread they cost 25 ms)
    for (size_t a = b; a < e; ++a) {
      memcpy(&blob[off[a]], R.seqs[a].data(), R.seqs[a].size());
      memcpy(ex.data() + off[a], R.exon[a].data(), R.exon[a].size());
    }
  });
  rc = t1k_ref_upload(job->ctx, blob.data(), off.data(), ex.data(), (uint32_t)R.seqs.size());
  if (rc != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  for (size_t i = 0; i < job->more.size(); ++i)
    if ((rc = t1k_ref_share(job->more[i], job->ctx)) != T1K_OK) {  // not enough memory: run with the pipelines we have
      for (size_t j = i; j < job->more.size(); ++j) t1k_ctx_destroy(job->more[j]);
      job->more.resize(i);
      break;
    }
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k job] reference: parse + naming + gene similarity %.1f ms (the contexts were ready %.1f ms after it), pack + index + upload + contexts %.1f ms\n", tLoaded - t0, t1 - tLoaded, nowMs() - t1);
  return T1K_OK;
}

void t1k_job_destroy(t1k_job *job) {
  if (!job) return;
  if (job->bgWriter.joinable()) job->bgWriter.join();
  if (!job->stream.empty()) streamClose(job, true);  // a run that never reached t1k_job_finish
  if (job->rows) t1k_rowset_destroy(job->rows);
  for (t1k_readset *rs : job->archive) t1k_readset_destroy(rs);
  job->archive.clear();
  for (t1k_ctx *c : job->more) t1k_ctx_destroy(c);  // before job->ctx, whose reference they alias
  for (t1k_ctx *c : job->reader) if (c) t1k_ctx_destroy(c);
  if (job->ctx) t1k_ctx_destroy(job->ctx);
  job->more.clear(); job->ctx = nullptr;
  // What is left is host memory: the mapped read files (unmapping 6 GB of touched pages takes 0.3 - 0.5 s), the record index, the
  // group tables.  Nobody waits for that: a detached thread releases it while the caller goes on (T1K_SYNC_DESTROY=1: in place).
  if (getenv("T1K_SYNC_DESTROY")) { delete job; return; }
  // (... and not right away: unmapping hundreds of MB holds the process's memory-map lock, and a caller that creates its next job at once
  // -- the benchmark's steps, a service -- had its reference parse slowed from 55 to 130 ms by the allocations waiting for that lock; half
  // a second later the next job's host threads are waiting for the GPU)
  // The deferred releases are known to the process: at exit the ones still waiting are dropped (the process's memory goes back to the
  // system anyway) and one that is in the middle of its delete is waited for, so no thread frees containers while the static
  // destructors run.
  struct Deferred {
    std::mutex m;
    std::condition_variable cv;
    int pending = 0;       // threads started and not finished
    bool exiting = false;
    static Deferred &get() {
      static Deferred *d = [] {
        Deferred *x = new Deferred();  // never destroyed: threads may outlive the static destructors' turn
        atexit([] {
          Deferred &q = Deferred::get();
          std::unique_lock<std::mutex> lk(q.m);
          q.exiting = true;
          q.cv.notify_all();
          q.cv.wait_for(lk, std::chrono::seconds(10), [&] { return q.pending == 0; });
        });
        return x;
      }();
      return *d;
    }
  };
  Deferred &d = Deferred::get();
  { std::lock_guard<std::mutex> g(d.m); ++d.pending; }
  std::thread([job, &d] {
    bool drop;
    {
      std::unique_lock<std::mutex> lk(d.m);
      d.cv.wait_for(lk, std::chrono::milliseconds(500), [&] { return d.exiting; });
      drop = d.exiting;
    }
    if (!drop) delete job;  // (at exit the job is left to the process teardown)
    { std::lock_guard<std::mutex> g(d.m); --d.pending; }
    d.cv.notify_all();
  }).detach();
}

const char *t1k_job_last_error(const t1k_job *job) { return job ? job->err.c_str() : "no job"; }
t1k_ctx *t1k_job_ctx(t1k_job *job) { return job ? job->ctx : nullptr; }

int t1k_job_load_reads_multi(t1k_job *job, const char *const *files1, uint32_t n1, const char *const *files2, uint32_t n2, const char *barcodeFile) {
  if (!job || !files1 || n1 == 0) return T1K_ERR_ARG;
  const double t0 = nowMs();
  job->in.reset(new ReadInput());
  job->ran = false; job->localDone = false;
  std::vector<std::string> f1(files1, files1 + n1), f2;
  if (files2) f2.assign(files2, files2 + n2);
  bool whole = true;
  if (job->nRanks > 1 && job->comm && !(barcodeFile && *barcodeFile) && !getenv("T1K_NO_SHARDED_INPUT")) {
    // a rank of a sharded job (t1k_job_set_shard came first): index this rank's fragments only -- collective over the communicator
    ReadInput::ShardComm sc;
    sc.rank = job->rank; sc.nRanks = job->nRanks;
    t1k_comm *comm = job->comm;
    sc.allgatherv = [comm](void *buf, const uint64_t *bytes, const uint64_t *displ, uint64_t total) { return t1k_comm_allgatherv_host(comm, buf, bytes, displ, total) == T1K_OK; };
    const int r = job->in->openSharded(f1, f2, hostThreads(job), sc, job->err);
    if (r < 0) { job->in.reset(); return T1K_ERR_IO; }
    if (r > 0) whole = false;
    else job->in.reset(new ReadInput());
  }
  if (whole && !job->in->open(f1, f2, barcodeFile ? barcodeFile : "", hostThreads(job), job->err)) { job->in.reset(); return T1K_ERR_IO; }
  job->msLoad = nowMs() - t0;
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k job] read files mapped + indexed: %zu of %zu fragments, %.1f ms\n", job->in->nFrag(), job->in->nAll(), job->msLoad);
  return T1K_OK;
}

// the read input on its own (include/t1k_gpu.h): opened by a second thread of the caller while t1k_job_create runs
struct t1k_reads {
  std::unique_ptr<ReadInput> in;
  std::string err;
  double ms = 0;
};
int t1k_reads_open(const char *const *files1, uint32_t n1, const char *const *files2, uint32_t n2, const char *barcodeFile, int threads, t1k_reads **out) {
  if (!out) return T1K_ERR_ARG;
  *out = nullptr;
  if (!files1 || n1 == 0) return T1K_ERR_ARG;
  t1k_reads *r = new t1k_reads();
  *out = r;  // handed back even on failure so the caller can read the message
  const double t0 = nowMs();
  r->in.reset(new ReadInput());
  std::vector<std::string> f1(files1, files1 + n1), f2;
  if (files2) f2.assign(files2, files2 + n2);
  if (!r->in->open(f1, f2, barcodeFile ? barcodeFile : "", hostThreadsFor(threads), r->err)) { r->in.reset(); return T1K_ERR_IO; }
  r->ms = nowMs() - t0;
  return T1K_OK;
}
int t1k_reads_open_stream(const char *const *files1, uint32_t n1, const char *const *files2, uint32_t n2, const char *barcodeFile, int threads, t1k_reads **out) {
  if (!out) return T1K_ERR_ARG;
  *out = nullptr;
  if (!files1 || n1 == 0) return T1K_ERR_ARG;
  const char *e = getenv("T1K_STREAM_GZ");
  if (!(e && atoi(e) == 0) && !getenv("T1K_LONG_READS")) {  // (setting reads aside needs the longest kept read before the loop)
    t1k_reads *r = new t1k_reads();
    const double t0 = nowMs();
    r->in.reset(new ReadInput());
    std::vector<std::string> f1(files1, files1 + n1), f2;
    if (files2) f2.assign(files2, files2 + n2);
    if (r->in->openStreaming(f1, f2, barcodeFile ? barcodeFile : "", r->err)) { r->ms = nowMs() - t0; *out = r; return T1K_OK; }
    const bool failed = !r->err.empty();
    if (failed) { r->in.reset(); *out = r; return T1K_ERR_IO; }
    delete r;  // not eligible: opened whole
  }
  return t1k_reads_open(files1, n1, files2, n2, barcodeFile, threads, out);
}
const char *t1k_reads_last_error(const t1k_reads *r) { return r ? r->err.c_str() : "no read input"; }
int t1k_reads_fragments(const t1k_reads *r, uint64_t *nFragments) {
  if (!r || !r->in || !nFragments) return T1K_ERR_ARG;
  if (r->in->streaming) { r->in->streamWait(~(size_t)0); *nFragments = r->in->streamAvail(); return T1K_OK; }  // (the stream's end: every record is counted)
  *nFragments = r->in->nAll();
  return T1K_OK;
}
void t1k_reads_close(t1k_reads *r) { delete r; }
int t1k_job_attach_reads(t1k_job *job, t1k_reads *r) {
  if (!job || !r) { delete r; return T1K_ERR_ARG; }
  if (!r->in) { job->err = r->err.empty() ? "t1k_job_attach_reads: the read input was not opened" : r->err; delete r; return T1K_ERR_IO; }
  // (a rank of a sharded job opens its own share: t1k_job_load_reads -- also where that call itself would have opened the whole input, e.g.
  // with a barcode file: the handle cannot know, and a sharded caller has the collective open anyway)
  if (job->nRanks > 1 && job->comm && !getenv("T1K_NO_SHARDED_INPUT")) {
    delete r;
    return jobFail(job, T1K_ERR_STATE, "t1k_job_attach_reads: a rank of a sharded job indexes its own fragments (t1k_job_set_shard, then t1k_job_load_reads)");
  }
  job->in = std::move(r->in);
  job->ran = false; job->localDone = false;
  job->msLoad = r->ms;
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k job] read files mapped + indexed beside the job's creation: %zu of %zu fragments, %.1f ms\n", job->in->nFrag(), job->in->nAll(), job->msLoad);
  delete r;
  return T1K_OK;
}

int t1k_job_load_reads(t1k_job *job, const char *file1, const char *file2, const char *barcodeFile) {
  if (!job || !file1) return T1K_ERR_ARG;
  return t1k_job_load_reads_multi(job, &file1, 1, file2 ? &file2 : nullptr, file2 ? 1 : 0, barcodeFile);
}

int t1k_job_set_reads(t1k_job *job, const char *seq1, const uint64_t *off1, const char *seq2, const uint64_t *off2, uint32_t nFragments) {
  if (!job || !seq1 || !off1 || (seq2 && !off2)) return T1K_ERR_ARG;
  job->in.reset(new ReadInput());
  job->ran = false; job->localDone = false;
  job->in->setMemory(seq1, off1, seq2, off2, nFragments);
  return T1K_OK;
}

int t1k_job_set_shard(t1k_job *job, int rank, int nRanks, t1k_comm *comm) {
  if (!job || nRanks < 1 || rank < 0 || rank >= nRanks || (nRanks > 1 && !comm)) return T1K_ERR_ARG;
  job->rank = rank; job->nRanks = nRanks; job->comm = comm;
  return T1K_OK;
}

int t1k_job_share_reads(t1k_job *dst, t1k_job *src) {
  if (!dst || !src || !src->in) return T1K_ERR_ARG;
  dst->in = src->in;
  dst->ran = false; dst->localDone = false;
  return T1K_OK;
}

// Genotyper::InitAlleleAbundance (Genotyper.hpp:1016-1051): "-a FILE" injects abundances and bypasses the EM
static bool loadAbundance(t1k_job *job) {
  FILE *fp = fopen(job->abundanceFile.c_str(), "r");
  if (!fp) { job->err = "cannot open " + job->abundanceFile; return false; }
  std::map<std::string, int> byName;
  for (size_t a = 0; a < job->ref.al.size(); ++a) byName[job->ref.al[a].name] = (int)a;
  char name[512];
  int t1, t2;
  double count, abundance;
  if (fscanf(fp, "%511s %511s %511s %511s %511s", name, name, name, name, name) != 5) { fclose(fp); job->err = "bad abundance file"; return false; }
  while (fscanf(fp, "%511s %d %d %lf %lf", name, &t1, &t2, &count, &abundance) == 5) {
    auto it = byName.find(name);
    int a = it == byName.end() ? 0 : it->second;
    job->ref.al[a].abundance = count;
  }
  fclose(fp);
  for (auto &members : job->gt.ecAlleles) {
    double total = 0;
    for (int a : members) total += job->ref.al[a].abundance;
    for (int a : members) job->ref.al[a].ecAbundance = total;
  }
  job->gt.setAbundance(nullptr, {});
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// The device half of the stage (Genotyper.cpp:443-650 up to the coalesced read groups).
//
// Fragments stream through the GPU in windows of T1K_WINDOW fragments (file order).  A window is prepared by its own thread
// on one of two read-set contexts -- the read-ends' text gathered from the mapped files by the host threads, uploaded, packed,
// identical read-ends collapsed (t1k_reads_dedupe: the reference's sort + run-length loop, Genotyper.cpp:451-480) -- while the
// pipelines work on the window before it.  Work items of a window: AssignRead over ranges of its distinct read-ends, and mate pairing
// over ranges of its fragments straight into the job's rowset.  The distinct read-ends are numbered in the order of their first use
// (t1k_reads_dedupe), so the fragments of pairing range q only name distinct read-ends below pairNeed[q] x the assignment range:
// the range is paired as soon as that prefix of the assignment ranges is done, beside the assignment of the rest (with the window
// that holds 82 % of a 10 M-pair job paired only after its last assignment range, the loop ended in 0.4 s of pairing alone).  A pipeline that finds no item left in the oldest window starts on the next one (its lists go
// to the other slot of the pipeline's overlap store), so the GPU does not drain at window boundaries.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Window {
  uint32_t f0 = 0, f1 = 0;          // fragments [f0, f1)
  int slot = 0;                     // read-set context and overlap-store slot
  uint32_t nDistinct = 0;
  std::vector<uint32_t> distinctOf; // per read-end of the window: index in the distinct set
  std::vector<uint8_t> hasN;        // per fragment
  uint32_t assignBatch = 0, nAssign = 0, nextAssign = 0, doneAssign = 0;
  uint32_t pairBatch = 0, nPair = 0, nextPair = 0, donePair = 0;
  std::vector<uint32_t> pairNeed;   // per pairing range: how many assignment ranges (a prefix) hold the read-ends of its fragments and of all before
  std::vector<char> assignDone;     // per assignment range
  uint32_t assignPrefix = 0;        // assignment ranges [0, assignPrefix) are done
  bool ready = false, done = false;
  std::vector<char> pairDone;       // per pairing range of the window: finished (the output writer follows these)
  double tReady = 0, tDone = 0, msPrep = 0;
  std::vector<char> touched;        // per pipeline: has it attached to this window yet (first touch empties its store slot)
  uint64_t ovlRecords = 0;          // overlap records its ranges left in the store
  bool deferred = false;            // its coverage is added later, for the selected alleles only: the read set is kept when the window is done
  uint32_t nExternal = 0;           // distinct read-ends whose sequence an earlier kept window assigned (t1k_xwin_link): not assigned again
  bool linking = false, linked = false;  // their list-table entries are being / have been copied from those windows (before the first pairing range)
};
}  // namespace


int t1k_job_run_local(t1k_job *job) {
  if (!job || !job->ctx) return jobFail(job, T1K_ERR_STATE, "this job has no GPU context (device = -1): it cannot run");
  if (!job->in) return jobFail(job, T1K_ERR_STATE, "no reads loaded");
  int rc;
  // a writer still formatting the last run's records reads fragAssigned and the stream files this run is about to reset
  if (job->bgWriter.joinable()) job->bgWriter.join();
  if (!job->stream.empty()) streamClose(job, false);
  job->bgStarted = false;
  // fresh state (a job may be run repeatedly, e.g. by the benchmark)
  Genotyper &gt = job->gt;
  // A streamed input (t1k_reads_open_stream) hands its fragments over while the loop runs: one rank, the genotyper's own loop.  Anything
  // else waits for the end of the stream here and goes on as with a file opened whole.
  if (job->in->streaming && (job->nRanks > 1 || job->analyzer || getenv("T1K_LONG_READS"))) {
    if (!job->in->streamFinish(job->err)) return jobFail(job, T1K_ERR_IO, job->err);
  }
  const bool gzStream = job->in->streaming;
  const ReadInput &in = *job->in;
  gt.groupPtr.assign(1, 0); gt.groupEnt.clear(); gt.groupFirst.clear(); gt.groupOfHash.clear(); gt.assignedFragments = 0; gt.emIterations = 0;
  gt.readLength = in.maxLen;  // Genotyper.cpp:443
  for (auto &a : job->ref.al) { a.rank = -1; a.quality = -1; a.abundance = a.ecAbundance = 0; a.ec = -1; a.missingCov = 0; }
  const uint32_t Fall = (uint32_t)in.nAll();
  const uint32_t fBeg = (uint32_t)((uint64_t)Fall * job->rank / job->nRanks), fEnd = (uint32_t)((uint64_t)Fall * (job->rank + 1) / job->nRanks);
  uint32_t F = fEnd - fBeg;  // this rank's fragments; local index f <-> fragment fBeg + f of the input (a streamed input: their upper bound until the loop has ended)
  if (in.sharded && (in.shardRank != job->rank || in.shardRanks != job->nRanks || in.base != fBeg || in.nFrag() != F))
    return jobFail(job, T1K_ERR_STATE, "the reads were loaded for another shard than the one this job runs as");
  const uint32_t inBase = in.base;  // fragment f of the input = record in.frag[f - inBase] held here
  const uint32_t per = in.paired ? 2 : 1;
  job->fragAssigned.assign(Fall, 0);
  job->assignText.clear();
  memset(&job->stats, 0, sizeof(job->stats));
  job->distinctReadEnds = 0; job->readEnds = (uint64_t)F * per;
  // Reads longer than the kernels' hit masks span (320 bases) stop the run before any output exists -- unless the caller asks for them
  // to be set aside (T1K_LONG_READS=drop): their fragments then take no part (never assigned, not written to the aligned-read files,
  // counted in a warning), which is NOT what the reference does with them; everything else is genotyped as usual.
  const uint32_t lenLimit = job->prm.dev.max_read_len > 0 ? (uint32_t)job->prm.dev.max_read_len : 0xFFFFFFFFu;
  bool dropLong = false;
  if ((uint32_t)in.maxLen > lenLimit) {
    const char *e = getenv("T1K_LONG_READS");
    if (!(e && !strcmp(e, "drop")))
      return jobFail(job, T1K_ERR_ARG, "a read of " + std::to_string(in.maxLen) + " bases is longer than this build handles (" + std::to_string(job->prm.dev.max_read_len) +
                                           "); T1K_LONG_READS=drop sets the fragments of such reads aside instead of stopping");
    dropLong = true;
    // Genotyper.cpp:443: the longest read -- of the fragments that take part
    uint64_t keptMax = 0;
    {
      std::mutex mm;
      parallelRanges(in.nFrag(), hostThreads(job), [&](int, size_t b, size_t e) {
        uint64_t mx = 0;
        for (size_t i = b; i < e; ++i) {
          const uint32_t r = in.frag[i];
          bool skip = false;
          uint32_t here = 0;
          for (uint32_t m = 0; m < (in.paired ? 2u : 1u); ++m) { const uint32_t l = in.side[m].seqL[r]; skip = skip || l > lenLimit; here = std::max(here, l); }
          if (!skip) mx = std::max<uint64_t>(mx, here);
        }
        std::lock_guard<std::mutex> g(mm);
        keptMax = std::max(keptMax, mx);
      });
    }
    if (job->nRanks > 1 && job->comm) {
      std::vector<uint64_t> all((size_t)job->nRanks, 0);
      if (t1k_comm_allgather_u64(job->comm, &keptMax, 1, all.data()) != T1K_OK) return jobFail(job, T1K_ERR_DEVICE, t1k_comm_last_error(job->comm));
      for (uint64_t v : all) keptMax = std::max(keptMax, v);
    }
    gt.readLength = (int)keptMax;
  }
  std::atomic<uint64_t> droppedFragments{0};
  if ((rc = t1k_coverage_reset(job->ctx)) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  for (t1k_ctx *c : job->more)
    if ((rc = t1k_coverage_reset(c)) != T1K_OK) return jobFail(job, rc, t1k_last_error(c));
  if (job->rows) { t1k_rowset_destroy(job->rows); job->rows = nullptr; }
  {
    std::vector<uint8_t> wl(job->whitelist.begin(), job->whitelist.end());
    if ((rc = t1k_rowset_create(job->ctx, F, wl.empty() ? nullptr : wl.data(), &job->rows)) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
    if (job->analyzer) t1k_rowset_set_raw(job->rows, 1);
  }
  for (t1k_readset *rs : job->archive) t1k_readset_destroy(rs);
  job->archive.clear();
  job->coverRecords = 0; job->msCover = 0;
  {
    const char *cm = getenv("T1K_COVERAGE");
    job->covDeferred = !job->analyzer && !(cm && !strcmp(cm, "eager"));
  }
  // device memory the kept read sets may take (about 6 KB per fragment of an HLA-sized reference: 58 GB at 10 M pairs); windows beyond
  // it fall back to the per-range coverage updates -- both kinds add up, only the selected alleles' sums are read
  uint64_t archiveBudget = 0, archivedBytes = 0, archivedFrags = 0, archivedNeed = 0;  // (Bytes: what the kept sets hold, chunk slack included; Need: their records and reads alone)
  bool eagerFromNowOn = !job->covDeferred;
  if (job->covDeferred) {
    uint64_t freeB = 0, totalB = 0;
    (void)t1k_device_memory(job->prm.device, &freeB, &totalB);
    archiveBudget = totalB / 2;
    if (const char *e = getenv("T1K_ARCHIVE_GB")) archiveBudget = (uint64_t)(atof(e) * 1073741824.0);
  }
  const double tStart = nowMs();
  const int T = hostThreads(job);
  // identical read-ends across windows: only between windows whose lists stay resident (T1K_CROSS_WINDOW=0 turns it off)
  struct XwinHolder { t1k_xwin *x = nullptr; ~XwinHolder() { t1k_xwin_destroy(x); } } xw;
  std::vector<t1k_ctx *> pipes{job->ctx};
  pipes.insert(pipes.end(), job->more.begin(), job->more.end());
  const int P = (int)pipes.size();
  // Windows of fragments are cut while the job runs (by the preparation thread).  Large windows matter because identical read-ends
  // collapse per window: error-free reads of a locus come back in every window (10 M pairs: 5.89 M distinct read-ends with 2 M-fragment
  // windows, 5.28 M with 8 M).  But a window can only be prepared while the GPU still has the previous one to work on.  So: a small
  // first window starts the GPU after a short preparation; a later window is as large as can be prepared in the time the GPU needs
  // for what is already prepared -- both rates are measured on the windows before it -- up to T1K_WINDOW fragments (default 8 M: a
  // window's overlap lists stay in the store until its fragments are paired, about 6 KB per fragment of an HLA-sized reference).
  uint32_t windowFrags = 1u << 23;
  if (const char *e = getenv("T1K_WINDOW")) windowFrags = (uint32_t)std::max(8, atoi(e));
  uint32_t assignBatch = job->prm.batch_fragments > 0 ? (uint32_t)job->prm.batch_fragments * per : 32768u;
  if (const char *eb = getenv("T1K_BATCH")) assignBatch = (uint32_t)std::max(8, atoi(eb)) * per;  // tuning / test aid (in fragments, as in round 1)
  uint32_t pairBatch = 1u << 16;
  if (const char *eb = getenv("T1K_PAIR_BATCH")) pairBatch = (uint32_t)std::max(8, atoi(eb));  // test aid
  const uint32_t maxWindows = 4096;
  {
    const char *e = getenv("T1K_CROSS_WINDOW");
    if (job->covDeferred && F > 0 && !(e && atoi(e) == 0)) {
      // (no memory for the table: the windows then assign their own copies, as before -- an optimisation must not end the job)
      if (t1k_xwin_create(job->ctx, (uint64_t)F * per, maxWindows, &xw.x) != T1K_OK) {
        xw.x = nullptr;
        if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k job] no table of read-ends across windows: %s\n", t1k_last_error(job->ctx));
      }
    }
  }
  std::vector<Window> win;
  win.reserve(maxWindows);  // windows are appended while other threads hold references: the vector never reallocates
  uint32_t firstWindow = std::min<uint32_t>(windowFrags, std::max<uint32_t>(65536u, windowFrags / 32));
  if (const char *e = getenv("T1K_FIRST_WINDOW")) firstWindow = (uint32_t)std::max(8, atoi(e));
  double fixedGrowth = 0;  // T1K_WINDOW_GROWTH: a fixed factor instead of the measured one
  if (const char *e = getenv("T1K_WINDOW_GROWTH")) fixedGrowth = std::max(1.0, atof(e));
  struct Shared {
    std::mutex m;
    std::condition_variable cv;
    int err = T1K_OK;
    std::string errMsg;
    uint32_t oldest = 0;  // first window that is not done
    uint32_t created = 0; // windows cut so far
    bool allCreated = false;
    uint64_t pairedFrags = 0;  // fragments whose rows are written (with stats.rows: the job's rows per fragment so far)
  } sh;
  auto fail = [&](int code, const std::string &msg) {
    std::lock_guard<std::mutex> g(sh.m);
    if (sh.err == T1K_OK) { sh.err = code; sh.errMsg = msg; }
    sh.cv.notify_all();
  };
  sh.allCreated = F == 0;
  double msPrep = 0;
  const bool traceTasks = getenv("T1K_DEBUG_TASKS") != nullptr;
  // ---- window preparation ------------------------------------------------------------------------------------------
  auto prepare = [&] {
    // offsets of a window's read-ends (per read-set slot) and the staging slots its text goes through: plain allocations, never
    // value-initialised (round 2 gathered the text into a std::vector whose resize zeroed 2.5 GB on this one thread: 0.3 s)
    // ... and page-locked (t1k_pinned_alloc, cached per process): the upload is then one DMA instead of a copy staged on this thread
    struct Raw {
      void *p = nullptr; size_t cap = 0; bool pinned = false;
      ~Raw() { drop(); }
      void drop() { if (pinned) t1k_pinned_free(p); else free(p); p = nullptr; cap = 0; }
      void *need(size_t bytes) {
        if (bytes <= cap) return p;
        drop();
        static const bool noPin = getenv("T1K_NO_PINNED_TEXT") != nullptr;
        p = noPin ? nullptr : t1k_pinned_alloc(bytes);
        pinned = p != nullptr;
        if (!p) p = malloc(bytes);
        cap = p ? bytes : 0;
        return p;
      }
    } offs[2], stage;     // (the three staging slots are ONE block, page-locked when the first window is prepared: pinning them one by one as they are first
                          // filled -- the first window fills one -- was tried in round 5 and met a 4.5 s stall of a later window's first piece in one cold run)
    // size of window w (sh.m held): what can be prepared while the GPU works off the windows that are ready but not done
    auto windowSize = [&](uint32_t w) -> uint64_t {
      uint64_t size = firstWindow;
      if (w > 0) {
        const Window &prev = win[w - 1];
        const double tp = prev.msPrep / std::max<double>(1, prev.f1 - prev.f0);  // ms per fragment, preparation
        double tg = 0;                                                            // ms per fragment, GPU (the last finished window)
        for (uint32_t v = w; v-- > 0;)
          if (win[v].done && win[v].tDone > win[v].tReady) { tg = (win[v].tDone - win[v].tReady) / std::max<double>(1, win[v].f1 - win[v].f0); break; }
        uint64_t waiting = 0;  // fragments ready for the GPU and not done yet
        for (uint32_t v = sh.oldest; v < w; ++v) waiting += win[v].f1 - win[v].f0;
        double factor = fixedGrowth > 0 ? fixedGrowth : (tg > 0 && tp > 0 ? 0.85 * tg / tp : 6.0);
        factor = std::min(16.0, std::max(1.0, factor));
        size = (uint64_t)(factor * (double)std::max<uint64_t>(waiting, firstWindow / 2));
        size = std::min<uint64_t>(windowFrags, std::max<uint64_t>(size, firstWindow));
      }
      return size;
    };
    uint64_t fNext = 0;
    uint64_t have = F;      // fragments that can be cut into windows (a streamed input: the records indexed so far)
    bool ended = !gzStream; // ... and whether that is all there will be
    for (uint32_t w = 0; gzStream ? true : fNext < F; ++w) {
      if (gzStream) {
        // the window the loop's own rule asks for, if the stream has got that far; what is there (a quarter of a first window at least) when
        // the GPU has nothing left to work on; all that is left at the stream's end (and when the window table is nearly full)
        // (the FIRST window waits for its full size: the stream has been running since before the contexts came up and is far past it by the time
        // this thread asks; windows cut smaller than the loop's own first window put its growth rule on a path it was not tuned for)
        const uint64_t least = fNext + (w == 0 ? firstWindow : std::max<uint32_t>(16384u, firstWindow / 4));
        for (;;) {
          const int st = in.streamState();
          have = in.streamAvail();
          if (st < 0) { std::string e; (void)job->in->streamFinish(e); fail(T1K_ERR_IO, e.empty() ? "cannot read the .gz input" : e); return; }
          if (st == 1) { have = in.streamAvail(); ended = true; break; }
          bool idle;
          uint64_t want;
          { std::lock_guard<std::mutex> g(sh.m); if (sh.err != T1K_OK) return; idle = sh.oldest >= sh.created; want = fNext + windowSize(w); }
          if (w + 8 >= maxWindows) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }  // (the rest as one window)
          if (have >= want || (have >= least && idle)) break;
          std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
        if (ended && fNext >= have) {
          std::lock_guard<std::mutex> g(sh.m);
          sh.allCreated = true;
          sh.cv.notify_all();
          break;
        }
      }
      {
        std::unique_lock<std::mutex> lk(sh.m);
        sh.cv.wait(lk, [&] { return sh.err != T1K_OK || w < 2 || win[w - 2].done; });  // the read-set context of window w - 2 is free again
        if (sh.err != T1K_OK) return;
        if (gzStream && !ended) { const int st = in.streamState(); have = in.streamAvail(); if (st == 1) { have = in.streamAvail(); ended = true; } }  // (what arrived while this thread waited for the slot)
        // size of this window: what can be prepared while the GPU works off the windows that are ready but not done
        const uint64_t size = windowSize(w);
        Window N;
        N.f0 = (uint32_t)fNext; N.f1 = (uint32_t)std::min<uint64_t>(have, fNext + size);
        if (ended && (have - N.f1 < size / 4 || w + 1 >= maxWindows)) N.f1 = (uint32_t)have;  // no small tail window
        N.slot = (int)(w & 1);
        N.touched.assign(P, 0);
        if (!eagerFromNowOn) {
          // what a window's read set will take: its fragments x the bytes per fragment of the windows kept so far (16 B per overlap record
          // + the packed distinct read-ends; larger windows collapse more read-ends, so this errs high), a quarter to spare, and the
          // slack of a partly filled store chunk per pipeline
          const double perFrag = archivedFrags ? 1.25 * (double)archivedNeed / (double)archivedFrags : 0.0;
          const uint64_t slack = (uint64_t)P * (uint64_t)std::max(1, job->prm.dev.store_chunk_mb) << 20;
          auto estimate = [&](const Window &X) { return archivedFrags ? (uint64_t)(perFrag * (double)(X.f1 - X.f0)) + slack : 0ull; };
          uint64_t open = 0;  // windows cut earlier whose read sets are not kept yet
          for (uint32_t v = sh.oldest; v < w; ++v)
            if (win[v].deferred && !win[v].done) open += estimate(win[v]);
          bool fits = archivedBytes + open + estimate(N) <= archiveBudget;
          // ... and beside everything else the job holds at its peak (50 M pairs of the benchmark's sample ended with "out of memory" in the
          // overlap store under the half-the-device rule alone, 30 M ran within 20 GB of the device): what is allocated now beside rows and
          // kept sets (arenas, reference, the windows in flight), the rows projected for the whole job (the chunk cursors over the fragments
          // paired so far), one more full window in flight, the coalescing's work space.
          // (test aid T1K_TEST_ARCHIVE_HEADROOM_MB: evaluate the rule from the first paired fragment on and pretend the device ends that
          // many MB above what is allocated now, so that a small job exercises the fallback)
          static const char *testHeadroom = getenv("T1K_TEST_ARCHIVE_HEADROOM_MB");
          if (fits && sh.pairedFrags >= (testHeadroom ? 1u : 65536u) && !getenv("T1K_ARCHIVE_GB")) {
            uint64_t freeNow = 0, totalNow = 0, rowsNow = 0, entriesNow = 0;
            if (t1k_device_memory(job->prm.device, &freeNow, &totalNow) == T1K_OK && t1k_rowset_device_bytes(job->rows, &rowsNow, &entriesNow) == T1K_OK && totalNow) {
              if (testHeadroom) { const uint64_t used = totalNow - std::min(freeNow, totalNow); totalNow = used + ((uint64_t)atoll(testHeadroom) << 20); freeNow = totalNow - used; }
              const double projectedRows = (double)entriesNow / (double)sh.pairedFrags * (double)F * sizeof(t1k_row_entry) * 1.1 + 32.0 * (double)F + std::min(4e9, 0.015 * (double)totalNow);  // (+ 10 %, the per-fragment tables, chunk slack: 4 GB on this part, never more than 1.5 % of a smaller device)
              const uint64_t usedNow = totalNow - std::min(freeNow, totalNow);
              const uint64_t other = usedNow > archivedBytes + rowsNow ? usedNow - archivedBytes - rowsNow : 0;
              const double coalesceWork = 48.0 * (double)F + std::min(8e9, 0.03 * (double)totalNow);  // (sort keys and tables of the coalescing, the EM's arrays, a margin: 8 GB here, at most 3 % of the device)
              // (a window that is not kept still holds its read set while it is in flight, and so does the one prepared behind it: what is
              // allocated now covers the windows in flight now, one more of full size may come on top while fragments remain)
              const double transient = perFrag * (double)std::min<uint64_t>(windowFrags, F - N.f1);
              // ... and a reserve the projection does not see: the arenas of the pipelines still grow with the ranges' demand, the runtime needs
              // device memory of its own (queue scratch, code objects), and the runtime ABORTS the process when it cannot have it -- round 5 saw
              // two 30 M-pair runs in six end that way at 290 of 309 GB (T1K_ARCHIVE_RESERVE_GB overrides; 4 % of the device, 8 GB at least)
              static const double reserve = [] { const char *e = getenv("T1K_ARCHIVE_RESERVE_GB"); return e ? atof(e) * 1073741824.0 : -1.0; }();
              const double keepFree = reserve >= 0 ? reserve : std::max(8e9, 0.04 * (double)totalNow);
              fits = (double)other + (double)(archivedBytes + open + estimate(N)) + std::max(projectedRows, (double)rowsNow) + coalesceWork + transient + keepFree <= (double)totalNow;
              if (getenv("T1K_DEBUG_ARCHIVE"))
                fprintf(stderr, "[t1k job] window %u (%u fragments): %llu row entries in the chunks, %llu fragments paired, stats.rows %llu, rows %.1f GB, kept %.1f + open %.1f + this %.1f GB, other %.1f GB, transient %.1f GB -> %s\n",
                        w, N.f1 - N.f0, (unsigned long long)entriesNow, (unsigned long long)sh.pairedFrags, (unsigned long long)job->stats.rows, rowsNow / 1e9, archivedBytes / 1e9, open / 1e9,
                        estimate(N) / 1e9, other / 1e9, transient / 1e9, fits ? "kept" : "not kept");
              if (!fits && getenv("T1K_DEBUG_PHASES"))
                fprintf(stderr, "[t1k job] window %u: read sets are not kept from here on: %.1f GB of rows projected for the job (%.1f GB so far), %.1f GB kept, %.1f GB other, device %.1f GB\n",
                        w, projectedRows / 1e9, rowsNow / 1e9, archivedBytes / 1e9, other / 1e9, totalNow / 1e9);
            }
          }
          if (fits) N.deferred = true;
          else eagerFromNowOn = true;
        }
        fNext = N.f1;
        win.push_back(std::move(N));
        sh.created = w + 1;
        if (ended && fNext >= have) sh.allCreated = true;
      }
      Window &W = win[w];
      const double t0 = nowMs();
      const uint32_t nf = W.f1 - W.f0, ne = nf * per;
      uint64_t *off = (uint64_t *)offs[W.slot].need(((size_t)ne + 1) * 8);
      if (!off) { fail(T1K_ERR_DEVICE, "window preparation: out of host memory"); return; }
      W.hasN.resize(nf);
      // lengths -> offsets (pieces, then a carry per piece)
      std::vector<uint64_t> pieceBytes(T + 2, 0);
      std::vector<uint32_t> pieceMax(T + 1, 0);
      // (a fragment with an over-long read, under T1K_LONG_READS=drop: both ends go in as empty sequences)
      auto setAside = [&](uint32_t r) {
        if (!dropLong) return false;
        for (uint32_t m = 0; m < per; ++m) if (in.side[m].seqL[r] > lenLimit) return true;
        return false;
      };
      parallelRanges(nf, T, [&](int t, size_t b, size_t e) {
        uint64_t run = 0, aside = 0;
        uint32_t mx = 0;
        for (size_t i = b; i < e; ++i) {
          const uint32_t r = in.frag[fBeg - inBase + W.f0 + i];
          const bool skip = setAside(r);
          aside += skip ? 1 : 0;
          for (uint32_t m = 0; m < per; ++m) {
            const uint32_t len = skip ? 0 : in.side[m].seqL[r];
            off[i * per + m] = run; run += len; mx = std::max(mx, len);
          }
        }
        pieceBytes[t + 1] = run; pieceMax[t] = mx;
        if (aside) droppedFragments += aside;
      });
      for (int t = 0; t < T; ++t) pieceBytes[t + 1] += pieceBytes[t];  // pieces the loop did not use hold 0
      const uint64_t total = pieceBytes[T];
      uint32_t maxLen = 0;
      for (int t = 0; t < T; ++t) maxLen = std::max(maxLen, pieceMax[t]);
      if (gzStream && maxLen > lenLimit) {  // (a file opened whole is checked before any work; a stream meets the read when it gets there)
        fail(T1K_ERR_ARG, "a read of " + std::to_string(maxLen) + " bases is longer than this build handles (" + std::to_string(job->prm.dev.max_read_len) +
                              "); T1K_LONG_READS=drop sets the fragments of such reads aside instead of stopping");
        return;
      }
      parallelRanges(nf, T, [&](int t, size_t b, size_t e) {
        const uint64_t carry = pieceBytes[t];
        for (size_t i = b; i < e; ++i)
          for (uint32_t m = 0; m < per; ++m) off[i * per + m] += carry;
      });
      off[ne] = total;
      // the text goes to the device through three staging slots of page-locked memory: while a slot's piece is on its way (one DMA,
      // nobody's small copies wait behind it) the host threads gather the next piece into another slot.  (Pinning the whole 2.5 GB
      // text instead was as fast for a process that keeps the buffer -- and 0.4 s slower for the executable, which pins it once.)
      static const size_t slotBytes = [] { const char *e = getenv("T1K_STAGE_MB"); return (size_t)std::max(1, e ? atoi(e) : 96) << 20; }();
      const int nSlots = 3;
      char *ring = (char *)stage.need(nSlots * slotBytes);
      if (!ring) { fail(T1K_ERR_DEVICE, "window preparation: out of host memory"); return; }
      t1k_ctx *rd = job->reader[W.slot];
      int r = t1k_reads_upload_begin(rd, ne, total, (int)maxLen);
      if (r == T1K_OK) r = t1k_reads_upload_piece(rd, 1, off, 0, ((uint64_t)ne + 1) * 8, 3);
      double msGather = 0;
      uint32_t nPieces = 0;
      for (size_t i0 = 0; i0 < nf && r == T1K_OK; ++nPieces) {
        const uint64_t base = off[i0 * per];
        size_t lo = i0 + 1, hi = nf;  // the last fragment boundary whose text still fits the slot (one fragment always does)
        while (lo < hi) {
          const size_t mid = lo + (hi - lo + 1) / 2;
          if (off[mid * per] - base <= slotBytes) lo = mid; else hi = mid - 1;
        }
        const size_t i1 = lo;
        const int slot = (int)(nPieces % nSlots);
        if ((r = t1k_reads_upload_wait(rd, slot)) != T1K_OK) break;
        char *dst = ring + (size_t)slot * slotBytes;
        const double tg = nowMs();
        parallelRanges(i1 - i0, T, [&](int, size_t b, size_t e) {
          for (size_t i = i0 + b; i < i0 + e; ++i) {
            const uint32_t rr = in.frag[fBeg - inBase + W.f0 + i];
            bool n = false;
            const bool skip = setAside(rr);
            for (uint32_t m = 0; m < per; ++m) {
              const uint32_t len = skip ? 0 : in.side[m].seqL[rr];
              memcpy(dst + (off[i * per + m] - base), in.side[m].seqP[rr], len);
              n = n || memchr(in.side[m].seqP[rr], 'N', len) != nullptr;  // Genotyper.cpp: hasN = strchr(seq, 'N')
            }
            W.hasN[i] = n ? 1 : 0;
          }
        });
        msGather += nowMs() - tg;
        r = t1k_reads_upload_piece(rd, 0, dst, base, off[i1 * per] - base, slot);
        i0 = i1;
      }
      const double tText = t0 + msGather;  // (reported as "text": the gathering alone; "upload" is the rest of the pipelined loop)
      if (r == T1K_OK) r = t1k_reads_upload_end(rd);
      const double tUp = nowMs();
      W.distinctOf.resize(ne);
      const double tRes = nowMs();
      if (r == T1K_OK) r = t1k_reads_dedupe(rd, W.distinctOf.data(), &W.nDistinct);
      if (r != T1K_OK) { fail(r, t1k_last_error(rd)); return; }
      if (xw.x && W.deferred) {  // sequences an earlier kept window assigned are not assigned again; this window's own become known to later ones
        if ((r = t1k_xwin_link(xw.x, rd, w, &W.nExternal)) != T1K_OK) { fail(r, t1k_xwin_last_error(xw.x)); return; }
        W.linked = W.nExternal == 0;
      } else W.linked = true;
      const double tDd = nowMs();
      // (a range is sized by the read-ends it has to assign: the ones linked to earlier windows ride along for nothing)
      W.assignBatch = assignBatch;
      if (W.nExternal) W.assignBatch = (uint32_t)std::min<uint64_t>(2ull * assignBatch, ((uint64_t)assignBatch * W.nDistinct + (W.nDistinct - W.nExternal)) / std::max<uint32_t>(1u, W.nDistinct - W.nExternal));
      W.nAssign = (W.nDistinct + W.assignBatch - 1) / W.assignBatch;
      W.pairBatch = pairBatch; W.nPair = (nf + pairBatch - 1) / pairBatch;
      W.pairDone.assign(W.nPair, 0);
      W.assignDone.assign(W.nAssign, 0);
      W.pairNeed.assign(W.nPair, 0);
      {  // highest distinct read-end named by each pairing range, then the running maximum over the ranges before it
        parallelRanges(W.nPair, T, [&](int, size_t b, size_t e) {
          for (size_t q = b; q < e; ++q) {
            const size_t i0 = q * (size_t)pairBatch * per, i1 = std::min<size_t>((size_t)ne, (q + 1) * (size_t)pairBatch * per);
            uint32_t mx = 0;
            for (size_t i = i0; i < i1; ++i) mx = std::max(mx, W.distinctOf[i]);
            W.pairNeed[q] = mx / W.assignBatch + 1;
          }
        });
        for (uint32_t q = 1; q < W.nPair; ++q) W.pairNeed[q] = std::max(W.pairNeed[q], W.pairNeed[q - 1]);
        for (uint32_t q = 0; q < W.nPair; ++q) W.pairNeed[q] = std::min(W.pairNeed[q], W.nAssign);
      }
      if (traceTasks) fprintf(stderr, "[t1k task] prep window %u: %.1f .. %.1f ms (text %.1f ms, upload %.1f, resize %.1f, dedupe %.1f, pairing needs %.1f)\n", w, t0 - tStart, nowMs() - tStart, tText - t0, tUp - tText, tRes - tUp, tDd - tRes, nowMs() - tDd);
      {
        std::lock_guard<std::mutex> g(sh.m);
        job->distinctReadEnds += W.nDistinct - W.nExternal;
        job->stats.read_ends -= W.nExternal;  // (the ranges count every read-end they were given; the linked ones were not assigned)
        msPrep += nowMs() - t0;
        W.msPrep = nowMs() - t0; W.tReady = nowMs();
        W.ready = true;
      }
      sh.cv.notify_all();
    }
  };
  // ---- pipelines -------------------------------------------------------------------------------------------------------
  // AssignRead over distinct read-ends [b0, b0 + nb) of a window; on a capacity error of a stage before anything is committed the range is split
  std::function<int(t1k_ctx *, Window &, uint32_t, uint32_t, std::string &)> assignRange = [&](t1k_ctx *ctx, Window &Wn, uint32_t b0, uint32_t nb, std::string &msg) -> int {
    int r = t1k_assign_range(ctx, b0, nb);
    if (r == T1K_ERR_CAPACITY && nb > 64) {
      if ((r = assignRange(ctx, Wn, b0, nb / 2, msg)) != T1K_OK) return r;
      return assignRange(ctx, Wn, b0 + nb / 2, nb - nb / 2, msg);
    }
    if (r != T1K_OK) { msg = t1k_last_error(ctx); return r; }
    t1k_stats st;
    t1k_stats_get(ctx, &st);
    if (traceTasks) fprintf(stderr, "[t1k task]   range of %u read-ends at %u: kernels by HIP events: seed %.1f chain %.1f extend %.1f select %.1f align+truncate %.1f ms\n", nb, b0, st.ms_seed, st.ms_chain, st.ms_extend, st.ms_select, st.ms_fullalign);
    std::lock_guard<std::mutex> g(sh.m);
    Wn.ovlRecords += st.extended;
    job->stats.read_ends += st.read_ends; job->stats.lookups += st.lookups; job->stats.postings += st.postings; job->stats.hits += st.hits;
    job->stats.groups += st.groups; job->stats.candidates += st.candidates; job->stats.extended += st.extended; job->stats.near_best += st.near_best;
    job->stats.dp_calls += st.dp_calls; job->stats.ms_chain += st.ms_chain; job->stats.ms_extend += st.ms_extend; job->stats.ms_select += st.ms_select;
    job->stats.ms_fullalign += st.ms_fullalign; job->stats.ms_seed += st.ms_seed; job->stats.batches += 1; job->stats.dp_cells += st.dp_cells;
    return T1K_OK;
  };
  auto worker = [&](int pi) {
    t1k_ctx *ctx = pipes[pi];
    int attached = -1;
    std::vector<uint32_t> e1, e2;
    for (;;) {
      uint32_t w = 0, item = 0;
      int kind = -1;  // 0 assign, 1 pair, 2 copy the table entries of the read-ends an earlier window assigned
      {
        std::unique_lock<std::mutex> lk(sh.m);
        for (;;) {
          if (sh.err != T1K_OK) return;
          if (sh.allCreated && sh.oldest >= sh.created) return;
          for (w = sh.oldest; w < sh.created && w < sh.oldest + 2 && win[w].ready && kind < 0; ++w) {
            Window &W = win[w];
            if (W.nextPair < W.nPair && W.assignPrefix >= W.pairNeed[W.nextPair]) {
              if (W.linked) { kind = 1; item = W.nextPair++; }
              else if (!W.linking) {
                // its linked read-ends take their lists from earlier windows: those must have finished their assignment ranges (the windows
                // before sh.oldest are done altogether)
                bool srcDone = true;
                for (uint32_t v = sh.oldest; v < w; ++v) srcDone = srcDone && win[v].ready && win[v].doneAssign == win[v].nAssign;
                if (srcDone) { kind = 2; W.linking = true; }
              }
            }
            if (kind < 0 && W.nextAssign < W.nAssign) { kind = 0; item = W.nextAssign++; }
            if (kind >= 0) break;
          }
          if (kind >= 0) break;
          sh.cv.wait(lk);
        }
      }
      Window &W = win[w];
      int r = T1K_OK;
      std::string msg;
      const double tTask = nowMs();
      if (kind == 2) {
        r = t1k_xwin_resolve(xw.x, ctx, w);
        if (r != T1K_OK) { fail(r, t1k_xwin_last_error(xw.x)); return; }
        if (traceTasks) fprintf(stderr, "[t1k task] pipe %d window %u: %u read-ends linked to earlier windows: %.1f .. %.1f ms\n", pi, w, W.nExternal, tTask - tStart, nowMs() - tStart);
        { std::lock_guard<std::mutex> g(sh.m); W.linked = true; }
        sh.cv.notify_all();
        continue;
      }
      if (attached != (int)w) {
        r = t1k_reads_attach(ctx, job->reader[W.slot], W.slot, W.touched[pi] ? 0 : 1);
        if (r != T1K_OK) msg = t1k_last_error(ctx);
        W.touched[pi] = 1;
        attached = (int)w;
      }
      if (r == T1K_OK) (void)t1k_ctx_set_coverage_mode(ctx, (W.deferred || job->analyzer) ? 1 : 0);
      if (r == T1K_OK && kind == 0) {
        const uint32_t b0 = item * W.assignBatch, nb = std::min(W.assignBatch, W.nDistinct - b0);
        r = assignRange(ctx, W, b0, nb, msg);
      } else if (r == T1K_OK) {
        const uint32_t q0 = item * W.pairBatch, nq = std::min(W.pairBatch, (W.f1 - W.f0) - q0);
        e1.resize(nq); e2.resize(nq);
        for (uint32_t i = 0; i < nq; ++i) { e1[i] = W.distinctOf[(size_t)(q0 + i) * per]; if (per == 2) e2[i] = W.distinctOf[(size_t)(q0 + i) * per + 1]; }
        r = t1k_pair_into(ctx, job->rows, e1.data(), per == 2 ? e2.data() : nullptr, W.hasN.data() + q0, nq, (uint64_t)W.f0 + q0);
        if (r != T1K_OK) msg = t1k_last_error(ctx);
        else {
          t1k_stats st;
          t1k_stats_get(ctx, &st);
          std::lock_guard<std::mutex> g(sh.m);
          job->stats.ms_pair += st.ms_pair; job->stats.pair_overlaps += st.pair_overlaps; job->stats.rows += st.rows;
          sh.pairedFrags += nq;
        }
      }
      if (r != T1K_OK) { fail(r, msg); return; }
      if (traceTasks) fprintf(stderr, "[t1k task] pipe %d window %u %s %u: %.1f .. %.1f ms\n", pi, w, kind == 0 ? "assign" : "pair", item, tTask - tStart, nowMs() - tStart);
      {
        std::lock_guard<std::mutex> g(sh.m);
        if (kind == 0) {
          ++W.doneAssign; W.assignDone[item] = 1;
          while (W.assignPrefix < W.nAssign && W.assignDone[W.assignPrefix]) ++W.assignPrefix;
        } else { ++W.donePair; W.pairDone[item] = 1; }
        if (W.doneAssign == W.nAssign && W.donePair == W.nPair) {
          if (W.deferred) {
            // every task of the window has finished (each ends with its stream drained): its read set and the lists the pipelines
            // wrote change owner before the preparation thread may upload window w + 2 into the same context / store slot
            t1k_readset *rs = nullptr;
            if (t1k_readset_detach(job->reader[W.slot], &rs) == T1K_OK) {
              for (int q = 0; q < P; ++q)
                if (W.touched[q]) (void)t1k_readset_take_store(rs, pipes[q], W.slot);
              archivedBytes += t1k_readset_bytes(rs); archivedFrags += W.f1 - W.f0;
              archivedNeed += W.ovlRecords * 16 + (uint64_t)W.nDistinct * 220;
              job->archive.push_back(rs);
            } else if (sh.err == T1K_OK) { sh.err = T1K_ERR_INTERNAL; sh.errMsg = std::string("keeping the window's read set: ") + t1k_last_error(job->reader[W.slot]); }
          }
          W.done = true; W.tDone = nowMs();
          if (traceTasks) fprintf(stderr, "[t1k task] window %u done: fragments %u .. %u, %u distinct read-ends (%u of them assigned by earlier windows), %u + %u ranges, ready at %.1f ms, done at %.1f ms%s\n", w, W.f0, W.f1,
                                  W.nDistinct, W.nExternal, W.nAssign, W.nPair, W.tReady - tStart, W.tDone - tStart, W.deferred ? ", read set kept" : "");
          std::vector<uint32_t>().swap(W.distinctOf);
          while (sh.oldest < sh.created && win[sh.oldest].done) ++sh.oldest;
        }
      }
      sh.cv.notify_all();
    }
  };
  // ---- read files of a single-GPU job: written behind the loop, pairing range by pairing range (what is not written when the loop
  // ends is left to the writer that runs beside the EM, t1k_job_finish) ----------------------------------------------------------
  job->streamDone = 0;
  const bool streaming = !job->outPrefix.empty() && job->nRanks == 1 && !in.sharded && !job->analyzer && F > 0 && !getenv("T1K_NO_STREAM_OUTPUT");
  bool loopEnded = false;
  auto follow = [&] {
    if (!streamOpen(job, job->outPrefix)) { fail(T1K_ERR_IO, job->err); return; }  // (truncating last run's files takes a while: not on the loop's thread)
    // the writer stays right behind the pairing: a fragment's flag is final when its pairing range is done, and the ranges of a
    // window are handed out in order, so it appends every run of finished ranges as soon as it is contiguous with what is written
    for (uint32_t w = 0;; ++w) {
      {
        std::unique_lock<std::mutex> lk(sh.m);
        sh.cv.wait(lk, [&] { return sh.err != T1K_OK || loopEnded || w < sh.created || sh.allCreated; });
        if (sh.err != T1K_OK || w >= sh.created) return;  // (all windows written, or the loop ended early)
      }
      Window &W = win[w];
      uint32_t next = 0;  // first pairing range of this window that is not written yet
      for (;;) {
        uint32_t upto = next;
        {
          std::unique_lock<std::mutex> lk(sh.m);
          sh.cv.wait(lk, [&] { return sh.err != T1K_OK || loopEnded || (W.ready && (next >= W.nPair || W.pairDone[next])); });
          if (sh.err != T1K_OK) return;
          if (!W.ready) return;  // the loop ended early
          while (upto < W.nPair && W.pairDone[upto]) ++upto;
          if (upto == next && next < W.nPair) return;  // loop ended with ranges open (an error elsewhere)
        }
        if (upto == next) break;  // window complete
        const uint32_t fLo = W.f0 + next * W.pairBatch, fHi = (uint32_t)std::min<uint64_t>(W.f1, (uint64_t)W.f0 + (uint64_t)upto * W.pairBatch);
        int r = t1k_rowset_assigned_range(job->rows, fLo, fHi - fLo, job->fragAssigned.data() + fBeg + fLo);
        if (r != T1K_OK) { fail(r, t1k_rowset_last_error(job->rows)); return; }
        if (!streamAppend(job, fLo, fHi, true)) { fail(T1K_ERR_IO, job->err); return; }
        job->streamDone = fHi;
        // nobody reads these records' bytes again: unmap them here, beside the loop (6 GB of page-table entries cost 0.4 s at the end)
        if (!job->prm.output_read_assignment && fHi > fLo) job->in->release(in.frag[fLo], (size_t)in.frag[fHi - 1] + 1);
        next = upto;
      }
    }
  };
  {
    std::thread prep(prepare);
    std::thread writer;
    if (streaming) writer = std::thread(follow);
    std::vector<std::thread> others;
    for (int i = 1; i < P; ++i) others.emplace_back(worker, i);
    worker(0);
    for (auto &t : others) t.join();
    { std::lock_guard<std::mutex> g(sh.m); if (sh.err == T1K_OK && (!sh.allCreated || sh.oldest < sh.created) && F > 0) { sh.err = T1K_ERR_INTERNAL; sh.errMsg = "window loop ended early"; } loopEnded = true; }
    sh.cv.notify_all();
    prep.join();
    if (writer.joinable()) writer.join();
  }
  if (sh.err != T1K_OK) { streamClose(job, true); return jobFail(job, sh.err, sh.errMsg); }
  if (gzStream) {
    // the stream has ended and every fragment went through the loop: the tables shrink to the fragments that exist
    if (!job->in->streamFinish(job->err)) { streamClose(job, true); return jobFail(job, T1K_ERR_IO, job->err); }
    F = (uint32_t)in.nFrag();
    job->readEnds = (uint64_t)F * per;
    gt.readLength = in.maxLen;  // Genotyper.cpp:443
    if ((rc = t1k_rowset_trim(job->rows, F)) != T1K_OK) return jobFail(job, rc, "the streamed input holds more fragments than the job's tables were sized for");
    job->fragAssigned.resize(F);
  }
  if (dropLong)
    fprintf(stderr, "genotyper: WARNING: %llu fragment(s)%s hold a read longer than %u bases and were set aside (T1K_LONG_READS=drop): the reference would have genotyped them\n",
            (unsigned long long)droppedFragments.load(), job->nRanks > 1 ? " of this rank" : "", lenLimit);
  const double tDev = nowMs();
  if (getenv("T1K_DEBUG_MEM")) {  // what the contexts hold at the end of the window loop (their peak: arenas only grow), the kept read sets, the driver's view
    uint64_t sum = 0; int i = 0;
    for (t1k_ctx *c : pipes) { char tag[32]; snprintf(tag, sizeof tag, "pipeline %d", i++); sum += t1k_ctx_mem_report(c, tag, 1); }
    i = 0;
    for (t1k_ctx *c : job->reader) { char tag[32]; snprintf(tag, sizeof tag, "read-set context %d", i++); sum += t1k_ctx_mem_report(c, tag, 1); }
    uint64_t fr = 0, tt = 0;
    (void)t1k_device_memory(job->prm.device, &fr, &tt);
    fprintf(stderr, "[t1k mem] contexts %.1f GB; device: %.1f of %.1f GB free (cached pool blocks counted as free)\n", sum / 1073741824.0, fr / 1073741824.0, tt / 1073741824.0);
  }
  for (t1k_ctx *c : job->more)
    if ((rc = t1k_coverage_absorb(job->ctx, c)) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  if (job->analyzer) {  // the per-barcode summary reads the fragment assignment lists themselves (t1k_analyzer_main)
    if ((rc = t1k_rowset_assigned_download(job->rows, job->fragAssigned.data())) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
    job->msDevice = tDev - tStart;
    job->localDone = true;
    return T1K_OK;
  }
  const bool sharded = job->nRanks > 1;
  if (sharded && !job->covDeferred) {  // per-base coverage of all ranks: integers, exact in any order (a deferred job reduces inside select())
    void *cov = nullptr; uint64_t covN = 0;
    if ((rc = t1k_coverage_device(job->ctx, &cov, &covN)) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
    if ((rc = t1k_comm_allreduce(job->comm, cov, covN, 0)) != T1K_OK) return jobFail(job, rc, t1k_comm_last_error(job->comm));
  }
  if (job->prm.output_read_assignment) {  // Genotyper.cpp:553-560: the rows in the reference's order, before coalescing
    const uint32_t step = 1u << 18;
    std::vector<uint32_t> cnt;
    std::vector<t1k_row_entry> rows;
    char num[64];
    for (uint32_t f0 = 0; f0 < F; f0 += step) {
      const uint32_t n = std::min(step, F - f0);
      cnt.resize(n);
      uint64_t total = 0;
      if ((rc = t1k_rowset_rows_download(job->rows, f0, n, cnt.data(), nullptr, 0, &total)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
      rows.resize(total);
      if (total && (rc = t1k_rowset_rows_download(job->rows, f0, n, cnt.data(), rows.data(), total, &total)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
      uint64_t p = 0;
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t r = in.frag[fBeg - inBase + f0 + i];
        const std::string id = in.noIds ? "r" + std::to_string(fBeg + f0 + i) : std::string(in.side[0].idP[r], in.side[0].idL[r]);
        for (uint32_t j = 0; j < cnt[i]; ++j, ++p) {
          job->assignText += id; job->assignText += '\t'; job->assignText += job->ref.al[rows[p].allele_idx].name;
          snprintf(num, sizeof(num), "\t%d\t%d\n", rows[p].start, rows[p].end);
          job->assignText += num;
        }
      }
    }
    if (job->nRanks > 1) {  // the table of all fragments = the ranks' tables in rank order (= fragment order); rank 0 writes it
      std::vector<uint64_t> sizes(job->nRanks, 0), eight(job->nRanks, 8), at8(job->nRanks);
      for (int r = 0; r < job->nRanks; ++r) at8[r] = 8 * (uint64_t)r;
      sizes[job->rank] = job->assignText.size();
      if ((rc = t1k_comm_allgatherv_host(job->comm, sizes.data(), eight.data(), at8.data(), 8 * (uint64_t)job->nRanks)) != T1K_OK) return jobFail(job, rc, t1k_comm_last_error(job->comm));
      std::vector<uint64_t> displ(job->nRanks, 0);
      uint64_t total = 0;
      for (int r = 0; r < job->nRanks; ++r) { displ[r] = total; total += sizes[r]; }
      std::string all(total, '\0');
      if (!job->assignText.empty()) memcpy(&all[displ[job->rank]], job->assignText.data(), job->assignText.size());
      if ((rc = t1k_comm_allgatherv_host(job->comm, total ? &all[0] : nullptr, sizes.data(), displ.data(), total)) != T1K_OK) return jobFail(job, rc, t1k_comm_last_error(job->comm));
      job->assignText.swap(all);
    }
  }
  // ---- CoalesceReadAssignments over all fragments (t1k_coalesce.hip), groups back to the host ---------------------------
  // Sharded: every row first travels to the rank that owns its pattern, which folds the group over ALL its fragments in global
  // order; the owners' tables are gathered on every rank and merged by first fragment (= first-appearance numbering).
  uint64_t G = 0, N = 0, assigned = 0;
  static_assert(sizeof(GroupEntry) == sizeof(t1k_group_entry), "group entry layouts differ");
  const double tEx0 = nowMs();
  if (getenv("T1K_DEBUG_ARCHIVE")) {
    uint64_t b = 0, e = 0;
    if (t1k_rowset_device_bytes(job->rows, &b, &e) == T1K_OK)
      fprintf(stderr, "[t1k job] after the loop: %llu row entries in the chunks (%.1f GB allocated for rows), %llu fragments paired, stats.rows %llu\n",
              (unsigned long long)e, b / 1e9, (unsigned long long)sh.pairedFrags, (unsigned long long)job->stats.rows);
  }
  if (sharded && (rc = t1k_rowset_exchange(job->rows, job->comm, fBeg)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
  const double tEx1 = nowMs();
  // (one GPU: the host tables of the groups are sized and their pages first touched by all host threads -- 100 k page faults on one
  // thread were 90 ms, 27 ms on all -- while the device is still folding the groups)
  struct Sizer { Genotyper *gt; int T; bool on; } sizer{&gt, T, !sharded};
  auto sizeTables = [](uint64_t g, uint64_t n, void *u) {
    Sizer &z = *(Sizer *)u;
    if (!z.on) return;
    z.gt->groupPtr.assign(g + 1, 0);
    z.gt->groupEnt.resize(n);  // (not zeroed: GroupVec)
    z.gt->groupFirst.resize(g);
    parallelRanges(n * sizeof(GroupEntry) / 4096 + 1, z.T, [&](int, size_t b, size_t e) {
      volatile char *base = (volatile char *)z.gt->groupEnt.data();
      const size_t bytes = n * sizeof(GroupEntry);
      for (size_t pg = b; pg < e; ++pg) if (pg * 4096 < bytes) base[pg * 4096] = 0;
    });
  };
  if ((rc = t1k_rowset_coalesce_sized(job->rows, &G, &N, &assigned, sizeTables, &sizer)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
  const double tCo = nowMs();
  if (!sharded) {
    if (gt.groupPtr.size() != G + 1) { gt.groupPtr.assign(G + 1, 0); gt.groupEnt.resize(N); gt.groupFirst.resize(G); }  // (no row at all: the callback was not called)
    const double tRes = nowMs();
    if ((rc = t1k_rowset_groups_download(job->rows, gt.groupPtr.data(), (t1k_group_entry *)gt.groupEnt.data(), gt.groupFirst.data())) != T1K_OK)
      return jobFail(job, rc, t1k_rowset_last_error(job->rows));
    const double tDl = nowMs();
    if ((rc = t1k_rowset_assigned_download(job->rows, job->fragAssigned.data())) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
    if (getenv("T1K_DEBUG_PHASES"))
      fprintf(stderr, "[t1k job] after the loop: coverage of the pipelines + coalescing on the device %.1f ms, host tables sized %.1f ms, groups downloaded %.1f ms, flags %.1f ms\n",
              tCo - tDev, tRes - tCo, tDl - tRes, nowMs() - tDl);
  } else {
    const uint64_t Gl = G, Nl = N;
    if ((rc = t1k_rowset_groups_gather(job->rows, job->comm, &G, &N, &assigned)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
    std::vector<uint32_t> sizes(G), first(G);
    GroupVec ents;
    ents.resize(N);  // (not zeroed; pages first touched by all host threads, as for the single-GPU table above)
    parallelRanges(N * sizeof(GroupEntry) / 4096 + 1, T, [&](int, size_t b, size_t e) {
      volatile char *base = (volatile char *)ents.data();
      const size_t bytes = N * sizeof(GroupEntry);
      for (size_t pg = b; pg < e; ++pg) if (pg * 4096 < bytes) base[pg * 4096] = 0;
    });
    if ((rc = t1k_rowset_groups_download_all(job->rows, sizes.data(), (t1k_group_entry *)ents.data(), first.data())) != T1K_OK)
      return jobFail(job, rc, t1k_rowset_last_error(job->rows));
    const double tGa = nowMs();
    gt.setGroupsMerged(sizes, ents, first);
    if (getenv("T1K_DEBUG_PHASES"))
      fprintf(stderr, "[t1k job] rank %d of %d after the loop (waits for the slowest rank included): rows to their pattern owners %.1f ms, coalescing of the owned patterns %.1f ms "
                      "(%llu groups, %llu entries on this rank), group tables gathered + downloaded %.1f ms (%llu groups, %llu entries in all), merged by first fragment %.1f ms\n",
              job->rank, job->nRanks, tEx1 - tEx0, tCo - tEx1, (unsigned long long)Gl, (unsigned long long)Nl, tGa - tCo,
              (unsigned long long)G, (unsigned long long)N, nowMs() - tGa);
    // fragmentAssigned of every rank's slice on every rank (rank 0 writes the *_aligned*.fa files)
    if ((rc = t1k_rowset_assigned_download(job->rows, job->fragAssigned.data() + fBeg)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
    if (!in.sharded) {  // (ranks that indexed only their own reads write only their own part of the files)
      std::vector<uint64_t> bytes(job->nRanks), displ(job->nRanks);
      for (int r = 0; r < job->nRanks; ++r) { displ[r] = (uint64_t)Fall * r / job->nRanks; bytes[r] = (uint64_t)Fall * (r + 1) / job->nRanks - displ[r]; }
      if ((rc = t1k_comm_allgatherv_host(job->comm, job->fragAssigned.data(), bytes.data(), displ.data(), Fall)) != T1K_OK) return jobFail(job, rc, t1k_comm_last_error(job->comm));
    }
  }
  gt.assignedFragments = assigned;
  job->stats.read_ends_total = job->readEnds;
  t1k_rowset_destroy(job->rows);
  job->rows = nullptr;
  job->msDevice = tDev - tStart; job->msCoalesce = nowMs() - tDev; job->msHost = 0;
  job->stats.ms_load = job->msLoad; job->stats.ms_device = job->msDevice; job->stats.ms_coalesce = job->msCoalesce;
  if (getenv("T1K_DEBUG_PHASES")) {
    double ms = 0; uint64_t by = 0, b1 = 0;
    for (t1k_ctx *c : pipes) { ms += t1k_alloc_ms(c, &b1); by += b1; }
    for (t1k_ctx *c : job->reader) { ms += t1k_alloc_ms(c, &b1); by += b1; }
    fprintf(stderr, "[t1k job] device memory: %.1f GB allocated by the contexts in %.1f ms of hipMalloc (summed over threads)\n", by / 1073741824.0, ms);
  }
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k job] %u windows, %llu read-ends -> %llu distinct; window preparation %.1f ms (overlapped), device loop %.1f ms, coalesce + download %.1f ms (%llu groups, %llu entries)\n",
            sh.created, (unsigned long long)job->readEnds, (unsigned long long)job->distinctReadEnds, msPrep, job->msDevice, job->msCoalesce, (unsigned long long)G, (unsigned long long)N);
  if (getenv("T1K_DEBUG_PHASES") && job->covDeferred)
    fprintf(stderr, "[t1k job] coverage deferred to selection: read sets of %zu of %u windows kept (%.2f GB of device memory)\n", job->archive.size(), sh.created, archivedBytes / 1073741824.0);
  job->localDone = true;
  return T1K_OK;
}

// ">id\nSEQ\n" of every assigned fragment (Genotyper.cpp:680-718), formatted by the host threads straight from the mapped input and
// written with pwrite at precomputed offsets; what = 0 / 1: the mate's sequence, 2: the barcode.
// Two steps: the plan (bytes per host-thread piece; for ranks that each indexed their own reads also the rank's offset in the
// shared file -- one small all-gather, and rank 0 creates the file before it) and the writing itself, which needs no communication
// and so may run beside the EM.
namespace {
struct AlignedPlan {
  std::string path;
  int what = 0, T = 1;
  std::vector<uint64_t> pieceBytes;  // exclusive prefix over the T pieces of this rank's fragments
  uint64_t baseOffset = 0;           // of this rank's part in the file
  bool create = true;                // this rank truncates / creates the file (done in the plan step when the job's input is sharded)
};
}  // namespace

// bytes of ">id\nSEQ\n" of the assigned fragments among the local fragments [fLo, fHi), as an exclusive prefix over T pieces
static void alignedSizes(t1k_job *job, int what, uint32_t fLo, uint32_t fHi, int T, std::vector<uint64_t> &pieceBytes) {
  const ReadInput &in = *job->in;
  const uint32_t base = in.base;
  const ReadInput::Side &seqSide = what == 2 ? in.bc : in.side[what];
  const ReadInput::Side &idSide = what == 1 ? in.side[1] : in.side[0];  // the barcode file carries mate 1's name (Genotyper.cpp:709-718)
  pieceBytes.assign(T + 2, 0);
  parallelRanges(fHi - fLo, T, [&](int t, size_t b, size_t e) {
    uint64_t run = 0;
    char tmp[32];
    for (size_t f = fLo + b; f < fLo + e; ++f)
      if (job->fragAssigned[base + f]) {
        const uint32_t r = in.frag[f];
        run += 3 + (in.noIds ? (size_t)snprintf(tmp, 32, "r%u", (uint32_t)(base + f)) : (size_t)idSide.idL[r]) + seqSide.seqL[r];
      }
    pieceBytes[t + 1] = run;
  });
  for (int t = 0; t < T + 1; ++t) pieceBytes[t + 1] += pieceBytes[t];
}

// ... and the records themselves, piece t at offset + pieceBytes[t].
// Buffered writes to one file take the inode lock one at a time, so a file fills at the speed of one copying thread however many
// threads format records.  When this process is the file's only writer (mapped == true) the byte range is reserved with fallocate --
// a full disk is reported here, not as a fault later -- and mapped, and the threads format straight into the page cache in parallel;
// anything the file system refuses falls back to pwrite.
static bool alignedWrite(t1k_job *job, int fd, int what, uint32_t fLo, uint32_t fHi, int T, const std::vector<uint64_t> &pieceBytes, uint64_t offset, bool mapped) {
  const ReadInput &in = *job->in;
  const uint32_t base = in.base;
  const ReadInput::Side &seqSide = what == 2 ? in.bc : in.side[what];
  const ReadInput::Side &idSide = what == 1 ? in.side[1] : in.side[0];
  const uint64_t total = pieceBytes[T];
  if (mapped && total >= (1u << 20) && fallocate(fd, 0, (off_t)offset, (off_t)total) == 0) {
    const uint64_t pg = (uint64_t)sysconf(_SC_PAGESIZE), a0 = offset & ~(pg - 1);
    void *m = mmap(nullptr, (size_t)(offset + total - a0), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)a0);
    if (m != MAP_FAILED) {
      char *out = (char *)m + (offset - a0);
      parallelRanges(fHi - fLo, T, [&](int t, size_t b, size_t e) {
        char *at = out + pieceBytes[t];
        char tmp[32];
        for (size_t f = fLo + b; f < fLo + e; ++f) {
          if (!job->fragAssigned[base + f]) continue;
          const uint32_t r = in.frag[f];
          *at++ = '>';
          if (in.noIds) { const int n = snprintf(tmp, 32, "r%u", (uint32_t)(base + f)); memcpy(at, tmp, (size_t)n); at += n; }
          else { memcpy(at, idSide.idP[r], idSide.idL[r]); at += idSide.idL[r]; }
          *at++ = '\n';
          memcpy(at, seqSide.seqP[r], seqSide.seqL[r]); at += seqSide.seqL[r];
          *at++ = '\n';
        }
      });
      munmap(m, (size_t)(offset + total - a0));
      return true;
    }
  }
  std::atomic<bool> ok{true};
  parallelRanges(fHi - fLo, T, [&](int t, size_t b, size_t e) {
    uint64_t at = offset + pieceBytes[t];
    std::vector<char> buf;
    buf.reserve(8u << 20);
    char tmp[32];
    auto flush = [&] {
      size_t done = 0;
      while (done < buf.size()) {
        ssize_t w = pwrite(fd, buf.data() + done, buf.size() - done, (off_t)(at + done));
        if (w <= 0) { ok = false; break; }
        done += (size_t)w;
      }
      at += buf.size();
      buf.clear();
    };
    for (size_t f = fLo + b; f < fLo + e; ++f) {
      if (!job->fragAssigned[base + f]) continue;
      const uint32_t r = in.frag[f];
      buf.push_back('>');
      if (in.noIds) { const int n = snprintf(tmp, 32, "r%u", (uint32_t)(base + f)); buf.insert(buf.end(), tmp, tmp + n); }
      else buf.insert(buf.end(), idSide.idP[r], idSide.idP[r] + idSide.idL[r]);
      buf.push_back('\n');
      buf.insert(buf.end(), seqSide.seqP[r], seqSide.seqP[r] + seqSide.seqL[r]); buf.push_back('\n');
      if (buf.size() > (7u << 20)) flush();
    }
    flush();
  });
  return ok;
}

static bool mappedOutput() { static const bool on = getenv("T1K_NO_MMAP_OUTPUT") == nullptr; return on; }
static bool planAligned(t1k_job *job, AlignedPlan &pl) {
  const ReadInput &in = *job->in;
  const int T = pl.T;
  alignedSizes(job, pl.what, 0, (uint32_t)in.nFrag(), T, pl.pieceBytes);
  pl.baseOffset = 0; pl.create = true;
  if (in.sharded) {
    pl.create = false;
    if (job->rank == 0) {
      ::unlink(pl.path.c_str());  // (see streamOpen: a truncated-and-rewritten file is flushed when it is closed)
      FILE *fp = fopen(pl.path.c_str(), "w");
      if (!fp) { job->err = "cannot write " + pl.path; return false; }
      fclose(fp);
    }
    std::vector<uint64_t> sizes(job->nRanks, 0), bytes(job->nRanks, 8), displ(job->nRanks);
    for (int r = 0; r < job->nRanks; ++r) displ[r] = 8 * (uint64_t)r;
    sizes[job->rank] = pl.pieceBytes[T];
    if (t1k_comm_allgatherv_host(job->comm, sizes.data(), bytes.data(), displ.data(), 8 * (uint64_t)job->nRanks) != T1K_OK) { job->err = t1k_comm_last_error(job->comm); return false; }
    for (int r = 0; r < job->rank; ++r) pl.baseOffset += sizes[r];
  }
  return true;
}

static bool writeAligned(t1k_job *job, const AlignedPlan &pl) {
  if (pl.create) ::unlink(pl.path.c_str());
  const int fd = ::open(pl.path.c_str(), pl.create ? (O_RDWR | O_CREAT | O_TRUNC) : O_WRONLY, 0644);
  if (fd < 0) { job->err = "cannot write " + pl.path; return false; }
  const bool ok = alignedWrite(job, fd, pl.what, 0, (uint32_t)job->in->nFrag(), pl.T, pl.pieceBytes, pl.baseOffset, pl.create && mappedOutput());
  ::close(fd);
  if (!ok) { job->err = "cannot write " + pl.path; return false; }
  return true;
}

// A single-GPU job writes the read files while the device loop is still running: the writer follows the windows of the loop
// (their fragment flags are final once the window's pairing tasks are done) and appends each window's records.
static bool streamOpen(t1k_job *job, const std::string &pfx) {
  const bool paired = job->in->paired;
  job->stream.clear();
  auto add = [&](const std::string &path, int what) {
    t1k_job::StreamOut o; o.path = path; o.what = what;
    // a file of an earlier run goes first: ext4 (auto_da_alloc) flushes a file that was truncated and rewritten when it is closed --
    // 0.3 s per 1.6 GB file at the end of the job -- while a newly created one just stays in the page cache like the reference's fclose
    ::unlink(path.c_str());
    o.fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    job->stream.push_back(o);
    return o.fd >= 0;
  };
  bool ok = add(paired ? pfx + "_aligned_1.fa" : pfx + "_aligned.fa", 0);
  if (ok && paired) ok = add(pfx + "_aligned_2.fa", 1);
  if (ok && job->in->hasBarcode) ok = add(pfx + "_aligned_bc.fa", 2);
  if (!ok) job->err = "cannot write " + job->stream.back().path;
  return ok;
}
static void streamClose(t1k_job *job, bool removeFiles) {
  for (auto &o : job->stream) {
    if (o.fd >= 0) ::close(o.fd);
    if (removeFiles) ::unlink(o.path.c_str());
  }
  job->stream.clear();
}
// local fragments [fLo, fHi): flags must be in job->fragAssigned
static bool streamAppend(t1k_job *job, uint32_t fLo, uint32_t fHi, bool besideLoop) {
  // behind the device loop a few threads per file keep up (the pairing of a large window hands over 2 - 3 GB of records in about a
  // second); the rest of the machine feeds the GPU
  const int T = besideLoop ? 6 : std::max(1, hostThreads(job) / (int)std::max<size_t>(1, job->stream.size()));
  std::vector<char> ok(job->stream.size(), 1);
  auto one = [&](size_t i) {
    auto &o = job->stream[i];
    std::vector<uint64_t> pieceBytes;
    alignedSizes(job, o.what, fLo, fHi, T, pieceBytes);
    ok[i] = alignedWrite(job, o.fd, o.what, fLo, fHi, T, pieceBytes, o.offset, !besideLoop && mappedOutput()) ? 1 : 0;
    o.offset += pieceBytes[T];
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < job->stream.size(); ++i) th.emplace_back(one, i);
  one(0);
  for (auto &t : th) t.join();
  for (size_t i = 0; i < ok.size(); ++i)
    if (!ok[i]) { job->err = "cannot write " + job->stream[i].path; return false; }
  return true;
}

// reads with at least one fragment assignment (Genotyper.cpp:680-718): the mates' files and the barcode file
static bool planAlignedFiles(t1k_job *job, const std::string &pfx, std::vector<AlignedPlan> &plans) {
  const int T = hostThreads(job);
  const bool paired = job->in->paired;
  const int per = std::max(1, T / (1 + (paired ? 1 : 0) + (job->in->hasBarcode ? 1 : 0)));
  plans.clear();
  auto add = [&](const std::string &path, int what) { AlignedPlan pl; pl.path = path; pl.what = what; pl.T = per; plans.push_back(pl); };
  add(paired ? pfx + "_aligned_1.fa" : pfx + "_aligned.fa", 0);
  if (paired) add(pfx + "_aligned_2.fa", 1);
  if (job->in->hasBarcode) add(pfx + "_aligned_bc.fa", 2);
  for (auto &pl : plans)
    if (!planAligned(job, pl)) return false;
  return true;
}
static bool writePlannedFiles(t1k_job *job, const std::vector<AlignedPlan> &plans) {
  std::vector<char> ok(plans.size(), 1);
  std::vector<std::thread> th;
  for (size_t i = 1; i < plans.size(); ++i) th.emplace_back([&, i] { ok[i] = writeAligned(job, plans[i]) ? 1 : 0; });
  ok[0] = writeAligned(job, plans[0]) ? 1 : 0;
  for (auto &t : th) t.join();
  for (char o : ok) if (!o) return false;
  return true;
}
// who writes: rank 0 when every rank holds the whole input; every rank its own part when each indexed only its own reads
static bool writesAligned(const t1k_job *job) { return job->rank == 0 || (job->in && job->in->sharded); }


int t1k_job_set_output_prefix(t1k_job *job, const char *prefix) {
  if (!job) return T1K_ERR_ARG;
  job->outPrefix = prefix ? prefix : "";
  return T1K_OK;
}

int t1k_job_finish(t1k_job *job) {
  if (!job || !job->ctx || !job->localDone) return jobFail(job, T1K_ERR_STATE, "t1k_job_finish: t1k_job_run_local has not completed");
  if (job->bgWriter.joinable()) job->bgWriter.join();
  job->bgStarted = false; job->bgOk = true;
  if (!job->stream.empty()) {  // the read files were started behind the device loop: the rest of the fragments now, beside the EM
    job->bgStarted = true;
    job->bgWriter = std::thread([job] {
      const double t0 = nowMs();
      const uint32_t from = job->streamDone;
      job->bgOk = streamAppend(job, job->streamDone, (uint32_t)job->in->nFrag(), false);
      const double t1 = nowMs();
      streamClose(job, !job->bgOk);
      if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k job] read files: fragments %u .. %u written after the loop in %.1f ms, files closed in %.1f ms\n", from, (uint32_t)job->in->nFrag(), t1 - t0, nowMs() - t1);
    });
  } else if (!job->outPrefix.empty() && writesAligned(job) && !job->analyzer) {  // the flags are final: start on the big files now
    std::vector<AlignedPlan> plans;
    if (!planAlignedFiles(job, job->outPrefix, plans)) return T1K_ERR_IO;
    job->bgStarted = true;
    job->bgWriter = std::thread([job, plans] { job->bgOk = writePlannedFiles(job, plans); });
  }
  Genotyper &gt = job->gt;
  int rc;
  double t2 = nowMs();
  std::vector<int32_t> cov(job->ref.al.size(), 0);  // per allele: exon positions with too little coverage
  gt.missingCoverageHook = nullptr;
  int hookRc = T1K_OK;
  if (!job->covDeferred) {
    if ((rc = t1k_missing_coverage(job->ctx, cov.data())) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  } else {
    // the value is read for the alleles on selection's candidate lists only (Genotyper.hpp:1754, 1870-1878): select() asks for them
    gt.missingCoverageHook = [job, &hookRc](const std::vector<int> &need) {
      if (need.empty()) return true;  // (the same on every rank: selection is replicated)
      const double t0 = nowMs();
      std::vector<uint8_t> sel(job->ref.al.size(), 0);
      for (int a : need) sel[a] = 1;
      // every kept read set stays alive until ALL of them are scanned: a later window's list table holds addresses inside the overlap-store
      // chunks of the earlier windows whose lists it shares (t1k_xwin_resolve), and those chunks belong to the earlier sets
      for (t1k_readset *rs : job->archive) {
        uint64_t n = 0;
        if ((hookRc = t1k_coverage_selected(job->ctx, rs, sel.data(), &n)) != T1K_OK) { job->err = t1k_last_error(job->ctx); return false; }
        job->coverRecords += n;
      }
      for (t1k_readset *&rs : job->archive) { t1k_readset_destroy(rs); rs = nullptr; }
      job->archive.clear();
      if (job->nRanks > 1) {  // per-base coverage of all ranks: integers, exact in any order
        void *dcov = nullptr; uint64_t covN = 0;
        if ((hookRc = t1k_coverage_device(job->ctx, &dcov, &covN)) != T1K_OK) { job->err = t1k_last_error(job->ctx); return false; }
        if ((hookRc = t1k_comm_allreduce(job->comm, dcov, covN, 0)) != T1K_OK) { job->err = t1k_comm_last_error(job->comm); return false; }
      }
      std::vector<int32_t> miss(job->ref.al.size());
      if ((hookRc = t1k_missing_coverage(job->ctx, miss.data())) != T1K_OK) { job->err = t1k_last_error(job->ctx); return false; }
      for (int a : need) job->ref.al[a].missingCov = miss[a];
      job->msCover = nowMs() - t0;
      if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k job] coverage of the %zu alleles on selection's lists: %llu records aligned, %.1f ms\n", need.size(), (unsigned long long)job->coverRecords, job->msCover);
      return true;
    };
  }
  gt.finalize(cov);
  double t3 = nowMs();
  if (!job->abundanceFile.empty()) {
    if (!loadAbundance(job)) return T1K_ERR_IO;
  } else {
    if (gt.quantify(job->ctx, job->comm, job->err) < 0) return T1K_ERR_DEVICE;
  }
  double t4 = nowMs();
  gt.dropUnlikely();
  double t4b = nowMs();
  gt.select();
  gt.missingCoverageHook = nullptr;
  if (gt.hookFailed) return jobFail(job, hookRc != T1K_OK ? hookRc : T1K_ERR_INTERNAL, job->err);
  double t5 = nowMs();
  job->msHost += (t3 - t2) + (t5 - t4); job->msEm = t4 - t3;
  job->stats.ms_total = job->msLoad + job->msDevice + job->msCoalesce + job->msHost + job->msEm;
  job->stats.ms_em = job->msEm;
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k job] device+download %.1f ms, host coalesce+finalize %.1f ms, EM %.1f ms, dropUnlikely %.1f ms, select %.1f ms\n", job->msDevice,
            job->msHost, t4 - t3, t4b - t4, t5 - t4b);
  job->ran = true;
  return T1K_OK;
}

int t1k_job_run(t1k_job *job) {
  int rc = t1k_job_run_local(job);
  // A streamed .gz input whose text left the layout the streaming reader follows behind the head it checked (a blank line between two
  // records, a last record without its quality line, reads that get much shorter than the head's): the whole-file reader takes such
  // text as the reference's reader does (kseq.h:94-150), so the files are opened whole and the job starts over -- nothing of the
  // failed run is kept (its partial read files are truncated by the new run).
  if (rc != T1K_OK && job && job->in && job->in->streamGaveUp.load() && job->nRanks == 1) {
    fprintf(stderr, "[t1k] %s -- the read files are opened whole and the job starts over\n", job->err.c_str());
    std::unique_ptr<ReadInput> whole(new ReadInput());
    const std::vector<std::string> f1 = job->in->streamFiles1, f2 = job->in->streamFiles2;
    const std::string bc = job->in->streamBarcodeFile;
    const bool drop = job->in->dropInflatedText;
    job->in.reset();  // (its text reservations first: the whole reader inflates the files again)
    const double t0 = nowMs();
    std::string err;
    if (!whole->open(f1, f2, bc, hostThreads(job), err)) return jobFail(job, T1K_ERR_IO, err);
    whole->dropInflatedText = drop;
    job->in = std::move(whole);
    job->ran = false; job->localDone = false;
    job->msLoad = nowMs() - t0;
    rc = t1k_job_run_local(job);
  }
  if (rc == T1K_OK) rc = t1k_job_finish(job);
  // a rank of a sharded job that fails tells the others (they would wait for it in the next exchange otherwise)
  if (rc != T1K_OK && job && job->comm && job->nRanks > 1) (void)t1k_comm_abort(job->comm);
  return rc;
}

// group table <-> byte string: [u64 nGroups][u64 nEntries][u64 assignedFragments][u64 groupPtr[nGroups+1]][u32 firstFragment[nGroups]][GroupEntry entries[nEntries]]
int t1k_job_groups_serialize(t1k_job *job, void *buf, uint64_t cap, uint64_t *needed) {
  if (!job) return T1K_ERR_ARG;
  const Genotyper &gt = job->gt;
  const uint64_t G = gt.nGroups(), N = gt.groupEnt.size();
  const uint64_t bytes = 24 + (G + 1) * 8 + G * 4 + N * sizeof(GroupEntry);
  if (needed) *needed = bytes;
  if (!buf) return T1K_OK;
  if (cap < bytes) return jobFail(job, T1K_ERR_ARG, "group buffer too small");
  if (gt.groupFirst.size() != G) return jobFail(job, T1K_ERR_STATE, "group table without first fragments");
  uint8_t *p = (uint8_t *)buf;
  uint64_t head[3] = {G, N, gt.assignedFragments};
  memcpy(p, head, 24); p += 24;
  memcpy(p, gt.groupPtr.data(), (G + 1) * 8); p += (G + 1) * 8;
  if (G) memcpy(p, gt.groupFirst.data(), G * 4);
  p += G * 4;
  if (N) memcpy(p, gt.groupEnt.data(), N * sizeof(GroupEntry));
  return T1K_OK;
}

// The host half of the multi-GPU merge: the group tables of all pattern owners (serialized as above; every pattern lives in exactly
// one of them) become this job's table, groups ordered by their first fragment.
int t1k_job_groups_merge(t1k_job *job, const void *const *bufs, const uint64_t *lens, uint32_t n) {
  if (!job || !bufs || !lens) return T1K_ERR_ARG;
  std::vector<uint32_t> sizes, first;
  GroupVec ents;
  uint64_t assigned = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint8_t *p = (const uint8_t *)bufs[i];
    if (!p || lens[i] < 24) return jobFail(job, T1K_ERR_ARG, "truncated group table");
    uint64_t head[3];
    memcpy(head, p, 24);
    const uint64_t G = head[0], N = head[1];
    if (lens[i] < 24 + (G + 1) * 8 + G * 4 + N * sizeof(GroupEntry)) return jobFail(job, T1K_ERR_ARG, "truncated group table");
    assigned += head[2];
    std::vector<uint64_t> gp(G + 1);
    memcpy(gp.data(), p + 24, (G + 1) * 8);
    const size_t g0 = sizes.size(), e0 = ents.size();
    sizes.resize(g0 + G); first.resize(g0 + G); ents.resize(e0 + N);
    for (uint64_t g = 0; g < G; ++g) sizes[g0 + g] = (uint32_t)(gp[g + 1] - gp[g]);
    if (G) memcpy(first.data() + g0, p + 24 + (G + 1) * 8, G * 4);
    if (N) memcpy(ents.data() + e0, p + 24 + (G + 1) * 8 + G * 4, N * sizeof(GroupEntry));
  }
  job->gt.setGroupsMerged(sizes, ents, first);
  job->gt.assignedFragments = assigned;
  return T1K_OK;
}

// host-side CoalesceReadAssignments on caller-provided fragment rows, in order; fragments[i] = global index of fragment i (NULL: 0, 1, ...)
int t1k_job_coalesce_rows(t1k_job *job, const t1k_row_entry *rows, const uint32_t *rowCounts, const uint32_t *fragments, uint32_t nFragments) {
  if (!job || !rowCounts || (!rows && nFragments)) return T1K_ERR_ARG;
  std::vector<t1k_row_entry> tmp;
  uint64_t p = 0;
  for (uint32_t f = 0; f < nFragments; ++f) {
    tmp.assign(rows + p, rows + p + rowCounts[f]);
    p += rowCounts[f];
    job->gt.coalesce(tmp.data(), (uint32_t)tmp.size(), fragments ? fragments[f] : f);
  }
  return T1K_OK;
}

// ---- novel-variant calling of the analyzer stage (host/variants.cpp) behind the C ABI ---------------------------------------------
struct t1k_variants {
  std::unique_ptr<VariantCaller> vc;
  const RefSet *ref = nullptr;
};

static bool variantInputOk(const RefSet &ref, const t1k_frag_assignment &a, uint32_t l1, uint32_t l2, bool haveR2) {
  // the windows must lie inside the allele and the read they name (the reference trusts its own lists; this entry point has callers)
  if (a.allele_idx < 0 || (size_t)a.allele_idx >= ref.seqs.size()) return false;
  const int L = (int)ref.seqs[a.allele_idx].size();
  for (int k = 0; k < (a.has_mate_pair ? 2 : 1); ++k) {
    const t1k_overlap &o = k ? a.o2 : a.o1;
    const bool second = k == 1 || a.o1_from_r2;
    if (second && !haveR2) return false;
    const int len = (int)(second ? l2 : l1);
    if (o.seq_idx != a.allele_idx || (o.strand != 1 && o.strand != -1)) return false;
    if (o.seq_start < 0 || o.seq_end < o.seq_start - 1 || o.seq_end >= L) return false;
    if (o.read_start < 0 || o.read_end < o.read_start - 1 || o.read_end >= len) return false;
  }
  return true;
}
// the edit string must spell exactly the two windows (columns that consume an allele base / a read base)
static bool variantOpsOk(const t1k_overlap &o, const int8_t *e, uint32_t n) {
  int64_t t = 0, p = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (e[i] < 0 || e[i] > 3) return false;
    if (e[i] != 2) ++t;
    if (e[i] != 3) ++p;
  }
  return t == (int64_t)o.seq_end - o.seq_start + 1 && p == (int64_t)o.read_end - o.read_start + 1;
}

int t1k_fragment_details(const t1k_overlap *l1, uint32_t n1, const t1k_overlap *l2, uint32_t n2, int paired, const int32_t *alleles, uint32_t nAlleles,
                         t1k_frag_assignment *out) {
  if ((n1 && !l1) || (paired && n2 && !l2) || (nAlleles && (!alleles || !out))) return T1K_ERR_ARG;
  return fragmentDetails(l1, n1, l2, paired ? n2 : 0, paired != 0, alleles, nAlleles, out) ? T1K_OK : T1K_ERR_ARG;
}

int t1k_variants_call(t1k_job *job, const double *abundance, int32_t var_max_group, uint32_t nFragments, const uint64_t *asgPtr, const t1k_frag_assignment *asg,
                      const int8_t *ops, const char *const *read1, const uint32_t *len1, const char *const *read2, const uint32_t *len2, t1k_variants **out) {
  if (!job || !out || !abundance || (nFragments && (!asgPtr || !read1 || !len1)) || ((read2 == nullptr) != (len2 == nullptr))) return T1K_ERR_ARG;
  *out = nullptr;
  const RefSet &ref = job->ref;
  std::vector<VariantCaller::Fragment> frags(nFragments);
  for (uint32_t f = 0; f < nFragments; ++f) {
    VariantCaller::Fragment &fr = frags[f];
    if (asgPtr[f + 1] < asgPtr[f]) return jobFail(job, T1K_ERR_ARG, "t1k_variants_call: asgPtr is not ascending");
    fr.asg = asg + asgPtr[f];
    fr.n = (uint32_t)(asgPtr[f + 1] - asgPtr[f]);
    fr.r1 = read1[f]; fr.l1 = len1[f];
    if (read2) { fr.r2 = read2[f]; fr.l2 = len2[f]; }
    for (uint32_t i = 0; i < fr.n; ++i) {
      const t1k_frag_assignment &a = fr.asg[i];
      if (!variantInputOk(ref, a, fr.l1, fr.l2, read2 != nullptr) || !ops || !variantOpsOk(a.o1, ops + a.ops1, a.n_ops1) ||
          (a.has_mate_pair && !variantOpsOk(a.o2, ops + a.ops2, a.n_ops2)))
        return jobFail(job, T1K_ERR_ARG, "t1k_variants_call: assignment " + std::to_string(i) + " of fragment " + std::to_string(f) +
                                             " names a window outside its allele or read, or its edit string does not spell the two windows");
    }
  }
  std::unique_ptr<t1k_variants> v(new t1k_variants);
  v->ref = &ref;
  v->vc.reset(new VariantCaller(ref, std::vector<double>(abundance, abundance + ref.seqs.size()), var_max_group));
  v->vc->compute(frags, ops);
  *out = v.release();
  return T1K_OK;
}

uint32_t t1k_variants_count(const t1k_variants *v) { return v ? (uint32_t)v->vc->variants.size() : 0; }

int t1k_variants_get(const t1k_variants *v, t1k_variant *out) {
  if (!v || !out) return T1K_ERR_ARG;
  for (size_t i = 0; i < v->vc->variants.size(); ++i) {
    const VariantRec &r = v->vc->variants[i];
    t1k_variant &o = out[i];
    o.allele_idx = r.allele; o.ref_pos = r.refPos;
    int e = 0;
    for (int p = 0; p < r.refPos; ++p) e += v->ref->exon[r.allele][p] ? 1 : 0;
    o.exon_pos = v->ref->exon[r.allele][r.refPos] ? e : -1;
    o.ref = r.ref; o.var = r.var; o.qual = r.qual; o.group = r.group; o.output_group = r.outputGroup;
    o.var_support = r.varSupport; o.all_support = r.allSupport; o.var_uniq_support = r.varUniqSupport;
  }
  return T1K_OK;
}

int t1k_variants_vcf(const t1k_variants *v, char *buf, uint64_t cap, uint64_t *needed) {
  if (!v) return T1K_ERR_ARG;
  const std::string s = v->vc->vcfText();
  if (needed) *needed = s.size();
  if (buf && cap > s.size()) { memcpy(buf, s.data(), s.size()); buf[s.size()] = 0; }
  else if (buf) return T1K_ERR_ARG;
  return T1K_OK;
}

int t1k_variants_adjust(const t1k_variants *v, const t1k_frag_assignment *asg, uint32_t n, const int8_t *ops, const char *read1, uint32_t len1, const char *read2, uint32_t len2,
                        uint8_t *keep) {
  if (!v || (n && (!asg || !keep || !ops))) return T1K_ERR_ARG;
  for (uint32_t i = 0; i < n; ++i)
    if (!variantInputOk(*v->ref, asg[i], len1, len2, read2 != nullptr) || !variantOpsOk(asg[i].o1, ops + asg[i].ops1, asg[i].n_ops1) ||
        (asg[i].has_mate_pair && !variantOpsOk(asg[i].o2, ops + asg[i].ops2, asg[i].n_ops2)))
      return T1K_ERR_ARG;
  VariantCaller::Fragment f;
  f.asg = asg; f.n = n; f.r1 = read1; f.l1 = len1; f.r2 = read2; f.l2 = len2;
  v->vc->adjust(f, ops, keep);
  return T1K_OK;
}

void t1k_variants_destroy(t1k_variants *v) { delete v; }

int t1k_job_genotype_text(t1k_job *job, char *buf, uint64_t cap, uint64_t *needed) {
  if (!job || !job->ran) return jobFail(job, T1K_ERR_STATE, "the job has not run");
  std::string s;
  for (size_t g = 0; g < job->ref.geneName.size(); ++g) s += job->gt.geneLine((int)g);
  if (needed) *needed = s.size();
  if (buf && cap > s.size()) { memcpy(buf, s.data(), s.size()); buf[s.size()] = 0; }
  else if (buf) return T1K_ERR_ARG;
  return T1K_OK;
}

int t1k_job_counts(t1k_job *job, uint64_t *fragments, uint64_t *assignedFragments, uint64_t *groups, uint64_t *ecs, int32_t *emIterations) {
  if (!job) return T1K_ERR_ARG;
  if (fragments) {
    if (job->in && job->in->streaming) { job->in->streamWait(~(size_t)0); *fragments = job->in->streamAvail(); }
    else *fragments = job->in ? job->in->nAll() : 0;
  }
  if (assignedFragments) *assignedFragments = job->gt.assignedFragments;
  if (groups) *groups = job->gt.nGroups();
  if (ecs) *ecs = job->gt.ecAlleles.size();
  if (emIterations) *emIterations = job->gt.emIterations;
  return T1K_OK;
}

int t1k_job_stats(t1k_job *job, t1k_stats *out) {
  if (!job || !out) return T1K_ERR_ARG;
  *out = job->stats;
  return T1K_OK;
}

static bool writeText(const std::string &path, const std::string &text, std::string &err) {
  FILE *fp = fopen(path.c_str(), "w");
  if (!fp) { err = "cannot write " + path; return false; }
  fwrite(text.data(), 1, text.size(), fp);
  fclose(fp);
  return true;
}

int t1k_job_write_outputs(t1k_job *job, const char *prefix) {
  if (!job || !prefix || !job->ran || !job->in) return T1K_ERR_STATE;
  const double t0 = nowMs();
  const std::string pfx = prefix;
  if (job->rank == 0) {
    std::string s;
    for (size_t g = 0; g < job->ref.geneName.size(); ++g) s += job->gt.geneLine((int)g);
    if (!writeText(pfx + "_genotype.tsv", s, job->err)) return T1K_ERR_IO;
    if (!writeText(pfx + "_allele.tsv", job->gt.alleleLines(), job->err)) return T1K_ERR_IO;
    if (job->prm.output_read_assignment && !writeText(pfx + "_assign.tsv", job->assignText, job->err)) return T1K_ERR_IO;
  }
  if (job->bgStarted && job->outPrefix == pfx) {  // already under way since the end of the device loop
    if (job->bgWriter.joinable()) job->bgWriter.join();
    job->bgStarted = false;
    if (!job->bgOk) return T1K_ERR_IO;
  } else {
    if (job->bgWriter.joinable()) job->bgWriter.join();
    if (writesAligned(job)) {
      std::vector<AlignedPlan> plans;
      if (!planAlignedFiles(job, pfx, plans) || !writePlannedFiles(job, plans)) return T1K_ERR_IO;
    }
  }
  job->msWrite = nowMs() - t0;
  job->stats.ms_write = job->msWrite;
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k job] outputs written in %.1f ms\n", job->msWrite);
  return T1K_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// the executable's entry point
// ------------------------------------------------------------------------------------------------------------------
static void logLine(const char *fmt, ...) {  // same shape as the reference's PrintLog (Genotyper.cpp:113-124): users grep these lines
  char msg[4096];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  time_t now = time(NULL);
  char stamp[128];
  strftime(stamp, sizeof(stamp), "%c", localtime(&now));
  fprintf(stderr, "[%s] %s\n", stamp, msg);
}

static const char *kUsage =
    "./genotyper [OPTIONS]:   (MI355X build of the T1K genotyper stage; same options as the reference)\n"
    "Required:\n"
    "\t-f STRING: fasta file with the allele reference sequences\n"
    "\t-u STRING: single-end read file, or\n"
    "\t-1 STRING -2 STRING: paired-end read files\n"
    "Optional:\n"
    "\t-a STRING: abundance file (skips the EM)\n"
    "\t-t INT: host threads (default: 1)\n"
    "\t-o STRING: output prefix (default: t1k)\n"
    "\t-n INT: maximal number of alleles per read (default: 2000)\n"
    "\t-s FLOAT: minimum alignment similarity (default: 0.8)\n"
    "\t--alleleWhitelist STRING: only consider reads aligned to the listed allele series\n"
    "\t--barcode STRING: barcode file\n"
    "\t--frac FLOAT: filter alleles below this fraction of the dominant allele (default: 0.15)\n"
    "\t--cov FLOAT: filter genes with average coverage below this value (default: 1.0)\n"
    "\t--crossGeneRate FLOAT: contribution of other genes' expression (default: 0.04)\n"
    "\t--relaxIntronAlign: allow one more mismatch in intronic alignment\n"
    "\t--alleleDigitUnits INT: number of name units in the genotyping result (default: automatic)\n"
    "\t--alleleDelimiter CHR: delimiter of the name units (default: automatic)\n"
    "\t--outputReadAssignment: write prefix_assign.tsv\n"
    "\t--squaremMinAlpha FLOAT: lower bound (negative) of the SQUAREM step length\n"
    "\t--device INT: GPU ordinal (default: $T1K_DEVICE or 0)\n"
    "\t--gpus INT: shard the fragments over the first INT GPUs ($T1K_GPUS=0,1,.. names them; a GPU may be named twice)\n";

int t1k_genotyper_main(int argc, char **argv) {
  if (argc <= 1) { fprintf(stderr, "%s", kUsage); return 0; }  // Genotyper.cpp:199-203
  const double tMain = nowMs();
  static struct option longOpts[] = {{"frac", required_argument, 0, 1000}, {"cov", required_argument, 0, 1001}, {"crossGeneRate", required_argument, 0, 1002},
                                     {"barcode", required_argument, 0, 1003}, {"relaxIntronAlign", no_argument, 0, 1004},
                                     {"alleleDigitUnits", required_argument, 0, 1005}, {"alleleDelimiter", required_argument, 0, 1006},
                                     {"alleleWhitelist", required_argument, 0, 1007}, {"outputReadAssignment", no_argument, 0, 1008},
                                     {"squaremMinAlpha", required_argument, 0, 1009}, {"device", required_argument, 0, 1010}, {"gpus", required_argument, 0, 1011}, {0, 0, 0, 0}};
  t1k_job_params p;
  t1k_job_params_default(&p);
  if (const char *d = getenv("T1K_DEVICE")) p.device = atoi(d);
  int nGpus = 0;
  std::string refFile, prefix = "t1k", barcode, whitelistFile, abundance;
  std::vector<const char *> f1, f2, single;  // every -u / -1 / -2 counts: the files are read back to back (ReadFiles::AddReadFile)
  optind = 1;
  int c, idx = 0;
  while ((c = getopt_long(argc, argv, "f:a:u:1:2:o:t:n:s:b:", longOpts, &idx)) != -1) {
    switch (c) {
      case 'f': refFile = optarg; break;
      case 'a': abundance = optarg; break;
      case 'u': single.push_back(optarg); break;
      case '1': f1.push_back(optarg); break;
      case '2': f2.push_back(optarg); break;
      case 'o': prefix = optarg; break;
      case 't': p.threads = atoi(optarg); break;
      case 'n': p.dev.max_assign_cnt = atoi(optarg); break;
      case 's': p.dev.ref_seq_similarity = atof(optarg); break;
      case 'b': break;
      case 1000: p.filter_frac = atof(optarg); break;
      case 1001: p.filter_cov = atof(optarg); break;
      case 1002: p.cross_gene_rate = atof(optarg); break;
      case 1003: barcode = optarg; break;
      case 1004: p.dev.relax_intron_align = 1; break;
      case 1005: p.allele_digit_units = atoi(optarg); break;
      case 1006: p.allele_delimiter = optarg[0]; break;
      case 1007: whitelistFile = optarg; break;
      case 1008: p.output_read_assignment = 1; break;
      case 1009: p.squarem_min_alpha = atof(optarg); break;
      case 1010: p.device = atoi(optarg); break;
      case 1011: nGpus = atoi(optarg); break;
      default: fprintf(stderr, "%s", kUsage); return EXIT_FAILURE;
    }
  }
  if (refFile.empty()) { fprintf(stderr, "Need to use -f to specify the reference sequences.\n"); return EXIT_FAILURE; }
  if (p.dev.max_assign_cnt == 0) p.dev.max_assign_cnt = -1;  // "-n 0" disables the cap in the reference (maxAssignCnt > 0 test)
  // GPUs of the job: --gpus N = the first N devices, T1K_GPUS = an explicit list; one rank (thread, job, context set) per entry
  std::vector<int> devices;
  if (const char *e = getenv("T1K_GPUS")) {
    for (const char *q = e; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
  } else if (nGpus > 1) {
    for (int d = 0; d < nGpus; ++d) devices.push_back(d);
  }
  if (devices.empty()) devices.push_back(p.device);
  const int R = (int)devices.size();
  std::vector<t1k_job *> jobs(R, nullptr);
  std::vector<int> rcs(R, T1K_OK);
  auto destroyAll = [&] { for (t1k_job *j : jobs) t1k_job_destroy(j); };
  const bool paired = !f2.empty();
  const std::vector<const char *> &first = !f1.empty() ? f1 : single;
  // T1K_SHARD_INPUT=1: every rank indexes only its own fragments and writes only its own part of the *_aligned*.fa files, as ranks
  // in separate processes do (bench.py under torchrun); by default the ranks of this process share one index built by all host threads
  const bool shardInput = R > 1 && getenv("T1K_SHARD_INPUT") && atoi(getenv("T1K_SHARD_INPUT")) != 0;
  // the read files are mapped and indexed while the reference is parsed and the contexts come up (the reference's main does the two
  // one after the other, Genotyper.cpp:226-232 and 365-454; neither needs the other)
  t1k_reads *opened = nullptr;
  int rcOpen = T1K_OK;
  std::thread opener;
  if (!shardInput && !first.empty() && !getenv("T1K_SERIAL_OPEN"))
    opener = std::thread([&] {
      // (one rank: an ordinary .gz input -- the barcode file with it -- is handed to the loop while it is still being inflated)
      if (R == 1) rcOpen = t1k_reads_open_stream(first.data(), (uint32_t)first.size(), paired ? f2.data() : nullptr, (uint32_t)f2.size(), barcode.empty() ? nullptr : barcode.c_str(), p.threads, &opened);
      else rcOpen = t1k_reads_open(first.data(), (uint32_t)first.size(), paired ? f2.data() : nullptr, (uint32_t)f2.size(), barcode.empty() ? nullptr : barcode.c_str(), p.threads, &opened);
    });
  {
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r)
      th.emplace_back([&, r] { t1k_job_params q = p; q.device = devices[r]; rcs[r] = t1k_job_create(&q, refFile.c_str(), &jobs[r]); });
    for (auto &t : th) t.join();
  }
  const bool openedBeside = opener.joinable();
  if (openedBeside) opener.join();
  for (int r = 0; r < R; ++r)
    if (rcs[r] != T1K_OK) {
      fprintf(stderr, "genotyper: %s\n", jobs[r] ? t1k_job_last_error(jobs[r]) : "initialisation failed");
      if (jobs[r] && jobs[r]->ref.al.empty()) fprintf(stderr, "Need to use -f to specify the reference sequences.\n");
      destroyAll();
      t1k_reads_close(opened);
      return EXIT_FAILURE;
    }
  t1k_job *job = jobs[0];
  if (!whitelistFile.empty()) {  // Genotyper::SetAlleleWhitelist (Genotyper.hpp:684-705): whole major-allele series
    FILE *fp = fopen(whitelistFile.c_str(), "r");
    if (!fp) { fprintf(stderr, "genotyper: cannot open %s\n", whitelistFile.c_str()); destroyAll(); t1k_reads_close(opened); return EXIT_FAILURE; }
    std::set<int> majors;
    std::map<std::string, int> majorId;
    for (size_t i = 0; i < job->ref.majorName.size(); ++i) majorId[job->ref.majorName[i]] = (int)i;
    char name[512];
    while (fscanf(fp, "%511s", name) == 1) {
      std::string g, m;
      job->ref.splitName(name, g, m, 0);
      auto it = majorId.find(m);
      if (it != majorId.end()) majors.insert(it->second);
    }
    fclose(fp);
    for (t1k_job *j : jobs) {
      j->whitelist.assign(j->ref.al.size(), 0);
      for (size_t a = 0; a < j->ref.al.size(); ++a) j->whitelist[a] = majors.count(j->ref.al[a].major) ? 1 : 0;
    }
  }
  for (t1k_job *j : jobs) j->abundanceFile = abundance;
  if (first.empty()) { fprintf(stderr, "genotyper: no read file given (-u, or -1 and -2)\n"); destroyAll(); t1k_reads_close(opened); return EXIT_FAILURE; }
  auto loadInto = [&](t1k_job *j) {
    return t1k_job_load_reads_multi(j, first.data(), (uint32_t)first.size(), paired ? f2.data() : nullptr, (uint32_t)f2.size(), barcode.empty() ? nullptr : barcode.c_str());
  };
  int rc = T1K_OK;
  bool foundLater = false;
  if (!shardInput) {
    if (openedBeside) { rc = t1k_job_attach_reads(job, opened); opened = nullptr; if (rc == T1K_OK) rc = rcOpen; }  // (a failed open: the handle carries the message into the job)
    else rc = loadInto(job);
    if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); destroyAll(); return EXIT_FAILURE; }
    job->in->dropInflatedText = R == 1 && !getenv("T1K_KEEP_TEXT");  // this process runs the job once: the text of written fragments is not needed again
    foundLater = job->in->streaming;  // (a streamed input: counted when the stream has ended, i.e. behind the loop)
    if (!foundLater) logLine("Found %d read fragments. Start read assignment.", (int)job->in->nAll());
    t1k_job_set_output_prefix(job, prefix.c_str());  // the aligned-read files are written while the EM runs
  }
  if (R == 1) rc = t1k_job_run(job);
  else {
    // one thread per rank: the ranks meet in the collectives of t1k_job_run (RCCL when every rank has its own GPU)
    t1k_comm_group *group = t1k_comm_group_create(R);
    std::vector<t1k_comm *> comms(R, nullptr);
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r)
      th.emplace_back([&, r] {
        int x = (r && !shardInput) ? t1k_job_share_reads(jobs[r], job) : T1K_OK;
        const int y = t1k_comm_init(t1k_job_ctx(jobs[r]), R, r, nullptr, group, -1, &comms[r]);  // collective: every rank calls it
        if (x == T1K_OK && y != T1K_OK) { jobs[r]->err = comms[r] ? t1k_comm_last_error(comms[r]) : "cannot create the communicator"; x = y; }
        if (x == T1K_OK) x = t1k_job_set_shard(jobs[r], r, R, comms[r]);
        if (x == T1K_OK && shardInput) {
          x = loadInto(jobs[r]);  // collective
          if (x == T1K_OK) {
            if (r == 0) logLine("Found %d read fragments. Start read assignment.", (int)jobs[r]->in->nAll());
            t1k_job_set_output_prefix(jobs[r], prefix.c_str());
          }
        }
        rcs[r] = x == T1K_OK ? t1k_job_run(jobs[r]) : x;
        if (rcs[r] == T1K_OK && shardInput && r) rcs[r] = t1k_job_write_outputs(jobs[r], prefix.c_str());  // its part of the read files (rank 0: below)
        // a rank that gives up must not leave the others waiting at the next exchange: they are released with an error of their own
        if (rcs[r] != T1K_OK && comms[r]) (void)t1k_comm_abort(comms[r]);
      });
    for (auto &t : th) t.join();
    for (int pass = 0; pass < 2 && rc == T1K_OK; ++pass)  // report the rank that failed, not the ones it released (T1K_ERR_STATE)
      for (int r = 0; r < R && rc == T1K_OK; ++r)
        if (rcs[r] != T1K_OK && (pass == 1 || rcs[r] != T1K_ERR_STATE)) { rc = rcs[r]; if (r) job->err = t1k_job_last_error(jobs[r]); }
    for (t1k_comm *c : comms) t1k_comm_destroy(c);
    t1k_comm_group_destroy(group);
  }
  if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); destroyAll(); return EXIT_FAILURE; }
  if (foundLater) logLine("Found %d read fragments. Start read assignment.", (int)job->in->nAll());
  logLine("Finish read end assignments.");
  const double groups = (double)job->gt.nGroups();
  logLine("Finish read fragment assignments. %d read fragments can be assigned (average %.2lf alleles/read).", (int)job->gt.assignedFragments,
          job->gt.sumAssign / groups);
  if (abundance.empty()) logLine("Finish allele quantification in %d EM iterations.", job->gt.emIterations);
  rc = t1k_job_write_outputs(job, prefix.c_str());
  if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); destroyAll(); return EXIT_FAILURE; }
  logLine("Genotyping finishes.");
  const double tOut = nowMs();
  destroyAll();
  if (getenv("T1K_DEBUG_PHASES")) {  // (what a stopwatch around the process sees beyond this: loading the executable and the HIP runtime before main, the exit behind it)
    fprintf(stderr, "[t1k job] main: %.1f ms from its first line to the outputs, %.1f ms to release the job\n", tOut - tMain, nowMs() - tOut);
    // what the process still maps when it leaves (the kernel takes the address space apart before the parent sees the exit)
    if (FILE *fp = fopen("/proc/self/smaps_rollup", "r")) {
      char line[256];
      std::string all;
      while (fgets(line, sizeof line, fp))
        if (!strncmp(line, "Rss:", 4) || !strncmp(line, "Anonymous:", 10) || !strncmp(line, "Shared_Clean:", 13) || !strncmp(line, "Shared_Dirty:", 13) || !strncmp(line, "Private_Clean:", 14) ||
            !strncmp(line, "Private_Dirty:", 14) || !strncmp(line, "AnonHugePages:", 14) || !strncmp(line, "Locked:", 7)) {
          std::string l(line);
          while (!l.empty() && (l.back() == '\n' || l.back() == ' ')) l.pop_back();
          size_t a = l.find(':');
          size_t b = l.find_first_not_of(' ', a + 1);
          all += l.substr(0, a + 1) + " " + (b == std::string::npos ? "" : l.substr(b)) + "; ";
        }
      fclose(fp);
      fprintf(stderr, "[t1k job] address space at the end of main: %s\n", all.c_str());
    }
    if (getenv("T1K_DEBUG_MAPS"))  // the largest resident mappings (what the exit has to take apart page by page)
      if (FILE *fp = fopen("/proc/self/smaps", "r")) {
        struct Reg { std::string head; unsigned long rss = 0, anon = 0; };
        std::vector<Reg> regs;
        char line[512];
        while (fgets(line, sizeof line, fp)) {
          unsigned long a, b;
          if (sscanf(line, "%lx-%lx ", &a, &b) == 2 && strchr(line, '-') && (strstr(line, " r") || strstr(line, " -"))) { Reg r; r.head = line; while (!r.head.empty() && r.head.back() == '\n') r.head.pop_back(); regs.push_back(r); }
          else if (!regs.empty() && !strncmp(line, "Rss:", 4)) regs.back().rss = strtoul(line + 4, nullptr, 10);
          else if (!regs.empty() && !strncmp(line, "Anonymous:", 10)) regs.back().anon = strtoul(line + 10, nullptr, 10);
        }
        fclose(fp);
        std::sort(regs.begin(), regs.end(), [](const Reg &x, const Reg &y) { return x.rss > y.rss; });
        for (size_t i = 0; i < regs.size() && i < 24; ++i) fprintf(stderr, "[t1k job]   rss %8lu kB (anonymous %8lu kB)  %s\n", regs[i].rss, regs[i].anon, regs[i].head.c_str());
      }
    if (FILE *fp = fopen("/proc/self/status", "r")) {
      char line[256];
      while (fgets(line, sizeof line, fp))
        if (!strncmp(line, "Threads:", 8) || !strncmp(line, "VmPeak:", 7) || !strncmp(line, "VmHWM:", 6) || !strncmp(line, "VmPTE:", 6)) fprintf(stderr, "[t1k job]   %s", line);
      fclose(fp);
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// analyzer: Analyzer.cpp:236-733 as run-t1k:438-449 starts it after the genotyper -- the aligned reads are
// assigned again, to the alleles named in <prefix>_allele.tsv only (Genotyper::InitRefSet with selectedAlleles, Genotyper.hpp:732-757;
// AssignRead with weight 0: no coverage is kept, Analyzer.cpp:139, 472), mates are paired, and BarcodeSummary (BarcodeSummary.hpp:24-80)
// turns every assigned fragment's allele list into 1/n fractional and unique counts per barcode: <prefix>_barcode_expr.tsv.
// Novel-variant calling (VariantCaller.hpp) follows as in the reference unless --varMaxGroup 0 is given (VariantCaller.hpp:980-981: no variant
// is called, <prefix>_allele.vcf is empty and AdjustFragmentAssignment hands every fragment's raw assignments back): analyzerCallVariants below.
// ------------------------------------------------------------------------------------------------------------------
// What the reference's analyzer does between its fragment assignment and its VariantCaller (Analyzer.cpp:560-684), for a job that has run its
// windows in analyzer mode (raw fragment rows resident in job->rows):
//   (1) Genotyper::SetReadAssignments + CoalesceReadAssignments + FinalizeReadAssignments + QuantifyAlleleEquivalentClass (570-609): the -n and
//       separator drops applied to the raw rows on the host, the rows coalesced (Genotyper::coalesce), the EM on the GPU (t1k_em_*) --
//       VariantCaller::SetSeqAbundance reads the alleles' abundances;
//   (2) the overlaps behind every kept assignment: the assigned fragments' distinct read-ends go through t1k_assign_batch once more on a
//       context of their own, in pieces of 32768, their final overlap lists come back (t1k_overlaps_download) and fragmentDetails takes
//       ReadAssignmentToFragmentAssignment's per-allele choice again (host/variants.cpp) -- the device rows keep the fragment's window only;
//   (3) SeqSet::AddFragmentAlignmentInfo (611-668): one global alignment per distinct (read-end, overlap) on the GPU (t1k_align_batch);
//   (4) VariantCaller::ComputeVariant on the host (host/variants.cpp).
struct AnalyzerVariants {
  std::vector<uint64_t> asgPtr;             // fragment -> its assignments
  std::vector<t1k_frag_assignment> asg;
  std::vector<int8_t> ops;
  std::unique_ptr<VariantCaller> vc;
  int emIterations = 0;
};

static int analyzerCallVariants(t1k_job *job, int varMaxGroup, AnalyzerVariants &V) {
  const double tv0 = nowMs();
  double msAssign = 0, msDetails = 0, msAlign = 0;
  uint64_t nEnds = 0, nJobs = 0;
  const ReadInput &in = *job->in;
  const RefSet &R = job->ref;
  const uint32_t F = (uint32_t)in.nFrag();
  const bool paired = in.paired;
  int rc;
  // every fragment's raw row (the reference's list order)
  std::vector<uint32_t> cnt(F);
  std::vector<uint64_t> rowAt(F + 1, 0);
  std::vector<t1k_row_entry> rows;
  {
    const uint32_t step = 1u << 18;
    std::vector<t1k_row_entry> part;
    for (uint32_t f0 = 0; f0 < F; f0 += step) {
      const uint32_t n = std::min(step, F - f0);
      uint64_t total = 0;
      if ((rc = t1k_rowset_rows_download(job->rows, f0, n, cnt.data() + f0, nullptr, 0, &total)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
      part.resize(total);
      if (total && (rc = t1k_rowset_rows_download(job->rows, f0, n, cnt.data() + f0, part.data(), total, &total)) != T1K_OK) return jobFail(job, rc, t1k_rowset_last_error(job->rows));
      rows.insert(rows.end(), part.begin(), part.end());
    }
    for (uint32_t f = 0; f < F; ++f) rowAt[f + 1] = rowAt[f] + cnt[f];
    if (rowAt[F] != rows.size()) return jobFail(job, T1K_ERR_INTERNAL, "analyzer: the row counts do not add up to the rows downloaded");
  }
  // (1) the analyzer's EM
  {
    Genotyper &gt = job->gt;
    const int maxAssign = job->prm.dev.max_assign_cnt;
    std::vector<t1k_row_entry> tmp;
    for (uint32_t f = 0; f < F; ++f) {
      const uint32_t k = cnt[f];
      if (!k || (maxAssign > 0 && (int)k > maxAssign)) continue;  // Genotyper.hpp:783-784
      bool sep = false;                                            // IsFragmentSpanSeparator (796-800): an N of the allele inside the fragment's window
      for (uint32_t j = 0; j < k && !sep; ++j) {
        const t1k_row_entry &e = rows[rowAt[f] + j];
        const std::string &sq = R.seqs[e.allele_idx];
        for (int p = std::max(e.start, 0); p <= e.end && p < (int)sq.size(); ++p)
          if (sq[p] == 'N') { sep = true; break; }
      }
      if (sep) continue;
      tmp.assign(rows.begin() + rowAt[f], rows.begin() + rowAt[f] + k);
      gt.coalesce(tmp.data(), k, f);
    }
    gt.finalize(std::vector<int32_t>(R.al.size(), 0));  // (missingCoverage is not read before selection, which the analyzer does not run)
    if (gt.nGroups() && (V.emIterations = gt.quantify(job->ctx, nullptr, job->err)) < 0) return T1K_ERR_DEVICE;
  }
  std::vector<double> abundance(R.al.size());
  for (size_t a = 0; a < R.al.size(); ++a) abundance[a] = R.al[a].abundance;
  const double tv1 = nowMs();
  // (2) + (3)
  t1k_ctx *vctx = nullptr;
  if ((rc = t1k_ctx_create(job->prm.device, &job->prm.dev, &vctx)) != T1K_OK) { if (vctx) t1k_ctx_destroy(vctx); return jobFail(job, rc, "analyzer: cannot create the context of the variant pass"); }
  struct CtxGuard { t1k_ctx *c; ~CtxGuard() { t1k_ctx_destroy(c); } } guard{vctx};
  if ((rc = t1k_ref_share(vctx, job->ctx)) != T1K_OK) return jobFail(job, rc, t1k_last_error(vctx));
  std::string refText;
  std::vector<uint64_t> refOff(R.seqs.size() + 1, 0);
  for (size_t a = 0; a < R.seqs.size(); ++a) refOff[a + 1] = refOff[a] + R.seqs[a].size();
  if (refOff.back() >= (1ull << 32)) return jobFail(job, T1K_ERR_CAPACITY, "analyzer: the selected alleles hold more than 4 G bases");
  refText.reserve(refOff.back());
  for (const std::string &sq : R.seqs) refText += sq;
  V.asgPtr.assign(F + 1, 0);
  V.asg.resize(rows.size());
  for (uint32_t f = 0; f < F; ++f) V.asgPtr[f + 1] = V.asgPtr[f] + (job->fragAssigned[f] ? cnt[f] : 0);
  V.asg.resize(V.asgPtr[F]);
  // read-ends per piece: the range size of the job's own loop (T1K_ANALYZER_PIECE: tests run several pieces on small inputs)
  const uint32_t pieceEnds = getenv("T1K_ANALYZER_PIECE") ? (uint32_t)std::max(2, atoi(getenv("T1K_ANALYZER_PIECE"))) : 32768u;
  auto readOf = [&](uint32_t f, int m) { const uint32_t r = in.frag[f]; return std::pair<const char *, uint32_t>(in.side[m].seqP[r], in.side[m].seqL[r]); };
  uint32_t f0 = 0;
  while (f0 < F) {
    // a piece: fragments [f0, f1) whose distinct read-ends fit one upload
    std::unordered_map<std::string, uint32_t> idOf;
    std::vector<std::pair<const char *, uint32_t>> ends;
    std::vector<uint32_t> endOf;  // (fragment - f0) * 2 + mate -> distinct read-end of the piece
    uint32_t f1 = f0;
    for (; f1 < F && ends.size() + 2 <= pieceEnds; ++f1) {
      endOf.push_back(~0u); endOf.push_back(~0u);
      if (!job->fragAssigned[f1] || !cnt[f1]) continue;
      for (int m = 0; m < (paired ? 2 : 1); ++m) {
        auto rd = readOf(f1, m);
        auto it = idOf.emplace(std::string(rd.first, rd.second), (uint32_t)ends.size());
        if (it.second) ends.push_back(rd);
        endOf[(size_t)(f1 - f0) * 2 + m] = it.first->second;
      }
    }
    const uint32_t E = (uint32_t)ends.size();
    if (E) {
      std::string text;
      std::vector<uint64_t> off(E + 1, 0);
      for (uint32_t e = 0; e < E; ++e) off[e + 1] = off[e] + ends[e].second;
      text.reserve(off[E]);
      for (uint32_t e = 0; e < E; ++e) text.append(ends[e].first, ends[e].second);
      const double ta = nowMs();
      nEnds += E;
      if ((rc = t1k_reads_upload(vctx, text.data(), off.data(), nullptr, E)) != T1K_OK) return jobFail(job, rc, t1k_last_error(vctx));
      if ((rc = t1k_assign_batch(vctx)) != T1K_OK) return jobFail(job, rc, t1k_last_error(vctx));
      std::vector<uint32_t> lc(E);
      uint64_t total = 0;
      if ((rc = t1k_overlaps_download(vctx, lc.data(), nullptr, 0, &total)) != T1K_OK) return jobFail(job, rc, t1k_last_error(vctx));
      std::vector<t1k_overlap> lists(total);
      if (total && (rc = t1k_overlaps_download(vctx, lc.data(), lists.data(), total, &total)) != T1K_OK) return jobFail(job, rc, t1k_last_error(vctx));
      std::vector<uint64_t> listAt(E + 1, 0);
      for (uint32_t e = 0; e < E; ++e) listAt[e + 1] = listAt[e] + lc[e];
      const double tb = nowMs();
      msAssign += tb - ta;
      // the overlaps behind every kept assignment
      std::vector<int32_t> alleles;
      for (uint32_t f = f0; f < f1; ++f) {
        if (!job->fragAssigned[f] || !cnt[f]) continue;
        const uint32_t k = cnt[f];
        alleles.resize(k);
        for (uint32_t j = 0; j < k; ++j) alleles[j] = rows[rowAt[f] + j].allele_idx;
        const uint32_t e1 = endOf[(size_t)(f - f0) * 2], e2 = paired ? endOf[(size_t)(f - f0) * 2 + 1] : 0;
        if (!fragmentDetails(lists.data() + listAt[e1], lc[e1], paired ? lists.data() + listAt[e2] : nullptr, paired ? lc[e2] : 0, paired, alleles.data(), k, V.asg.data() + V.asgPtr[f]))
          return jobFail(job, T1K_ERR_INTERNAL, "analyzer: fragment " + std::to_string(f) + " is assigned to an allele its read-ends' overlap lists do not hold");
      }
      // one alignment per distinct (read-end, overlap) an assignment names: the overlap is found again in its list by its address
      std::vector<int64_t> jobOf(total, -1);
      struct Job { uint32_t end, idx; };
      std::vector<Job> jobs;
      auto jobFor = [&](uint32_t e, const t1k_overlap &o) -> int64_t {
        for (uint32_t i = 0; i < lc[e]; ++i) {
          const t1k_overlap &c = lists[listAt[e] + i];
          if (c.seq_idx == o.seq_idx && c.read_start == o.read_start && c.read_end == o.read_end && c.seq_start == o.seq_start && c.seq_end == o.seq_end && c.strand == o.strand) {
            int64_t &slot = jobOf[listAt[e] + i];
            if (slot < 0) { slot = (int64_t)jobs.size(); jobs.push_back({e, i}); }
            return slot;
          }
        }
        return -1;
      };
      std::vector<int64_t> jobOfAsg[2];
      jobOfAsg[0].assign(V.asgPtr[f1] - V.asgPtr[f0], -1);
      jobOfAsg[1].assign(V.asgPtr[f1] - V.asgPtr[f0], -1);
      for (uint32_t f = f0; f < f1; ++f)
        for (uint64_t q = V.asgPtr[f]; q < V.asgPtr[f + 1]; ++q) {
          const t1k_frag_assignment &a = V.asg[q];
          const uint32_t eA = endOf[(size_t)(f - f0) * 2 + ((a.o1_from_r2 && !a.has_mate_pair) ? 1 : 0)];
          if ((jobOfAsg[0][q - V.asgPtr[f0]] = jobFor(eA, a.o1)) < 0) return jobFail(job, T1K_ERR_INTERNAL, "analyzer: an assignment's overlap is not in its read-end's list");
          if (a.has_mate_pair && (jobOfAsg[1][q - V.asgPtr[f0]] = jobFor(endOf[(size_t)(f - f0) * 2 + 1], a.o2)) < 0)
            return jobFail(job, T1K_ERR_INTERNAL, "analyzer: an assignment's overlap is not in its read-end's list");
        }
      // patterns: the read-ends as they are and, where an overlap is on the other strand, reverse-complemented (SeqSet.hpp:2663-2668)
      std::vector<uint64_t> rcAt(E, ~0ull);
      std::string pat = text;
      for (const Job &jb : jobs)
        if (lists[listAt[jb.end] + jb.idx].strand == -1 && rcAt[jb.end] == ~0ull) {
          rcAt[jb.end] = pat.size();
          const char *p = ends[jb.end].first;
          const uint32_t n = ends[jb.end].second;
          for (uint32_t i = 0; i < n; ++i) { const char c = p[n - 1 - i]; pat += c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
        }
      if (pat.size() >= (1ull << 32)) return jobFail(job, T1K_ERR_CAPACITY, "analyzer: a piece's read text exceeds 4 GB");
      std::vector<uint64_t> opsAtOfJob(jobs.size());
      std::vector<uint32_t> nOpsOfJob(jobs.size());
      const double tc = nowMs();
      msDetails += tc - tb;
      nJobs += jobs.size();
      const size_t callJobs = 1u << 18;
      for (size_t j0 = 0; j0 < jobs.size(); j0 += callJobs) {
        const uint32_t n = (uint32_t)std::min(callJobs, jobs.size() - j0);
        std::vector<uint32_t> tOff(n), tLen(n), pOff(n), pLen(n), oOff(n), nOps(n);
        uint64_t room = 0;
        for (uint32_t i = 0; i < n; ++i) {
          const Job &jb = jobs[j0 + i];
          const t1k_overlap &o = lists[listAt[jb.end] + jb.idx];
          tOff[i] = (uint32_t)(refOff[o.seq_idx] + (uint64_t)o.seq_start);
          tLen[i] = (uint32_t)(o.seq_end - o.seq_start + 1);
          pOff[i] = (uint32_t)((o.strand == -1 ? rcAt[jb.end] : off[jb.end]) + (uint64_t)o.read_start);
          pLen[i] = (uint32_t)(o.read_end - o.read_start + 1);
          oOff[i] = (uint32_t)room;
          room += (uint64_t)tLen[i] + pLen[i] + 2;
        }
        if (room >= (1ull << 32)) return jobFail(job, T1K_ERR_CAPACITY, "analyzer: the edit strings of one alignment call exceed 4 GB");
        std::vector<int8_t> buf(room + 64);
        if ((rc = t1k_align_batch(vctx, refText.data(), tOff.data(), tLen.data(), pat.data(), pOff.data(), pLen.data(), n, nullptr, nullptr, nullptr, nullptr, buf.data(), oOff.data(), nOps.data())) != T1K_OK)
          return jobFail(job, rc, t1k_last_error(vctx));
        for (uint32_t i = 0; i < n; ++i) {
          opsAtOfJob[j0 + i] = V.ops.size();
          nOpsOfJob[j0 + i] = nOps[i];
          V.ops.insert(V.ops.end(), buf.begin() + oOff[i], buf.begin() + oOff[i] + nOps[i]);
        }
      }
      for (uint64_t q = V.asgPtr[f0]; q < V.asgPtr[f1]; ++q) {
        t1k_frag_assignment &a = V.asg[q];
        const int64_t j1 = jobOfAsg[0][q - V.asgPtr[f0]], j2 = jobOfAsg[1][q - V.asgPtr[f0]];
        a.ops1 = opsAtOfJob[j1]; a.n_ops1 = nOpsOfJob[j1];
        if (a.has_mate_pair) { a.ops2 = opsAtOfJob[j2]; a.n_ops2 = nOpsOfJob[j2]; }
      }
      msAlign += nowMs() - tc;
    }
    f0 = f1;
  }
  const double tv2 = nowMs();
  // (4)
  std::vector<VariantCaller::Fragment> frags(F);
  for (uint32_t f = 0; f < F; ++f) {
    VariantCaller::Fragment &fr = frags[f];
    fr.asg = V.asg.data() + V.asgPtr[f];
    fr.n = (uint32_t)(V.asgPtr[f + 1] - V.asgPtr[f]);
    auto a = readOf(f, 0);
    fr.r1 = a.first; fr.l1 = a.second;
    if (paired) { auto b = readOf(f, 1); fr.r2 = b.first; fr.l2 = b.second; }
  }
  V.vc.reset(new VariantCaller(R, abundance, varMaxGroup));
  V.vc->compute(frags, V.ops.data());
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k analyzer] variant pass: rows + EM %.1f ms; %llu distinct read-ends re-assigned in %.1f ms, overlaps chosen in %.1f ms, %llu alignments in %.1f ms; "
                    "VariantCaller %.1f ms (%zu assignments, %zu variants); %.1f ms in all\n", tv1 - tv0, (unsigned long long)nEnds, msAssign, msDetails, (unsigned long long)nJobs, msAlign,
            nowMs() - tv2, V.asg.size(), V.vc->variants.size(), nowMs() - tv0);
  return T1K_OK;
}

static const char *kAnalyzerUsage =
    "./analyzer [OPTIONS]:   (MI355X build of the T1K post-analysis stage: re-assignment, novel variants, per-barcode summary)\n"
    "Required:\n"
    "\t-f STRING: fasta file with the allele reference sequences\n"
    "\t-a STRING: selected alleles list file (prefix_allele.tsv)\n"
    "\t-u STRING: single-end read file, or\n"
    "\t-1 STRING -2 STRING: paired-end read files\n"
    "Optional:\n"
    "\t-t INT: host threads (default: 1)\n"
    "\t-o STRING: output prefix (default: t1k)\n"
    "\t-n INT: maximal number of alleles per read (default: 2000)\n"
    "\t-s FLOAT: minimum alignment similarity (default: 0.8)\n"
    "\t--barcode STRING: barcode file\n"
    "\t--relaxIntronAlign: allow one more mismatch in intronic alignment\n"
    "\t--alleleDigitUnits INT, --alleleDelimiter CHR: as in genotyper\n"
    "\t--varMaxGroup INT: the maximum variant group size to call novel variant. -1 for no limitation, 0 for no variant calling (default: 8)\n"
    "\t--device INT: GPU ordinal (default: $T1K_DEVICE or 0)\n";

int t1k_analyzer_main(int argc, char **argv) {
  if (argc <= 1) { fprintf(stderr, "%s", kAnalyzerUsage); return 0; }  // Analyzer.cpp:241-245
  static struct option longOpts[] = {{"barcode", required_argument, 0, 10000}, {"relaxIntronAlign", no_argument, 0, 10004}, {"alleleDigitUnits", required_argument, 0, 10005},
                                     {"alleleDelimiter", required_argument, 0, 10006}, {"varMaxGroup", required_argument, 0, 10007}, {"device", required_argument, 0, 10010},
                                     {0, 0, 0, 0}};
  t1k_job_params p;
  t1k_job_params_default(&p);
  if (const char *d = getenv("T1K_DEVICE")) p.device = atoi(d);
  std::string refFile, alleleFile, prefix = "t1k", barcode;
  std::vector<const char *> f1, f2, single;
  int varMaxGroup = 8;  // Analyzer.cpp:251
  optind = 1;
  int c, idx = 0;
  while ((c = getopt_long(argc, argv, "f:a:u:1:2:o:t:n:s:", longOpts, &idx)) != -1) {
    switch (c) {
      case 'f': refFile = optarg; break;
      case 'a': alleleFile = optarg; break;
      case 'u': single.push_back(optarg); break;
      case '1': f1.push_back(optarg); break;
      case '2': f2.push_back(optarg); break;
      case 'o': prefix = optarg; break;
      case 't': p.threads = atoi(optarg); break;
      case 'n': p.dev.max_assign_cnt = atoi(optarg); break;
      case 's': p.dev.ref_seq_similarity = atof(optarg); break;
      case 10000: barcode = optarg; break;
      case 10004: p.dev.relax_intron_align = 1; break;
      case 10005: p.allele_digit_units = atoi(optarg); break;
      case 10006: p.allele_delimiter = optarg[0]; break;
      case 10007: varMaxGroup = atoi(optarg); break;
      case 10010: p.device = atoi(optarg); break;
      default: fprintf(stderr, "%s", kAnalyzerUsage); return EXIT_FAILURE;
    }
  }
  if (refFile.empty()) { fprintf(stderr, "Need to use -f to specify the reference sequences.\n"); return EXIT_FAILURE; }
  if (alleleFile.empty()) { fprintf(stderr, "Need to use -a to specify selected allele ids.\n"); return EXIT_FAILURE; }
  if (p.dev.max_assign_cnt == 0) p.dev.max_assign_cnt = -1;
  std::set<std::string> selected;
  {
    FILE *fp = fopen(alleleFile.c_str(), "r");  // first word of every line (Analyzer.cpp:347-356)
    if (!fp) { fprintf(stderr, "analyzer: cannot open %s\n", alleleFile.c_str()); return EXIT_FAILURE; }
    char line[10241], name[10241];
    while (fgets(line, sizeof(line), fp))
      if (sscanf(line, "%10240s", name) == 1) selected.insert(name);
    fclose(fp);
  }
  if (selected.empty()) {
    // nothing was genotyped (run-t1k starts the analyzer all the same): the reference loads no sequence, assigns no fragment and
    // leaves an empty VCF and a per-barcode table that is only its header
    FILE *fv = fopen((prefix + "_allele.vcf").c_str(), "w");
    if (!fv) { fprintf(stderr, "analyzer: cannot write %s_allele.vcf\n", prefix.c_str()); return EXIT_FAILURE; }
    fclose(fv);
    if (!barcode.empty()) {
      FILE *fb = fopen((prefix + "_barcode_expr.tsv").c_str(), "w");
      if (!fb) { fprintf(stderr, "analyzer: cannot write %s_barcode_expr.tsv\n", prefix.c_str()); return EXIT_FAILURE; }
      fprintf(fb, "#barcode\n");
      fclose(fb);
    }
    logLine("Post analysis finishes.");
    return 0;
  }
  t1k_job *job = nullptr;
  int rc = jobCreate(&p, refFile.c_str(), &selected, &job);
  if (rc != T1K_OK) {
    fprintf(stderr, "analyzer: %s\n", job ? t1k_job_last_error(job) : "initialisation failed");
    t1k_job_destroy(job);
    return EXIT_FAILURE;
  }
  job->analyzer = true;
  const bool paired = !f2.empty();
  const std::vector<const char *> &first = !f1.empty() ? f1 : single;
  if (first.empty()) { fprintf(stderr, "analyzer: no read file given (-u, or -1 and -2)\n"); t1k_job_destroy(job); return EXIT_FAILURE; }
  rc = t1k_job_load_reads_multi(job, first.data(), (uint32_t)first.size(), paired ? f2.data() : nullptr, (uint32_t)f2.size(), barcode.empty() ? nullptr : barcode.c_str());
  if (rc != T1K_OK) { fprintf(stderr, "analyzer: %s\n", t1k_job_last_error(job)); t1k_job_destroy(job); return EXIT_FAILURE; }
  const ReadInput &in = *job->in;
  const uint32_t F = (uint32_t)in.nFrag();
  logLine("Found %d read fragments. Start read assignment.", (int)F);
  rc = t1k_job_run_local(job);
  if (rc != T1K_OK) { fprintf(stderr, "analyzer: %s\n", t1k_job_last_error(job)); t1k_job_destroy(job); return EXIT_FAILURE; }
  logLine("Finish read end assignments.");
  uint64_t nAssigned = 0;
  for (uint32_t f = 0; f < F; ++f) nAssigned += job->fragAssigned[f] ? 1 : 0;
  logLine("Finish read fragment assignments. %d read fragments can be assigned.", (int)nAssigned);
  AnalyzerVariants V;
  if (varMaxGroup != 0) {  // (0: VariantCaller::ComputeVariant returns before it looks at a read, VariantCaller.hpp:980-981)
    rc = analyzerCallVariants(job, varMaxGroup, V);
    if (rc != T1K_OK) { fprintf(stderr, "analyzer: %s\n", t1k_job_last_error(job)); t1k_job_destroy(job); return EXIT_FAILURE; }
    logLine("Finish allele quantification in %d EM iterations.", V.emIterations);
  }
  {
    FILE *fp = fopen((prefix + "_allele.vcf").c_str(), "w");  // VariantCaller::OutputAlleleVCF (1202-1227)
    if (!fp) { fprintf(stderr, "analyzer: cannot write %s_allele.vcf\n", prefix.c_str()); t1k_job_destroy(job); return EXIT_FAILURE; }
    if (V.vc) { const std::string text = V.vc->vcfText(); fwrite(text.data(), 1, text.size(), fp); }
    fclose(fp);
  }
  if (in.hasBarcode) {
    // barcode ids in order of first appearance over ALL loaded fragments (Analyzer.cpp:380-392), counts in fragment order
    std::unordered_map<std::string, int> idOf;
    std::vector<std::string> names;
    std::vector<int> bcOf(F);
    for (uint32_t f = 0; f < F; ++f) {
      const uint32_t r = in.frag[f];
      std::string s(in.bc.seqP[r], in.bc.seqL[r]);
      auto it = idOf.find(s);
      if (it == idOf.end()) { it = idOf.emplace(s, (int)names.size()).first; names.push_back(s); }
      bcOf[f] = it->second;
    }
    const size_t A = job->ref.al.size();
    std::map<int, std::pair<std::vector<double>, std::vector<int>>> table;  // barcode -> (fractional counts, unique counts)
    const uint32_t step = 1u << 18;
    std::vector<uint32_t> cnt;
    std::vector<t1k_row_entry> rows;
    std::vector<uint8_t> keepFlag;
    for (uint32_t f0 = 0; f0 < F; f0 += step) {
      const uint32_t n = std::min(step, F - f0);
      cnt.resize(n);
      uint64_t total = 0;
      rc = t1k_rowset_rows_download(job->rows, f0, n, cnt.data(), nullptr, 0, &total);
      rows.resize(total);
      if (rc == T1K_OK && total) rc = t1k_rowset_rows_download(job->rows, f0, n, cnt.data(), rows.data(), total, &total);
      if (rc != T1K_OK) { fprintf(stderr, "analyzer: %s\n", t1k_rowset_last_error(job->rows)); t1k_job_destroy(job); return EXIT_FAILURE; }
      uint64_t q = 0;
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t k = cnt[i];
        if (!job->fragAssigned[f0 + i]) { q += k; continue; }
        auto &slot = table[bcOf[f0 + i]];  // BarcodeSummary::AddFragment (BarcodeSummary.hpp:24-57)
        if (slot.first.empty()) { slot.first.assign(A, 0.0); slot.second.assign(A, 0); }
        if (V.vc) {  // the assignments VariantCaller::AdjustFragmentAssignment keeps (1229-1311)
          const uint32_t f = f0 + i;
          VariantCaller::Fragment fr;
          fr.asg = V.asg.data() + V.asgPtr[f]; fr.n = (uint32_t)(V.asgPtr[f + 1] - V.asgPtr[f]);
          const uint32_t r = in.frag[f];
          fr.r1 = in.side[0].seqP[r]; fr.l1 = in.side[0].seqL[r];
          if (in.paired) { fr.r2 = in.side[1].seqP[r]; fr.l2 = in.side[1].seqL[r]; }
          keepFlag.assign(k, 0);
          if (fr.n == k) V.vc->adjust(fr, V.ops.data(), keepFlag.data());
          uint32_t kept = 0;
          for (uint32_t j = 0; j < k; ++j) kept += keepFlag[j];
          for (uint32_t j = 0; j < k; ++j, ++q) {
            if (!keepFlag[j]) continue;
            slot.first[rows[q].allele_idx] += 1.0 / kept;
            if (kept == 1) ++slot.second[rows[q].allele_idx];
          }
          continue;
        }
        for (uint32_t j = 0; j < k; ++j, ++q) {
          slot.first[rows[q].allele_idx] += 1.0 / k;
          if (k == 1) ++slot.second[rows[q].allele_idx];
        }
      }
    }
    FILE *fp = fopen((prefix + "_barcode_expr.tsv").c_str(), "w");  // BarcodeSummary::Output (59-80)
    if (!fp) { fprintf(stderr, "analyzer: cannot write %s_barcode_expr.tsv\n", prefix.c_str()); t1k_job_destroy(job); return EXIT_FAILURE; }
    fprintf(fp, "#barcode");
    for (size_t a = 0; a < A; ++a) fprintf(fp, "\t%s", job->ref.al[a].name.c_str());
    for (size_t a = 0; a < A; ++a) fprintf(fp, "\t%s_uniq", job->ref.al[a].name.c_str());
    fprintf(fp, "\n");
    for (auto &kv : table) {
      fprintf(fp, "%s", names[kv.first].c_str());
      for (size_t a = 0; a < A; ++a) fprintf(fp, "\t%lf", kv.second.first[a]);
      for (size_t a = 0; a < A; ++a) fprintf(fp, "\t%d", kv.second.second[a]);
      fprintf(fp, "\n");
    }
    fclose(fp);
  }
  logLine("Post analysis finishes.");
  t1k_job_destroy(job);
  return 0;
}

}  // extern "C"
