// t1k_amd/csrc/host/job.cpp -- the genotyper stage as a job, first half: the t1k_job_* / t1k_reads_* C ABI for creating a job and handing it its
// reads, and t1k_job_run_local -- the window loop, the device half of the stage (Genotyper.cpp:443-650 up to the coalesced read groups).
// The other stages of the job layer: job_finish.cpp, job_output.cpp, genotyper_main.cpp, analyzer.cpp (job_internal.h).
#include "job_internal.h"

extern "C" {


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

int t1k_job_create(const t1k_job_params *p, const char *refFasta, t1k_job **out) { return jobCreate(p, refFasta, nullptr, out); }

}  // extern "C"
namespace t1k {
int jobCreate(const t1k_job_params *p, const char *refFasta, const std::set<std::string> *selected, t1k_job **out) {
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
}  // namespace t1k
extern "C" {

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
}  // extern "C"
namespace t1k {
bool loadAbundance(t1k_job *job) {
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
}  // namespace t1k
extern "C" {

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


static int runLocalOnce(t1k_job *job) {
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

// A streamed .gz input whose text left the layout the streaming reader follows behind the head it checked (a blank line between two
// records, a last record without its quality line, reads that get much shorter than the head's): the whole-file reader takes such
// text as the reference's reader does (kseq.h:94-150), so the files are opened whole and the loop starts over -- nothing of the
// failed run is kept (its partial read files are truncated by the new run).  Here and not in t1k_job_run: a caller
// that drives t1k_job_run_local / t1k_job_finish itself gets the same fallback.
int t1k_job_run_local(t1k_job *job) {
  int rc = runLocalOnce(job);
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
    rc = runLocalOnce(job);
  }
  return rc;
}


}  // extern "C"
