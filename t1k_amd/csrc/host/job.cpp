// t1k_amd/csrc/host/job.cpp -- the whole genotyper stage as a job: the t1k_job_* C ABI and t1k_genotyper_main(), the
// argv-compatible replacement of the reference's genotyper executable (Genotyper.cpp:194-738, invoked by run-t1k:430,434).
#include <getopt.h>
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
  t1k_ctx *ctx = nullptr;
  std::vector<t1k_ctx *> more;      // further pipelines on the same GPU (own stream and batch arenas): several batches in flight
  // reads: read-end 2f / 2f+1 are the mates of fragment f (single-end: read-end f)
  bool paired = false;
  uint32_t nFrag = 0;
  std::string ends;                 // concatenated ASCII of all read-ends, interleaved
  std::vector<uint64_t> endOff;     // [nEnds + 1]
  std::vector<std::string> id1, id2, barcode;
  std::vector<uint8_t> hasN, fragAssigned;
  bool hasBarcode = false;
  int maxReadLen = 0;
  bool staged = false, ran = false, localDone = false;
  std::vector<char> whitelist;      // per allele, empty = everything allowed
  std::string abundanceFile;
  std::string assignText;           // --outputReadAssignment rows
  t1k_stats stats{};
  t1k_allreduce_fn allreduce = nullptr;
  void *allreduceUser = nullptr;
  double msUpload = 0, msDevice = 0, msHost = 0, msEm = 0;
};

static double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int jobFail(t1k_job *job, int code, const std::string &msg) {
  if (job) job->err = msg;
  return code;
}

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

int t1k_job_create(const t1k_job_params *p, const char *refFasta, t1k_job **out) {
  if (!out || !refFasta) return T1K_ERR_ARG;
  *out = nullptr;
  t1k_job *job = new t1k_job();
  if (p) job->prm = *p; else t1k_job_params_default(&job->prm);
  *out = job;  // handed back even on failure so the caller can read the message
  if (!job->ref.load(refFasta, job->prm.allele_digit_units, job->prm.allele_delimiter, job->err)) return T1K_ERR_IO;
  job->gt.ref = &job->ref;
  job->gt.prm = job->prm;
  if (job->prm.device < 0) return T1K_OK;  // host-only job: group bookkeeping for tests / merging, no device work possible
  int rc = t1k_ctx_create(job->prm.device, &job->prm.dev, &job->ctx);
  if (rc != T1K_OK) return jobFail(job, rc, "cannot create a GPU context on device " + std::to_string(job->prm.device) + " (this build has no CPU path)");
  // upload the reference
  const RefSet &R = job->ref;
  std::string blob;
  std::vector<uint64_t> off(R.seqs.size() + 1, 0);
  std::vector<uint8_t> ex;
  for (size_t a = 0; a < R.seqs.size(); ++a) {
    blob += R.seqs[a];
    off[a + 1] = blob.size();
    ex.insert(ex.end(), R.exon[a].begin(), R.exon[a].end());
  }
  rc = t1k_ref_upload(job->ctx, blob.data(), off.data(), ex.data(), (uint32_t)R.seqs.size());
  if (rc != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  // Most kernels of the path are latency-bound; a second, independent pipeline (context, stream, arenas) on the same GPU lets
  // the hardware overlap two batches.  T1K_PIPELINES=1 turns it off.
  const char *pl = getenv("T1K_PIPELINES");
  const int nPipe = pl ? std::max(1, std::min(8, atoi(pl))) : 4;
  for (int i = 1; i < nPipe; ++i) {
    t1k_ctx *c = nullptr;
    rc = t1k_ctx_create(job->prm.device, &job->prm.dev, &c);
    if (rc == T1K_OK) rc = t1k_ref_share(c, job->ctx);  // the pipelines read one copy of the reference and index
    if (rc != T1K_OK) { if (c) t1k_ctx_destroy(c); break; }  // not enough memory: run with the pipelines we have
    job->more.push_back(c);
  }
  job->gt.ref = &job->ref;
  job->gt.prm = job->prm;
  return T1K_OK;
}

void t1k_job_destroy(t1k_job *job) {
  if (!job) return;
  for (t1k_ctx *c : job->more) t1k_ctx_destroy(c);  // before job->ctx, whose reference and reads they alias
  if (job->ctx) t1k_ctx_destroy(job->ctx);
  delete job;
}

const char *t1k_job_last_error(const t1k_job *job) { return job ? job->err.c_str() : "no job"; }
t1k_ctx *t1k_job_ctx(t1k_job *job) { return job ? job->ctx : nullptr; }

static void addFragment(t1k_job *job, const std::string &s1, const std::string *s2) {
  job->ends += s1;
  job->endOff.push_back(job->ends.size());
  bool n = s1.find('N') != std::string::npos;
  job->maxReadLen = std::max<int>(job->maxReadLen, (int)s1.size());
  if (s2) {
    job->ends += *s2;
    job->endOff.push_back(job->ends.size());
    n = n || s2->find('N') != std::string::npos;
    job->maxReadLen = std::max<int>(job->maxReadLen, (int)s2->size());
  }
  job->hasN.push_back(n ? 1 : 0);
  ++job->nFrag;
}

static void clearReads(t1k_job *job) {
  job->ends.clear(); job->endOff.assign(1, 0); job->id1.clear(); job->id2.clear(); job->barcode.clear(); job->hasN.clear();
  job->nFrag = 0; job->maxReadLen = 0; job->staged = false; job->ran = false; job->hasBarcode = false;
}

int t1k_job_load_reads(t1k_job *job, const char *file1, const char *file2, const char *barcodeFile) {
  if (!job || !file1) return T1K_ERR_ARG;
  clearReads(job);
  std::vector<SeqRec> r1, r2, bc;
  if (!readSeqFile(file1, r1, job->err)) return T1K_ERR_IO;
  job->paired = file2 != nullptr;
  if (file2 && !readSeqFile(file2, r2, job->err)) return T1K_ERR_IO;
  if (file2 && r2.size() != r1.size()) return jobFail(job, T1K_ERR_IO, "mate files hold different numbers of reads");
  if (barcodeFile) {
    if (!readSeqFile(barcodeFile, bc, job->err)) return T1K_ERR_IO;
    if (bc.size() != r1.size()) return jobFail(job, T1K_ERR_IO, "barcode file and read file hold different numbers of records");
    job->hasBarcode = true;
  }
  for (size_t i = 0; i < r1.size(); ++i) {
    if (barcodeFile && bc[i].seq == "missing_barcode") continue;  // dropped with its mate (Genotyper.cpp:376-381)
    addFragment(job, r1[i].seq, file2 ? &r2[i].seq : nullptr);
    job->id1.push_back(r1[i].id);
    if (file2) job->id2.push_back(r2[i].id);
    if (barcodeFile) job->barcode.push_back(bc[i].seq);
  }
  return T1K_OK;
}

int t1k_job_set_reads(t1k_job *job, const char *seq1, const uint64_t *off1, const char *seq2, const uint64_t *off2, uint32_t nFragments) {
  if (!job || !seq1 || !off1 || (seq2 && !off2)) return T1K_ERR_ARG;
  clearReads(job);
  job->paired = seq2 != nullptr;
  for (uint32_t i = 0; i < nFragments; ++i) {
    std::string a(seq1 + off1[i], seq1 + off1[i + 1]);
    if (seq2) { std::string b(seq2 + off2[i], seq2 + off2[i + 1]); addFragment(job, a, &b); }
    else addFragment(job, a, nullptr);
  }
  return T1K_OK;
}

int t1k_job_stage_reads(t1k_job *job) {
  if (!job || !job->ctx) return T1K_ERR_STATE;
  double t0 = nowMs();
  uint32_t nEnds = (uint32_t)(job->endOff.size() - 1);
  int rc = t1k_reads_upload(job->ctx, job->ends.data(), job->endOff.data(), nullptr, nEnds);
  if (rc != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  for (t1k_ctx *c : job->more)
    if ((rc = t1k_reads_share(c, job->ctx)) != T1K_OK) return jobFail(job, rc, t1k_last_error(c));  // one packed copy of the reads
  job->staged = true;
  job->msUpload = nowMs() - t0;
  return T1K_OK;
}

int t1k_job_set_allreduce(t1k_job *job, t1k_allreduce_fn cb, void *user) {
  if (!job) return T1K_ERR_ARG;
  job->allreduce = cb; job->allreduceUser = user;
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

int t1k_job_run_local(t1k_job *job) {
  if (!job || !job->ctx) return jobFail(job, T1K_ERR_STATE, "this job has no GPU context (device = -1): it cannot run");
  int rc;
  if (!job->staged && (rc = t1k_job_stage_reads(job)) != T1K_OK) return rc;
  // fresh state (a job may be run repeatedly, e.g. by the benchmark)
  Genotyper &gt = job->gt;
  gt.groupPtr.assign(1, 0); gt.groupEnt.clear(); gt.groupOfHash.clear(); gt.assignedFragments = 0; gt.emIterations = 0;
  gt.readLength = job->maxReadLen;  // Genotyper.cpp:443
  for (auto &a : job->ref.al) { a.rank = -1; a.quality = -1; a.abundance = a.ecAbundance = 0; a.ec = -1; a.missingCov = 0; }
  job->fragAssigned.assign(job->nFrag, 0);
  job->assignText.clear();
  memset(&job->stats, 0, sizeof(job->stats));
  if ((rc = t1k_coverage_reset(job->ctx)) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  for (t1k_ctx *c : job->more)
    if ((rc = t1k_coverage_reset(c)) != T1K_OK) return jobFail(job, rc, t1k_last_error(c));
  const uint32_t F = job->nFrag;
  const uint32_t per = job->paired ? 2 : 1;
  double tDev = 0, tHost = 0;
  // One worker per pipeline.  A worker takes the next contiguous range of fragments, runs the device stages on its context,
  // downloads the rows, and then -- when every earlier range has been absorbed (coalescing is order-dependent) --
  // does the host half (whitelist filter, assignment text, read-group coalescing) while the other worker's batch is on the GPU.
  struct HostBatch {
    uint32_t b0 = 0, nb = 0;
    std::vector<uint32_t> rowCounts;
    std::vector<uint8_t> assigned;
    std::vector<t1k_row_entry> rows;
  };
  struct Shared {
    std::mutex m;
    std::condition_variable cv;
    uint32_t next = 0, absorbNext = 0, batch = 16384;
    int err = T1K_OK;
    std::string errMsg;
  } sh;
  sh.batch = job->prm.batch_fragments > 0 ? (uint32_t)job->prm.batch_fragments : 16384u;
  if (const char *eb = getenv("T1K_BATCH")) sh.batch = std::max(64, atoi(eb));  // tuning aid
  auto absorb = [job, &gt, &tHost](HostBatch &hb) {
    const double t1 = nowMs();
    uint64_t p = 0;
    for (uint32_t i = 0; i < hb.nb; ++i) {
      job->fragAssigned[hb.b0 + i] = hb.assigned[i];
      uint32_t n = hb.rowCounts[i];
      t1k_row_entry *row = hb.rows.data() + p;
      p += n;
      if (!job->whitelist.empty()) {  // SetReadAssignments skips alleles outside the whitelist (Genotyper.hpp:822-823)
        uint32_t w = 0;
        for (uint32_t j = 0; j < n; ++j)
          if (job->whitelist[row[j].allele_idx]) row[w++] = row[j];
        n = w;
      }
      if (job->prm.output_read_assignment) {
        const std::string id = job->id1.empty() ? "r" + std::to_string(hb.b0 + i) : job->id1[hb.b0 + i];
        for (uint32_t j = 0; j < n; ++j)
          job->assignText += id + "\t" + job->ref.al[row[j].allele_idx].name + "\t" + std::to_string(row[j].start) + "\t" + std::to_string(row[j].end) + "\n";
      }
      gt.coalesce(row, n);
    }
    tHost += nowMs() - t1;
  };
  // device stages of fragments [b0, b0 + nb) on one context; on a capacity error nothing has been committed: split the range
  std::function<int(t1k_ctx *, uint32_t, uint32_t, std::vector<HostBatch> &, std::string &)> runRange =
      [&](t1k_ctx *ctx, uint32_t b0, uint32_t nb, std::vector<HostBatch> &out, std::string &msg) -> int {
    int r = t1k_assign_range(ctx, (uint64_t)b0 * per, nb * per);
    if (r == T1K_ERR_CAPACITY && nb > 64) {
      { std::lock_guard<std::mutex> g(sh.m); sh.batch = std::max<uint32_t>(64, std::min(sh.batch, nb / 2)); }
      if ((r = runRange(ctx, b0, nb / 2, out, msg)) != T1K_OK) return r;
      return runRange(ctx, b0 + nb / 2, nb - nb / 2, out, msg);
    }
    if (r != T1K_OK) { msg = t1k_last_error(ctx); return r; }
    std::vector<uint32_t> e1(nb), e2(nb);
    for (uint32_t i = 0; i < nb; ++i) { e1[i] = i * per; e2[i] = i * per + 1; }
    if ((r = t1k_pair_batch(ctx, e1.data(), job->paired ? e2.data() : nullptr, job->hasN.data() + b0, nb)) != T1K_OK) { msg = t1k_last_error(ctx); return r; }
    out.emplace_back();
    HostBatch &hb = out.back();
    hb.b0 = b0; hb.nb = nb;
    hb.rowCounts.resize(nb); hb.assigned.resize(nb);
    uint64_t total = 0;
    if ((r = t1k_rows_download(ctx, hb.rowCounts.data(), hb.assigned.data(), nullptr, 0, &total)) != T1K_OK) { msg = t1k_last_error(ctx); return r; }
    hb.rows.resize(total);
    if ((r = t1k_rows_download(ctx, hb.rowCounts.data(), hb.assigned.data(), hb.rows.data(), total, &total)) != T1K_OK) { msg = t1k_last_error(ctx); return r; }
    t1k_stats st;
    t1k_stats_get(ctx, &st);
    std::lock_guard<std::mutex> g(sh.m);
    job->stats.read_ends += st.read_ends; job->stats.lookups += st.lookups; job->stats.postings += st.postings; job->stats.hits += st.hits;
    job->stats.groups += st.groups; job->stats.candidates += st.candidates; job->stats.extended += st.extended; job->stats.near_best += st.near_best;
    job->stats.dp_calls += st.dp_calls; job->stats.ms_chain += st.ms_chain; job->stats.ms_extend += st.ms_extend; job->stats.ms_select += st.ms_select;
    job->stats.ms_fullalign += st.ms_fullalign; job->stats.ms_pair += st.ms_pair; job->stats.ms_seed += st.ms_seed; job->stats.batches += 1;
    job->stats.rows += total;
    return T1K_OK;
  };
  auto worker = [&](t1k_ctx *ctx) {
    for (;;) {
      uint32_t b0, nb;
      {
        std::lock_guard<std::mutex> g(sh.m);
        if (sh.err != T1K_OK || sh.next >= F) return;
        b0 = sh.next; nb = std::min(sh.batch, F - b0); sh.next += nb;
      }
      std::vector<HostBatch> done;
      std::string msg;
      const int r = runRange(ctx, b0, nb, done, msg);
      std::unique_lock<std::mutex> lk(sh.m);
      sh.cv.wait(lk, [&] { return sh.absorbNext == b0 || sh.err != T1K_OK; });
      if (r != T1K_OK && sh.err == T1K_OK) { sh.err = r; sh.errMsg = msg; }
      if (sh.err != T1K_OK) { sh.cv.notify_all(); return; }
      lk.unlock();
      for (auto &hb : done) absorb(hb);  // this worker holds the turn: nobody else absorbs until absorbNext moves on
      lk.lock();
      sh.absorbNext = b0 + nb;
      sh.cv.notify_all();
    }
  };
  {
    const double t0 = nowMs();
    std::vector<std::thread> others;
    for (t1k_ctx *c : job->more) others.emplace_back(worker, c);
    worker(job->ctx);
    for (auto &t : others) t.join();
    tDev = nowMs() - t0;  // wall time of the batch loop; the host half of the batches is hidden inside it
    tHost = 0;
  }
  if (sh.err != T1K_OK) return jobFail(job, sh.err, sh.errMsg);
  for (t1k_ctx *c : job->more)
    if ((rc = t1k_coverage_absorb(job->ctx, c)) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  job->msDevice = tDev; job->msHost = tHost;
  job->localDone = true;
  return T1K_OK;
}

int t1k_job_finish(t1k_job *job, uint64_t emGroupBegin, uint64_t emGroupEnd) {
  if (!job || !job->ctx || !job->localDone) return jobFail(job, T1K_ERR_STATE, "t1k_job_finish: t1k_job_run_local has not completed");
  Genotyper &gt = job->gt;
  int rc;
  double t2 = nowMs();
  std::vector<int32_t> cov(job->ref.al.size());  // per allele: exon positions with too little coverage
  if ((rc = t1k_missing_coverage(job->ctx, cov.data())) != T1K_OK) return jobFail(job, rc, t1k_last_error(job->ctx));
  gt.finalize(cov);
  double t3 = nowMs();
  if (!job->abundanceFile.empty()) {
    if (!loadAbundance(job)) return T1K_ERR_IO;
  } else {
    if (gt.quantify(job->ctx, job->allreduce, job->allreduceUser, job->err, emGroupBegin, emGroupEnd) < 0) return T1K_ERR_DEVICE;
  }
  double t4 = nowMs();
  gt.dropUnlikely();
  double t4b = nowMs();
  gt.select();
  double t5 = nowMs();
  job->msHost += (t3 - t2) + (t5 - t4); job->msEm = t4 - t3;
  job->stats.ms_total = job->msDevice + job->msHost + job->msEm;
  job->stats.ms_em = job->msEm;
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k job] device+download %.1f ms, host coalesce+finalize %.1f ms, EM %.1f ms, dropUnlikely %.1f ms, select %.1f ms\n", job->msDevice,
            job->msHost, t4 - t3, t4b - t4, t5 - t4b);
  job->ran = true;
  return T1K_OK;
}

int t1k_job_run(t1k_job *job) {
  int rc = t1k_job_run_local(job);
  if (rc != T1K_OK) return rc;
  return t1k_job_finish(job, 0, ~0ull);
}

// group table <-> byte string: [u64 nGroups][u64 nEntries][u64 assignedFragments][u64 groupPtr[nGroups+1]][GroupEntry entries[nEntries]]
int t1k_job_groups_serialize(t1k_job *job, void *buf, uint64_t cap, uint64_t *needed) {
  if (!job) return T1K_ERR_ARG;
  const Genotyper &gt = job->gt;
  const uint64_t G = gt.nGroups(), N = gt.groupEnt.size();
  const uint64_t bytes = 24 + (G + 1) * 8 + N * sizeof(GroupEntry);
  if (needed) *needed = bytes;
  if (!buf) return T1K_OK;
  if (cap < bytes) return jobFail(job, T1K_ERR_ARG, "group buffer too small");
  uint8_t *p = (uint8_t *)buf;
  uint64_t head[3] = {G, N, gt.assignedFragments};
  memcpy(p, head, 24); p += 24;
  memcpy(p, gt.groupPtr.data(), (G + 1) * 8); p += (G + 1) * 8;
  if (N) memcpy(p, gt.groupEnt.data(), N * sizeof(GroupEntry));
  return T1K_OK;
}

int t1k_job_groups_reset(t1k_job *job) {
  if (!job) return T1K_ERR_ARG;
  Genotyper &gt = job->gt;
  gt.groupPtr.assign(1, 0); gt.groupEnt.clear(); gt.groupOfHash.clear(); gt.assignedFragments = 0;
  return T1K_OK;
}

int t1k_job_groups_absorb(t1k_job *job, const void *buf, uint64_t len) {
  if (!job || !buf || len < 24) return T1K_ERR_ARG;
  const uint8_t *p = (const uint8_t *)buf;
  uint64_t head[3];
  memcpy(head, p, 24);
  const uint64_t G = head[0], N = head[1];
  if (len < 24 + (G + 1) * 8 + N * sizeof(GroupEntry)) return jobFail(job, T1K_ERR_ARG, "truncated group table");
  const uint64_t *gp = (const uint64_t *)(p + 24);
  const GroupEntry *ent = (const GroupEntry *)(p + 24 + (G + 1) * 8);
  std::vector<GroupEntry> row;
  for (uint64_t g = 0; g < G; ++g) {
    row.assign(ent + gp[g], ent + gp[g + 1]);  // copy: the source may be unaligned for GroupEntry
    job->gt.absorb(row.data(), (uint32_t)row.size());
  }
  job->gt.assignedFragments += head[2];
  return T1K_OK;
}

int t1k_job_coalesce_rows(t1k_job *job, const t1k_row_entry *rows, const uint32_t *rowCounts, uint32_t nFragments) {
  if (!job || !rowCounts || (!rows && nFragments)) return T1K_ERR_ARG;
  std::vector<t1k_row_entry> tmp;
  uint64_t p = 0;
  for (uint32_t f = 0; f < nFragments; ++f) {
    tmp.assign(rows + p, rows + p + rowCounts[f]);
    p += rowCounts[f];
    job->gt.coalesce(tmp.data(), (uint32_t)tmp.size());
  }
  return T1K_OK;
}

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
  if (fragments) *fragments = job->nFrag;
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
  if (!job || !prefix || !job->ran) return T1K_ERR_STATE;
  const std::string pfx = prefix;
  std::string s;
  for (size_t g = 0; g < job->ref.geneName.size(); ++g) s += job->gt.geneLine((int)g);
  if (!writeText(pfx + "_genotype.tsv", s, job->err)) return T1K_ERR_IO;
  if (!writeText(pfx + "_allele.tsv", job->gt.alleleLines(), job->err)) return T1K_ERR_IO;
  if (job->prm.output_read_assignment && !writeText(pfx + "_assign.tsv", job->assignText, job->err)) return T1K_ERR_IO;
  // reads with at least one fragment assignment (Genotyper.cpp:680-718)
  const uint32_t per = job->paired ? 2 : 1;
  for (uint32_t m = 0; m < per; ++m) {
    std::string path = job->paired ? pfx + (m == 0 ? "_aligned_1.fa" : "_aligned_2.fa") : pfx + "_aligned.fa";
    FILE *fp = fopen(path.c_str(), "w");
    if (!fp) return jobFail(job, T1K_ERR_IO, "cannot write " + path);
    for (uint32_t f = 0; f < job->nFrag; ++f) {
      if (!job->fragAssigned[f]) continue;
      const std::vector<std::string> &ids = m == 0 ? job->id1 : job->id2;
      uint64_t e = (uint64_t)f * per + m;
      std::string id = ids.empty() ? "r" + std::to_string(f) : ids[f];
      fprintf(fp, ">%s\n%.*s\n", id.c_str(), (int)(job->endOff[e + 1] - job->endOff[e]), job->ends.data() + job->endOff[e]);
    }
    fclose(fp);
  }
  if (job->hasBarcode) {
    FILE *fp = fopen((pfx + "_aligned_bc.fa").c_str(), "w");
    if (!fp) return jobFail(job, T1K_ERR_IO, "cannot write " + pfx + "_aligned_bc.fa");
    for (uint32_t f = 0; f < job->nFrag; ++f)
      if (job->fragAssigned[f]) fprintf(fp, ">%s\n%s\n", job->id1[f].c_str(), job->barcode[f].c_str());
    fclose(fp);
  }
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
    "\t--device INT: GPU ordinal (default: $T1K_DEVICE or 0)\n";

int t1k_genotyper_main(int argc, char **argv) {
  if (argc <= 1) { fprintf(stderr, "%s", kUsage); return 0; }  // Genotyper.cpp:199-203
  static struct option longOpts[] = {{"frac", required_argument, 0, 1000}, {"cov", required_argument, 0, 1001}, {"crossGeneRate", required_argument, 0, 1002},
                                     {"barcode", required_argument, 0, 1003}, {"relaxIntronAlign", no_argument, 0, 1004},
                                     {"alleleDigitUnits", required_argument, 0, 1005}, {"alleleDelimiter", required_argument, 0, 1006},
                                     {"alleleWhitelist", required_argument, 0, 1007}, {"outputReadAssignment", no_argument, 0, 1008},
                                     {"squaremMinAlpha", required_argument, 0, 1009}, {"device", required_argument, 0, 1010}, {0, 0, 0, 0}};
  t1k_job_params p;
  t1k_job_params_default(&p);
  if (const char *d = getenv("T1K_DEVICE")) p.device = atoi(d);
  std::string refFile, f1, f2, single, prefix = "t1k", barcode, whitelistFile, abundance;
  optind = 1;
  int c, idx = 0;
  while ((c = getopt_long(argc, argv, "f:a:u:1:2:o:t:n:s:b:", longOpts, &idx)) != -1) {
    switch (c) {
      case 'f': refFile = optarg; break;
      case 'a': abundance = optarg; break;
      case 'u': single = optarg; break;
      case '1': f1 = optarg; break;
      case '2': f2 = optarg; break;
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
      default: fprintf(stderr, "%s", kUsage); return EXIT_FAILURE;
    }
  }
  if (refFile.empty()) { fprintf(stderr, "Need to use -f to specify the reference sequences.\n"); return EXIT_FAILURE; }
  if (p.dev.max_assign_cnt == 0) p.dev.max_assign_cnt = -1;  // "-n 0" disables the cap in the reference (maxAssignCnt > 0 test)
  t1k_job *job = nullptr;
  int rc = t1k_job_create(&p, refFile.c_str(), &job);
  if (rc != T1K_OK) {
    fprintf(stderr, "genotyper: %s\n", job ? t1k_job_last_error(job) : "initialisation failed");
    if (job && job->ref.al.empty()) fprintf(stderr, "Need to use -f to specify the reference sequences.\n");
    t1k_job_destroy(job);
    return EXIT_FAILURE;
  }
  if (!whitelistFile.empty()) {  // Genotyper::SetAlleleWhitelist (Genotyper.hpp:684-705): whole major-allele series
    FILE *fp = fopen(whitelistFile.c_str(), "r");
    if (!fp) { fprintf(stderr, "genotyper: cannot open %s\n", whitelistFile.c_str()); t1k_job_destroy(job); return EXIT_FAILURE; }
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
    job->whitelist.assign(job->ref.al.size(), 0);
    for (size_t a = 0; a < job->ref.al.size(); ++a) job->whitelist[a] = majors.count(job->ref.al[a].major) ? 1 : 0;
  }
  job->abundanceFile = abundance;
  const bool paired = !f2.empty();
  const std::string &first = !f1.empty() ? f1 : single;
  if (first.empty()) { fprintf(stderr, "genotyper: no read file given (-u, or -1 and -2)\n"); t1k_job_destroy(job); return EXIT_FAILURE; }
  rc = t1k_job_load_reads(job, first.c_str(), paired ? f2.c_str() : nullptr, barcode.empty() ? nullptr : barcode.c_str());
  if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); t1k_job_destroy(job); return EXIT_FAILURE; }
  logLine("Found %d read fragments. Start read assignment.", (int)job->nFrag);
  rc = t1k_job_run(job);
  if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); t1k_job_destroy(job); return EXIT_FAILURE; }
  logLine("Finish read end assignments.");
  const double groups = (double)job->gt.nGroups();
  logLine("Finish read fragment assignments. %d read fragments can be assigned (average %.2lf alleles/read).", (int)job->gt.assignedFragments,
          job->gt.sumAssign / groups);
  if (abundance.empty()) logLine("Finish allele quantification in %d EM iterations.", job->gt.emIterations);
  rc = t1k_job_write_outputs(job, prefix.c_str());
  if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); t1k_job_destroy(job); return EXIT_FAILURE; }
  logLine("Genotyping finishes.");
  t1k_job_destroy(job);
  return 0;
}

}  // extern "C"
