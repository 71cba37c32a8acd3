// t1k_amd/csrc/host/job_output.cpp -- the writers of the job layer: the *_aligned_{1,2,bc}.fa files (Genotyper.cpp:680-718), written beside the
// device loop / the EM, and the tables (Genotyper.cpp:653-678) behind t1k_job_write_outputs.
#include "job_internal.h"

namespace t1k {
// ">id\nSEQ\n" of every assigned fragment (Genotyper.cpp:680-718), formatted by the host threads straight from the mapped input and
// written with pwrite at precomputed offsets; what = 0 / 1: the mate's sequence, 2: the barcode.
// Two steps: the plan (bytes per host-thread piece; for ranks that each indexed their own reads also the rank's offset in the
// shared file -- one small all-gather, and rank 0 creates the file before it) and the writing itself, which needs no communication
// and so may run beside the EM.

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
bool streamOpen(t1k_job *job, const std::string &pfx) {
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
void streamClose(t1k_job *job, bool removeFiles) {
  for (auto &o : job->stream) {
    if (o.fd >= 0) ::close(o.fd);
    if (removeFiles) ::unlink(o.path.c_str());
  }
  job->stream.clear();
}
// local fragments [fLo, fHi): flags must be in job->fragAssigned
bool streamAppend(t1k_job *job, uint32_t fLo, uint32_t fHi, bool besideLoop) {
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
bool planAlignedFiles(t1k_job *job, const std::string &pfx, std::vector<AlignedPlan> &plans) {
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
bool writePlannedFiles(t1k_job *job, const std::vector<AlignedPlan> &plans) {
  std::vector<char> ok(plans.size(), 1);
  std::vector<std::thread> th;
  for (size_t i = 1; i < plans.size(); ++i) th.emplace_back([&, i] { ok[i] = writeAligned(job, plans[i]) ? 1 : 0; });
  ok[0] = writeAligned(job, plans[0]) ? 1 : 0;
  for (auto &t : th) t.join();
  for (char o : ok) if (!o) return false;
  return true;
}
// who writes: rank 0 when every rank holds the whole input; every rank its own part when each indexed only its own reads
bool writesAligned(const t1k_job *job) { return job->rank == 0 || (job->in && job->in->sharded); }
}  // namespace t1k

extern "C" {

int t1k_job_set_output_prefix(t1k_job *job, const char *prefix) {
  if (!job) return T1K_ERR_ARG;
  job->outPrefix = prefix ? prefix : "";
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

}  // extern "C"
