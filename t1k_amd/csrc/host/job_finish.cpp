// t1k_amd/csrc/host/job_finish.cpp -- the genotyper stage as a job, second half: t1k_job_finish (equivalence classes, EM, likelihood pruning, allele
// selection: Genotyper.hpp:1076-2090 through host/genotype.cpp), t1k_job_run, the group-table exchange of a sharded job, the variant-calling entry
// points of the analyzer stage (host/variants.cpp behind the C ABI) and the job's result getters.
#include "job_internal.h"

extern "C" {

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

}  // extern "C"
