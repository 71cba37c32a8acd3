// t1k_amd/csrc/host/analyzer.cpp -- t1k_analyzer_main(): the post-analysis stage (Analyzer.cpp:236-733 as run-t1k:438-449 starts it)
// on top of the job layer: re-assignment to the selected alleles, novel variants, per-barcode summary.
#include "job_internal.h"

extern "C" {

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
  uint64_t nEnds = 0, nJobs = 0, nFast = 0;
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
    // (identical read-ends of the piece are assigned once: an open-addressing table over a 64-bit hash of the bases, membership decided by comparing
    // the bases themselves -- the std::unordered_map<std::string, ...> of round 5 allocated a string per read-end: 0.4 s per million pairs)
    std::vector<std::pair<const char *, uint32_t>> ends;
    size_t slots = 64;
    while (slots < (size_t)pieceEnds * 4) slots <<= 1;
    std::vector<uint32_t> table(slots, ~0u);
    auto endId = [&](const char *p, uint32_t n) -> uint32_t {
      uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
      uint32_t i = 0;
      for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0xD6E8FEB86659FD93ull; h ^= h >> 29; }
      for (; i < n; ++i) { h = (h ^ (uint8_t)p[i]) * 0x100000001B3ull; }
      h ^= h >> 32;
      for (size_t q = (size_t)h & (slots - 1);; q = (q + 1) & (slots - 1)) {
        const uint32_t e = table[q];
        if (e == ~0u) { table[q] = (uint32_t)ends.size(); ends.emplace_back(p, n); return (uint32_t)ends.size() - 1; }
        if (ends[e].second == n && memcmp(ends[e].first, p, n) == 0) return e;
      }
    };
    std::vector<uint32_t> endOf;  // (fragment - f0) * 2 + mate -> distinct read-end of the piece
    uint32_t f1 = f0;
    for (; f1 < F && ends.size() + 2 <= pieceEnds; ++f1) {
      endOf.push_back(~0u); endOf.push_back(~0u);
      if (!job->fragAssigned[f1] || !cnt[f1]) continue;
      for (int m = 0; m < (paired ? 2 : 1); ++m) {
        auto rd = readOf(f1, m);
        endOf[(size_t)(f1 - f0) * 2 + m] = endId(rd.first, rd.second);
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
      // the overlaps behind every kept assignment, and where each of them sits in its read-end's list (found again by its coordinates): per
      // fragment, by the host threads (round 6); the alignment jobs -- one per distinct (read-end, overlap) -- are numbered behind them in
      // fragment order, as before
      const uint64_t q0 = V.asgPtr[f0], nAsg = V.asgPtr[f1] - q0;
      std::vector<int64_t> at[2];   // assignment -> index of its overlap(s) in `lists`
      at[0].assign(nAsg, -1); at[1].assign(nAsg, -1);
      std::atomic<int64_t> badFrag{-1};
      std::atomic<int> badKind{0};
      parallelRanges((size_t)(f1 - f0), hostThreads(job), [&](int, size_t lo, size_t hi) {
        std::vector<int32_t> alleles;
        auto find = [&](uint32_t e, const t1k_overlap &o) -> int64_t {
          for (uint32_t i = 0; i < lc[e]; ++i) {
            const t1k_overlap &c = lists[listAt[e] + i];
            if (c.seq_idx == o.seq_idx && c.read_start == o.read_start && c.read_end == o.read_end && c.seq_start == o.seq_start && c.seq_end == o.seq_end && c.strand == o.strand)
              return (int64_t)(listAt[e] + i);
          }
          return -1;
        };
        for (size_t x = lo; x < hi; ++x) {
          const uint32_t f = f0 + (uint32_t)x;
          if (!job->fragAssigned[f] || !cnt[f]) continue;
          const uint32_t k = cnt[f];
          alleles.resize(k);
          for (uint32_t j = 0; j < k; ++j) alleles[j] = rows[rowAt[f] + j].allele_idx;
          const uint32_t e1 = endOf[x * 2], e2 = paired ? endOf[x * 2 + 1] : 0;
          if (!fragmentDetails(lists.data() + listAt[e1], lc[e1], paired ? lists.data() + listAt[e2] : nullptr, paired ? lc[e2] : 0, paired, alleles.data(), k, V.asg.data() + V.asgPtr[f])) {
            badKind = 1; badFrag = f; return;
          }
          for (uint64_t q = V.asgPtr[f]; q < V.asgPtr[f + 1]; ++q) {
            const t1k_frag_assignment &a = V.asg[q];
            const uint32_t eA = endOf[x * 2 + ((a.o1_from_r2 && !a.has_mate_pair) ? 1 : 0)];
            if ((at[0][q - q0] = find(eA, a.o1)) < 0 || (a.has_mate_pair && (at[1][q - q0] = find(endOf[x * 2 + 1], a.o2)) < 0)) { badKind = 2; badFrag = f; return; }
          }
        }
      });
      if (badKind.load() == 1) return jobFail(job, T1K_ERR_INTERNAL, "analyzer: fragment " + std::to_string(badFrag.load()) + " is assigned to an allele its read-ends' overlap lists do not hold");
      if (badKind.load() == 2) return jobFail(job, T1K_ERR_INTERNAL, "analyzer: an assignment's overlap is not in its read-end's list");
      std::vector<int64_t> jobOf(total, -1);
      struct Job { uint32_t end, idx; };
      std::vector<Job> jobs;
      std::vector<uint32_t> endOfList(total);   // list entry -> its read-end
      for (uint32_t e = 0; e < E; ++e) for (uint64_t i = listAt[e]; i < listAt[e + 1]; ++i) endOfList[i] = e;
      std::vector<int64_t> jobOfAsg[2];
      jobOfAsg[0].assign(nAsg, -1);
      jobOfAsg[1].assign(nAsg, -1);
      for (uint64_t q = 0; q < nAsg; ++q)
        for (int m = 0; m < 2; ++m) {
          const int64_t li = at[m][q];
          if (li < 0) continue;
          int64_t &slot = jobOf[li];
          if (slot < 0) { slot = (int64_t)jobs.size(); jobs.push_back({endOfList[li], (uint32_t)(li - (int64_t)listAt[endOfList[li]])}); }
          jobOfAsg[m][q] = slot;
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
      // Round 6: most of these alignments never reach the device.  Text and pattern of equal length with at most two mismatches ('N' on
      // either side matches, AlignAlgo.hpp:304-305): the ungapped alignment scores 2L - 4x >= 2L - 8, anything with a gap at most 2L - 12 (one
      // insertion AND one deletion at least) -- the diagonal is the strict optimum at every prefix, and the reference's traceback takes the
      // diagonal wherever it attains the cell (AlignAlgo.hpp:335-343): the edit string is MATCH / MISMATCH by position (checked against
      // the reference's routine on 232 363 cases).  The host threads write those strings straight into the job's edit-string store; what is
      // left (reads with an indel or three and more mismatches: a few per cent) goes through t1k_align_batch as before.  T1K_ANALYZER_NO_FAST=1:
      // everything through the device (A/B and the test that both ways agree).
      static const bool noFast = getenv("T1K_ANALYZER_NO_FAST") != nullptr;
      const size_t nJ = jobs.size();
      std::vector<uint32_t> jT(nJ), jTL(nJ), jP(nJ), jPL(nJ);
      std::vector<uint8_t> fast(nJ, 0);
      const int TH = hostThreads(job);
      parallelRanges(nJ, TH, [&](int, size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          const Job &jb = jobs[j];
          const t1k_overlap &o = lists[listAt[jb.end] + jb.idx];
          jT[j] = (uint32_t)(refOff[o.seq_idx] + (uint64_t)o.seq_start);
          jTL[j] = (uint32_t)(o.seq_end - o.seq_start + 1);
          jP[j] = (uint32_t)((o.strand == -1 ? rcAt[jb.end] : off[jb.end]) + (uint64_t)o.read_start);
          jPL[j] = (uint32_t)(o.read_end - o.read_start + 1);
          if (noFast || jTL[j] != jPL[j]) continue;
          const char *t = refText.data() + jT[j], *q = pat.data() + jP[j];
          int x = 0;
          for (uint32_t i = 0; i < jTL[j] && x <= 2; ++i) x += (t[i] != q[i] && t[i] != 'N' && q[i] != 'N') ? 1 : 0;
          fast[j] = x <= 2;
        }
      });
      // the rest: through the device, in calls of at most 2^18
      std::vector<uint32_t> slow;
      for (size_t j = 0; j < nJ; ++j) if (!fast[j]) slow.push_back((uint32_t)j);
      std::vector<std::vector<int8_t>> slowBuf;
      std::vector<uint32_t> slowOff(slow.size()), slowCall(slow.size());
      const size_t callJobs = 1u << 18;
      for (size_t j0 = 0; j0 < slow.size(); j0 += callJobs) {
        const uint32_t n = (uint32_t)std::min(callJobs, slow.size() - j0);
        std::vector<uint32_t> tOff(n), tLen(n), pOff(n), pLen(n), oOff(n), nOps(n);
        uint64_t room = 0;
        for (uint32_t i = 0; i < n; ++i) {
          const uint32_t j = slow[j0 + i];
          tOff[i] = jT[j]; tLen[i] = jTL[j]; pOff[i] = jP[j]; pLen[i] = jPL[j];
          oOff[i] = (uint32_t)room;
          room += (uint64_t)tLen[i] + pLen[i] + 2;
        }
        if (room >= (1ull << 32)) return jobFail(job, T1K_ERR_CAPACITY, "analyzer: the edit strings of one alignment call exceed 4 GB");
        slowBuf.emplace_back(room + 64);
        if ((rc = t1k_align_batch(vctx, refText.data(), tOff.data(), tLen.data(), pat.data(), pOff.data(), pLen.data(), n, nullptr, nullptr, nullptr, nullptr, slowBuf.back().data(), oOff.data(), nOps.data())) != T1K_OK)
          return jobFail(job, rc, t1k_last_error(vctx));
        for (uint32_t i = 0; i < n; ++i) { nOpsOfJob[slow[j0 + i]] = nOps[i]; slowOff[j0 + i] = oOff[i]; slowCall[j0 + i] = (uint32_t)(slowBuf.size() - 1); }
      }
      // every job's place in the store (job order, as before), then the strings, by the host threads
      {
        uint64_t at = V.ops.size();
        for (size_t j = 0; j < nJ; ++j) { if (fast[j]) nOpsOfJob[j] = jTL[j]; opsAtOfJob[j] = at; at += nOpsOfJob[j]; }
        V.ops.resize(at);
        std::vector<uint32_t> slowIdx(nJ, 0);
        for (size_t i = 0; i < slow.size(); ++i) slowIdx[slow[i]] = (uint32_t)i;
        int8_t *store = V.ops.data();
        parallelRanges(nJ, TH, [&](int, size_t lo, size_t hi) {
          for (size_t j = lo; j < hi; ++j) {
            int8_t *dst = store + opsAtOfJob[j];
            if (fast[j]) {
              const char *t = refText.data() + jT[j], *q = pat.data() + jP[j];
              for (uint32_t i = 0; i < jTL[j]; ++i) dst[i] = (t[i] != q[i] && t[i] != 'N' && q[i] != 'N') ? 1 : 0;   // EDIT_MISMATCH : EDIT_MATCH (AlignAlgo.hpp:7-8)
            } else {
              const uint32_t i = slowIdx[j];
              memcpy(dst, slowBuf[slowCall[i]].data() + slowOff[i], nOpsOfJob[j]);
            }
          }
        });
      }
      nFast += nJ - slow.size();
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
    fprintf(stderr, "[t1k analyzer] variant pass: rows + EM %.1f ms; %llu distinct read-ends re-assigned in %.1f ms, overlaps chosen in %.1f ms, %llu alignments (%llu of them on the host: equal lengths, at most two mismatches) in %.1f ms; "
                    "VariantCaller %.1f ms (%zu assignments, %zu variants); %.1f ms in all\n", tv1 - tv0, (unsigned long long)nEnds, msAssign, msDetails, (unsigned long long)nJobs, (unsigned long long)nFast, msAlign,
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
