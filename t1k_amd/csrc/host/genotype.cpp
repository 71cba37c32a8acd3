// t1k_amd/csrc/host/genotype.cpp -- read-group coalescing, equivalence classes, the SQUAREM control loop around the
// device E-step, likelihood pruning, allele selection and the TSV text of the genotyper stage.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include "t1k_host.h"

#include <chrono>
#include <thread>

namespace {
double hostNowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// static-chunk parallel loop over [0, n) on the host cores (the per-allele / per-class post-processing is embarrassingly parallel)
template <class F>
void parallelFor(size_t n, F fn, size_t serialBelow = 256) {
  unsigned T = std::thread::hardware_concurrency();
  if (T > 32) T = 32;
  if (T < 2 || n < serialBelow || n < 2) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::vector<std::thread> th;
  const size_t chunk = (n + T - 1) / T;
  for (unsigned t = 0; t < T; ++t) {
    const size_t b = t * chunk, e = std::min(n, b + chunk);
    if (b >= e) break;
    th.emplace_back([b, e, &fn] { for (size_t i = b; i < e; ++i) fn(i); });
  }
  for (auto &x : th) x.join();
}
// the same over items of very unequal cost (an allele's list holds 1 .. 10^5 entries, long ones next to each other): the threads
// take `grain` items at a time from a shared counter
template <class F>
void parallelForDynamic(size_t n, size_t grain, F fn) {
  unsigned T = std::thread::hardware_concurrency();
  if (T > 32) T = 32;
  if (T < 2 || n <= grain) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<size_t> next{0};
  auto work = [&] {
    for (;;) {
      const size_t b = next.fetch_add(grain);
      if (b >= n) break;
      const size_t e = std::min(n, b + grain);
      for (size_t i = b; i < e; ++i) fn(i);
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
  work();
  for (auto &x : th) x.join();
}
}  // namespace

namespace t1k {

// ------------------------------------------------------------------------------------------------------------------
// coalescing (Genotyper::CoalesceReadAssignments, Genotyper.hpp:841-908): fragments with the same allele set share one
// read group; group ids follow first appearance; weights are floats accumulated in fragment order
// ------------------------------------------------------------------------------------------------------------------
void Genotyper::coalesce(t1k_row_entry *row, uint32_t n, uint32_t fragment) {
  if (n == 0) return;
  ++assignedFragments;
  std::sort(row, row + n, [](const t1k_row_entry &a, const t1k_row_entry &b) { return a.allele_idx < b.allele_idx; });
  uint64_t h = 1469598103934665603ull ^ n;
  for (uint32_t j = 0; j < n; ++j) { h ^= (uint64_t)(uint32_t)row[j].allele_idx; h *= 1099511628211ull; h ^= h >> 29; }
  std::vector<uint32_t> &bucket = groupOfHash[h];
  for (uint32_t gid : bucket) {
    uint64_t b = groupPtr[gid];
    if (groupPtr[gid + 1] - b != n) continue;
    bool same = true;
    for (uint32_t j = 0; j < n && same; ++j) same = groupEnt[b + j].allele == row[j].allele_idx;
    if (!same) continue;
    for (uint32_t j = 0; j < n; ++j) {
      GroupEntry &g = groupEnt[b + j];
      if (row[j].qual == 1) {
        if (row[j].start < g.start) g.start = row[j].start;
        if (row[j].end < g.end) g.end = row[j].start;  // sic: Genotyper.hpp:893-894
      }
      g.weight += row[j].weight;
      g.adjustWeight += row[j].adjust_weight;
    }
    return;
  }
  bucket.push_back((uint32_t)nGroups());
  for (uint32_t j = 0; j < n; ++j) groupEnt.push_back(GroupEntry{row[j].allele_idx, row[j].start, row[j].end, row[j].weight, row[j].adjust_weight});
  groupPtr.push_back(groupEnt.size());
  groupFirst.push_back(fragment);
}

void Genotyper::setGroupsMerged(const std::vector<uint32_t> &sizes, const GroupVec &entries, const std::vector<uint32_t> &first) {
  const size_t G = sizes.size();
  std::vector<uint64_t> at(G + 1, 0);
  for (size_t g = 0; g < G; ++g) at[g + 1] = at[g] + sizes[g];
  std::vector<uint32_t> order(G);
  for (size_t g = 0; g < G; ++g) order[g] = (uint32_t)g;
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return first[x] < first[y]; });  // first fragments are distinct
  groupPtr.assign(G + 1, 0);
  groupEnt.resize(entries.size());
  groupFirst.resize(G);
  groupOfHash.clear();
  for (size_t i = 0; i < G; ++i) groupPtr[i + 1] = groupPtr[i] + sizes[order[i]];
  parallelFor(G, [&](size_t i) {
    const uint32_t g = order[i];
    groupFirst[i] = first[g];
    std::copy(entries.begin() + at[g], entries.begin() + at[g + 1], groupEnt.begin() + groupPtr[i]);
  });
}

// ------------------------------------------------------------------------------------------------------------------
// FinalizeReadAssignments -> BuildAlleleEquivalentClass + missing coverage (Genotyper.hpp:912-939, 1072-1139;
// SeqSet::GetSeqMissingBaseCoverage SeqSet.hpp:2717-2755)
// ------------------------------------------------------------------------------------------------------------------
void Genotyper::finalize(const std::vector<int32_t> &missing) {
  const double tf0 = hostNowMs();
  RefSet &R = *ref;
  const int A = (int)R.al.size();
  const int G = (int)nGroups();
  sumAssign = 0;
  for (int g = 0; g < G; ++g) sumAssign += (double)(groupPtr[g + 1] - groupPtr[g]);
  // allele -> (group, slot) lists in group order: the groups are cut into one contiguous piece per thread; per-piece counts give
  // every piece its own range inside each allele's list, so the pieces fill in parallel and the order stays the sequential one
  {
    unsigned T = std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if (T < 1 || G < 4096) T = 1;
    const size_t piece = ((size_t)G + T - 1) / T;
    std::vector<std::vector<uint32_t>> cnt(T, std::vector<uint32_t>((size_t)A, 0));
    parallelFor(T, [&](size_t t) {
      const size_t g0 = t * piece, g1 = std::min((size_t)G, g0 + piece);
      for (size_t g = g0; g < g1; ++g)
        for (uint64_t p = groupPtr[g]; p < groupPtr[g + 1]; ++p) ++cnt[t][groupEnt[p].allele];
    }, 1);
    inAllele.assign(A, {});
    parallelForDynamic((size_t)A, 32, [&](size_t a) {
      uint32_t run = 0;
      for (unsigned t = 0; t < T; ++t) { const uint32_t c = cnt[t][a]; cnt[t][a] = run; run += c; }
      inAllele[a].resize(run);
    });
    parallelFor(T, [&](size_t t) {
      const size_t g0 = t * piece, g1 = std::min((size_t)G, g0 + piece);
      for (size_t g = g0; g < g1; ++g)
        for (uint64_t p = groupPtr[g]; p < groupPtr[g + 1]; ++p) {
          const int a = groupEnt[p].allele;
          inAllele[a][cnt[t][a]++] = {(int)g, (int)(p - groupPtr[g])};
        }
    }, 1);
  }
  const double tfa = hostNowMs();
  struct Key { int allele, fp; };
  std::vector<Key> keys(A);
  parallelForDynamic((size_t)A, 32, [&](size_t a) {
    R.al[a].ec = -1;
    int fp = -1;
    if (!inAllele[a].empty()) {
      fp = 0;
      for (auto &gs : inAllele[a]) fp = (int)(((uint32_t)fp * (uint32_t)G + (uint32_t)gs.first) % 1000003u);  // uint32 wrap-around is part of the order
    }
    keys[a] = Key{(int)a, fp};
  });
  std::sort(keys.begin(), keys.end(), [](const Key &x, const Key &y) { return x.fp != y.fp ? y.fp < x.fp : x.allele < y.allele; });
  const double tfb = hostNowMs();
  ecAlleles.clear();
  auto sameGroups = [&](int a, int b) {
    if (inAllele[a].size() != inAllele[b].size()) return false;
    for (size_t i = 0; i < inAllele[a].size(); ++i)
      if (inAllele[a][i].first != inAllele[b][i].first) return false;
    return true;
  };
  // which earlier allele of the same fingerprint an allele joins (the nearest one with the same group list, Genotyper.hpp:1100-1125) does
  // not depend on the class numbers: the list comparisons -- one walk over every allele's list, 24 M entries at 10 M pairs -- run on the
  // host threads, the numbering below stays sequential
  int nKeyed = 0;
  while (nKeyed < A && keys[nKeyed].fp != -1) ++nKeyed;
  std::vector<int> joinTo((size_t)nKeyed, -1);
  parallelForDynamic((size_t)nKeyed, 32, [&](size_t i) {
    for (int j = (int)i - 1; j >= 0 && keys[j].fp == keys[i].fp; --j)
      if (sameGroups(keys[i].allele, keys[j].allele)) { joinTo[i] = j; break; }
  });
  for (int i = 0; i < nKeyed; ++i) {
    const int joined = joinTo[i] < 0 ? -1 : R.al[keys[joinTo[i]].allele].ec;
    if (joined < 0) { R.al[keys[i].allele].ec = (int)ecAlleles.size(); ecAlleles.push_back({keys[i].allele}); }
    else { R.al[keys[i].allele].ec = joined; ecAlleles[joined].push_back(keys[i].allele); }
  }
  // RemoveLowMAPQAlleleInEquivalentClass (1330-1368) keeps everything: all assignment qualities are 1 and class members
  // share their group lists.
  const double tf1 = hostNowMs();
  for (int a = 0; a < A; ++a) R.al[a].missingCov = missing[a];  // GetSeqMissingBaseCoverage, computed on the device
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k host] finalize: group lists %.1f ms, fingerprints + sort %.1f ms, classes %.1f ms, rest %.1f ms\n", tfa - tf0, tfb - tfa, tf1 - tfb, hostNowMs() - tf1);
}

void Genotyper::setAbundance(const double *n, const std::vector<int> &ecLen) {
  RefSet &R = *ref;
  if (n) {
    for (auto &a : R.al) a.abundance = a.ecAbundance = 0;
    for (size_t e = 0; e < ecAlleles.size(); ++e) {
      double fpk = 0;
      fpk += n[e];
      fpk = fpk / ecLen[e] * 1000.0;
      const int size = (int)ecAlleles[e].size();
      for (int a : ecAlleles[e]) { R.al[a].abundance = fpk / size; R.al[a].ecAbundance = fpk; }
    }
  }
  geneAbund.assign(R.geneName.size(), 0);
  majorAbund.assign(R.majorName.size(), 0);
  geneMaxMajor.assign(R.geneName.size(), 0);
  for (auto &a : R.al) { majorAbund[a.major] += a.abundance; geneAbund[a.gene] += a.abundance; }
  for (auto &a : R.al)
    if (majorAbund[a.major] > geneMaxMajor[a.gene]) geneMaxMajor[a.gene] = majorAbund[a.major];
}

// ------------------------------------------------------------------------------------------------------------------
// QuantifyAlleleEquivalentClass (Genotyper.hpp:1142-1328): SQUAREM-accelerated EM; every EMupdate is t1k_em_update
// ------------------------------------------------------------------------------------------------------------------
extern "C" void t1k_em_times(const t1k_ctx *ctx, double *ms4);  // (t1k_em.hip: debug clocks of the updates)
int Genotyper::quantify(t1k_ctx *ctx, t1k_comm *comm, std::string &err) {
  const double tq0 = hostNowMs();
  RefSet &R = *ref;
  const size_t E = ecAlleles.size(), G = nGroups();
  const size_t gBegin = 0;
  // rows of the E-step: per read group its count (the largest weight of the row, 1155-1164) and its distinct classes in
  // first-appearance order (1165-1189).  Groups are independent: the host threads take contiguous pieces (a piece's rows keep their order).
  std::vector<uint64_t> rowPtr(G + 1, 0);
  std::vector<uint32_t, NoInitAlloc<uint32_t>> ecIdx;  // (95 MB at 10 M pairs: filled by the threads below, not zeroed by one first)
  std::vector<double> count(G);
  {
    unsigned T = std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if (T < 1 || G < 8192) T = 1;
    const size_t piece = (G + T - 1) / T;
    std::vector<std::vector<uint32_t>> part(T);
    std::vector<uint32_t> ecOf(R.al.size());  // (4 bytes an allele instead of a walk through the allele records: the table stays in cache)
    for (size_t a = 0; a < R.al.size(); ++a) ecOf[a] = (uint32_t)R.al[a].ec;
    parallelFor(T, [&](size_t t) {
      const size_t g0 = t * piece, g1 = std::min(G, g0 + piece);
      std::vector<int> seen(E, 0);
      std::vector<uint32_t> &out = part[t];
      if (g1 > g0) out.reserve((size_t)((groupPtr[g1] - groupPtr[g0]) / 2 + 16));
      for (size_t gl = g0; gl < g1; ++gl) {
        const size_t g = gBegin + gl;
        float c = groupEnt[groupPtr[g]].weight;
        for (uint64_t p = groupPtr[g] + 1; p < groupPtr[g + 1]; ++p)
          if (groupEnt[p].weight > c) c = groupEnt[p].weight;
        count[gl] = c;
        const size_t before = out.size();
        for (uint64_t p = groupPtr[g]; p < groupPtr[g + 1]; ++p) {
          const uint32_t ec = ecOf[groupEnt[p].allele];
          if (seen[ec] == (int)(gl - g0) + 1) continue;
          seen[ec] = (int)(gl - g0) + 1;
          out.push_back(ec);
        }
        rowPtr[gl + 1] = out.size() - before;  // length for now
      }
    }, 1);
    for (size_t gl = 0; gl < G; ++gl) rowPtr[gl + 1] += rowPtr[gl];
    ecIdx.resize(rowPtr[G]);
    parallelFor(T, [&](size_t t) {
      const size_t g0 = t * piece;
      if (g0 < G && !part[t].empty()) memcpy(ecIdx.data() + rowPtr[g0], part[t].data(), part[t].size() * 4);
    }, 1);
  }
  std::vector<int> ecLen(E);
  for (size_t e = 0; e < E; ++e) {
    int len = R.al[ecAlleles[e][0]].effLen;
    for (int a : ecAlleles[e]) len = std::min(len, R.al[a].effLen);
    ecLen[e] = len;
  }
  const double tq1 = hostNowMs();
  if (t1k_em_setup(ctx, rowPtr.data(), ecIdx.data(), count.data(), ecLen.data(), (uint32_t)G, (uint32_t)E, nullptr, nullptr) != T1K_OK) {
    err = t1k_last_error(ctx);
    return -1;
  }
  if (comm && t1k_comm_size(comm) > 1) {  // this rank's slice of the read groups for the row pass of every EMupdate
    const uint64_t n = (uint64_t)t1k_comm_size(comm), r = (uint64_t)t1k_comm_rank(comm);
    if (t1k_em_shard(ctx, (uint32_t)(G * r / n), (uint32_t)(G * (r + 1) / n), comm) != T1K_OK) { err = t1k_last_error(ctx); return -1; }
  }
  const double tq2 = hostNowMs();
  std::vector<double> x0(E), x1(E), x2(E), x3(E), n(E);
  for (size_t e = 0; e < E; ++e) {
    x0[e] = 0;
    for (int a : ecAlleles[e]) x0[e] += R.al[a].weight;
  }
  const int maxIterations = 1000, maskEvery = 10;
  int rounds = 0;
  double diff = 0;
  auto em = [&](std::vector<double> &from, std::vector<double> &to) {
    if (E == 0) return true;
    if (t1k_em_update(ctx, from.data(), to.data(), n.data(), &diff) != T1K_OK) { err = t1k_last_error(ctx); return false; }
    return true;
  };
  for (int t = 0; t < maxIterations; ++t) {
    ++rounds;
    if (!em(x0, x1) || !em(x1, x2)) return -1;
    double r2 = 0, v2 = 0;  // SQUAREMalpha (424-437)
    for (size_t e = 0; e < E; ++e) {
      r2 += (x1[e] - x0[e]) * (x1[e] - x0[e]);
      v2 += (x2[e] - 2 * x1[e] + x0[e]) * (x2[e] - 2 * x1[e] + x0[e]);
    }
    double alpha = v2 == 0 ? -1 : -sqrt(r2) / sqrt(v2);
    if (prm.squarem_min_alpha < 0 && alpha < prm.squarem_min_alpha) alpha = prm.squarem_min_alpha;
    for (size_t e = 0; e < E; ++e) x3[e] = x0[e] - 2 * alpha * (x1[e] - x0[e]) + alpha * alpha * (x2[e] - 2 * x1[e] + x0[e]);
    if (!em(x3, x1)) return -1;
    double moved = 0;
    for (size_t e = 0; e < E; ++e) { moved += fabs(x1[e] - x0[e]); x0[e] = x1[e]; }
    if (moved < 1e-5 && t < maxIterations - 2) t = maxIterations - 2;  // one forced extra round
    if (t > 0 && t % maskEvery == 0) {
      setAbundance(n.data(), ecLen);
      for (auto &a : R.al)
        if (majorAbund[a.major] < prm.filter_frac * 0.5 * geneMaxMajor[a.gene]) { a.abundance = 0; a.ecAbundance = 0; }
      for (size_t e = 0; e < E; ++e) x0[e] = R.al[ecAlleles[e][0]].ecAbundance;
    }
  }
  setAbundance(n.data(), ecLen);
  emIterations = rounds;
  if (getenv("T1K_DEBUG_PHASES")) {
    double u[4];
    t1k_em_times(ctx, u);
    fprintf(stderr, "[t1k host] quantify: rows %.1f ms, setup %.1f ms (nnz %zu), iterations %.1f ms (%.0f updates: staging + enqueue %.1f ms, waiting for the device %.1f ms, M-step %.1f ms)\n",
            tq1 - tq0, tq2 - tq1, ecIdx.size(), hostNowMs() - tq2, u[3], u[0], u[1], u[2]);
  }
  return rounds;
}

// ------------------------------------------------------------------------------------------------------------------
// RemoveLowLikelihoodAlleleInEquivalentClass (Genotyper.hpp:1371-1460)
// ------------------------------------------------------------------------------------------------------------------
void Genotyper::dropUnlikely() {
  RefSet &R = *ref;
  // The reference walks the groups of the class representative and picks out the members' entries (1398-1416).  Class members share
  // their group list by construction, so those entries are exactly each member's own entries: the covered span of EVERY allele (smallest
  // start, largest end over its entries) comes out of one pass over the entry table in storage order -- the host threads take contiguous
  // pieces and their minima / maxima are combined -- instead of 24 M scattered reads through the per-allele lists.
  const size_t A = R.al.size(), G = nGroups();
  std::vector<int> spanLo(A), spanHi(A, -1);
  for (size_t a = 0; a < A; ++a) spanLo[a] = R.al[a].seqLen;
  {
    unsigned T = std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if (T < 1 || G < 8192) T = 1;
    const uint64_t nEnt = G ? groupPtr[G] : 0, piece = (nEnt + T - 1) / T;
    std::vector<std::vector<int>> lo(T), hi(T);
    parallelFor(T, [&](size_t t) {
      const uint64_t p0 = t * piece, p1 = std::min(nEnt, p0 + piece);
      if (p0 >= p1) return;
      lo[t] = spanLo;  // (seqLen: the reference's initial value)
      hi[t].assign(A, -1);
      for (uint64_t p = p0; p < p1; ++p) {
        const GroupEntry &e = groupEnt[p];
        if (e.start < lo[t][e.allele]) lo[t][e.allele] = e.start;
        if (e.end > hi[t][e.allele]) hi[t][e.allele] = e.end;
      }
    }, 1);
    for (unsigned t = 0; t < T; ++t) {
      if (lo[t].empty()) continue;
      for (size_t a = 0; a < A; ++a) {
        if (lo[t][a] < spanLo[a]) spanLo[a] = lo[t][a];
        if (hi[t][a] > spanHi[a]) spanHi[a] = hi[t][a];
      }
    }
  }
  parallelFor(ecAlleles.size(), [&](size_t ci) {
    std::vector<int> &members = ecAlleles[ci];
    const int size = (int)members.size();
    std::vector<int> lo(size), hi(size, -1);
    for (int j = 0; j < size; ++j) { lo[j] = spanLo[members[j]]; hi[j] = spanHi[members[j]]; }
    std::vector<double> ll(size);
    double best = -1;
    for (int j = 0; j < size; ++j) {
      const int len = R.al[members[j]].seqLen;
      int span = hi[j] - lo[j] + 1;
      if (span > len) span = len;
      ll[j] = pow(double(span) / len, R.al[members[j]].ecAbundance);
      if (ll[j] > best) best = ll[j];
    }
    std::vector<int> kept;
    for (int j = 0; j < size; ++j)
      if (ll[j] / best >= 0.05 || ll[j] == best) kept.push_back(members[j]);
    members = kept;
  });
}

int Genotyper::geneTypes(int gene) const {
  if (selected[gene].empty()) return 0;
  int top = 0;
  for (auto &s : selected[gene]) top = std::max(top, s.second);
  return top + 1;
}

// upper/lower tail of the standard normal, Hill's algorithm AS 66 (Applied Statistics 22(3), 1973) as used by
// Genotyper.hpp:252-370
static double normalTail(double x, bool upper) {
  const double ltone = 7.0, utzero = 18.66, con = 1.28;
  bool up = upper;
  double z = x;
  if (z < 0) { up = !up; z = -z; }
  if (ltone < z && (!up || utzero < z)) return up ? 0.0 : 1.0;
  double y = 0.5 * z * z, v;
  if (z <= con)
    v = 0.5 - z * (0.398942280444 - 0.39990348504 * y / (y + 5.75885480458 + -29.8213557807 / (y + 2.62433121679 + 48.6959930692 / (y + 5.92885724438))));
  else
    v = 0.398942280385 * exp(-y) /
        (z + -0.000000038052 + 1.00000615302 / (z + 0.000398064794 + 1.98615381364 / (z + -0.151679116635 + 5.29330324926 / (z + 4.8385912808 + -15.1508972451 / (z + 0.742380924027 + 30.789933034 / (z + 3.99019417011))))));
  return up ? v : 1.0 - v;
}

// ------------------------------------------------------------------------------------------------------------------
// SelectAllelesForGenes (Genotyper.hpp:1462-2090)
// ------------------------------------------------------------------------------------------------------------------
void Genotyper::select() {
  RefSet &R = *ref;
  const int G = (int)nGroups(), nGenes = (int)R.geneName.size(), E = (int)ecAlleles.size();
  const double frac = prm.filter_frac;
  std::vector<char> groupCovered(G, 0);
  selected.assign(nGenes, {});
  hookFailed = false;
  const double ts0 = hostNowMs();
  // weight of a group's first entry, gathered once: the class loop below reads it for every group of every class it looks at, and the
  // entries themselves are hundreds of megabytes
  std::vector<double> firstW(G);
  parallelFor((size_t)G, [&](size_t g) { firstW[g] = (double)groupEnt[groupPtr[g]].weight; }, 4096);
  auto firstWeight = [&](int g) { return firstW[g]; };
  auto lowAbundance = [&](int a) {  // 1568-1570 == 1656-1658
    const AlleleMeta &m = R.al[a];
    return m.ecAbundance < frac * geneMaxMajor[m.gene] &&
           (m.ecAbundance * 3 >= majorAbund[m.major] || majorAbund[m.major] < 3 * frac * geneMaxMajor[m.gene]);
  };
  // classes by abundance (desc), class id (asc)
  std::vector<std::pair<int, double>> order;
  for (int e = 0; e < E; ++e) order.push_back({e, ecAlleles[e].empty() ? 0.0 : R.al[ecAlleles[e][0]].ecAbundance});
  std::sort(order.begin(), order.end(), [](const std::pair<int, double> &x, const std::pair<int, double> &y) { return x.second != y.second ? y.second < x.second : x.first < y.first; });
  std::vector<int> rejected;
  for (auto &oe : order) {
    const std::vector<int> &members = ecAlleles[oe.first];
    if (members.empty()) break;  // cannot happen: dropUnlikely always keeps the maximum
    const int lead = members[0];
    if (R.al[lead].ecAbundance <= 1e-6) break;
    double covered = 0, total = 0;
    for (auto &gs : inAllele[lead]) {
      double w = firstWeight(gs.first);
      if (groupCovered[gs.first]) covered += w;
      total += w;
    }
    std::vector<int> genesHere, toAdd;
    for (int a : members) {
      const AlleleMeta &m = R.al[a];
      bool reject = lowAbundance(a);
      if (covered == total &&
          (m.ecAbundance < 0.25 * geneMaxMajor[m.gene] || selected[m.gene].empty() || m.ecAbundance < 0.5 * R.al[selected[m.gene].back().first].ecAbundance))
        reject = true;
      if (reject) { rejected.push_back(a); continue; }
      if (std::find(genesHere.begin(), genesHere.end(), m.gene) == genesHere.end()) genesHere.push_back(m.gene);
      toAdd.push_back(a);
    }
    const int quality = genesHere.size() > 1 ? 0 : 60;
    if (!genesHere.empty())
      for (auto &gs : inAllele[lead]) groupCovered[gs.first] = 1;
    std::map<int, int> freshRank;
    for (int a : toAdd) {
      AlleleMeta &m = R.al[a];
      int rank = -1;
      for (auto &s : selected[m.gene])
        if (R.al[s.first].major == m.major) { rank = s.second; break; }
      if (rank == -1) {
        auto it = freshRank.find(m.gene);
        if (it != freshRank.end()) rank = it->second;
        else { rank = geneTypes(m.gene); freshRank[m.gene] = rank; }
      }
      m.quality = quality;
      m.rank = rank;
      if (lowAbundance(a)) m.quality = 0;
      selected[m.gene].push_back({a, rank});
    }
  }
  const double ts1 = hostNowMs();
  // rescue rejected alleles whose major allele did get selected (1669-1695)
  for (int a : rejected) {
    const AlleleMeta &m = R.al[a];
    int rank = -1;
    for (auto &s : selected[m.gene])
      if (R.al[s.first].major == m.major) { rank = s.second; break; }
    if (rank != -1) selected[m.gene].push_back({a, rank});
  }
  if (missingCoverageHook) {
    std::vector<int> need;
    for (int g = 0; g < nGenes; ++g)
      if (geneTypes(g) > 2)  // (the type-pair search below skips the other genes, and nothing else reads the value)
        for (auto &sa : selected[g]) need.push_back(sa.first);
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    hookFailed = !missingCoverageHook(need);
    if (hookFailed) return;
  }
  const double ts2 = hostNowMs();
  // genes with more than two allele types: pick the pair of types explaining the most reads (1697-1996)
  std::vector<int> groupUse(G, 0);
  auto forTopTwo = [&](int gene, std::set<int> &usedEc, int delta) {
    for (auto &s : selected[gene]) {
      if (s.second > 1) continue;
      if (!usedEc.insert(R.al[s.first].ec).second) continue;
      for (auto &gs : inAllele[s.first]) groupUse[gs.first] += delta;
    }
  };
  {
    std::set<int> usedEc;  // shared across genes here (1705-1729)
    for (int g = 0; g < nGenes; ++g) forTopTwo(g, usedEc, +1);
  }
  std::vector<std::map<int, double>> typeWeight(nGenes);  // missing coverage -> abundance of the best type carrying it (1733-1770)
  for (int g = 0; g < nGenes; ++g) {
    const int T = geneTypes(g);
    std::vector<int> miss(T, -1);
    std::vector<double> ab(T, 0);
    for (auto &s : selected[g]) {
      ab[s.second] += R.al[s.first].abundance;
      if (miss[s.second] == -1 || R.al[s.first].missingCov < miss[s.second]) miss[s.second] = R.al[s.first].missingCov;
    }
    for (int t = 0; t < T; ++t) {
      auto it = typeWeight[g].find(miss[t]);
      if (it == typeWeight[g].end() || it->second < ab[t]) typeWeight[g][miss[t]] = ab[t];
    }
  }
  // The reference scores a pair (j, k) by building a std::map over the read groups of the two types' alleles that no other gene's
  // top two use, and adding up the groups' adjusted weights in key order.  Here: which alleles of k enter is settled first (the
  // equivalence-class set that survives from one k to the next, SURVEY H21, is the only thing that ties the pairs of a gene together),
  // then every pair is scored on its own -- a bitmap over the groups, walked in ascending order (the map's order: the same sum, bit for
  // bit) -- by the host threads when there are enough list entries to share out.
  std::vector<double> adjW(G);
  parallelFor((size_t)G, [&](size_t g) { adjW[g] = groupEnt[groupPtr[g]].adjustWeight; }, 4096);
  const size_t BW = ((size_t)G + 63) / 64;
  struct PairJob { int j, k, lastJ; const std::vector<int> *lj; std::vector<int> lk; double score, product; };
  for (int iter = 0; iter < 1000; ++iter) {
    int changed = 0;
    for (int g = 0; g < nGenes; ++g) {
      const int T = geneTypes(g);
      if (T <= 2) continue;
      std::vector<std::pair<int, int>> &sel = selected[g];
      const int S = (int)sel.size();
      std::set<int> usedEc;
      forTopTwo(g, usedEc, -1);
      std::vector<PairJob> jobs;
      std::vector<int> listJ[2];
      int lastJ = 0;
      size_t entries = 0;
      for (int j = 0; j < T - 1 && j <= 1; ++j) {
        usedEc.clear();
        size_t entJ = 0;
        for (int l = 0; l < S; ++l) {
          if (sel[l].second != j) continue;
          if (!usedEc.insert(R.al[sel[l].first].ec).second) continue;
          listJ[j].push_back(sel[l].first);
          entJ += inAllele[sel[l].first].size();
          lastJ = l;
        }
        for (int k = j + 1; k < T; ++k) {
          PairJob pj{j, k, lastJ, &listJ[j], {}, 0.0, 0.0};
          entries += entJ;
          for (int l = 0; l < S; ++l) {  // usedEc deliberately survives from one k to the next
            if (sel[l].second != k) continue;
            if (!usedEc.insert(R.al[sel[l].first].ec).second) continue;
            pj.lk.push_back(sel[l].first);
            entries += inAllele[sel[l].first].size();
          }
          jobs.push_back(std::move(pj));
        }
      }
      const std::map<int, double> &tw = typeWeight[g];
      auto scoreJob = [&](size_t q) {
        PairJob &pj = jobs[q];
        std::vector<uint64_t> bits(BW, 0);
        for (const std::vector<int> *lst : {pj.lj, (const std::vector<int> *)&pj.lk})
          for (int a : *lst)
            for (auto &gs : inAllele[a])
              if (groupUse[gs.first] == 0) bits[(size_t)gs.first >> 6] |= 1ull << (gs.first & 63);
        double abJ = 0, abK = 0;
        int missJ = -1, missK = -1;
        for (int l = 0; l < S; ++l) {
          const AlleleMeta &m = R.al[sel[l].first];
          if (sel[l].second == pj.j) { abJ += m.abundance; if (missJ == -1 || m.missingCov < missJ) missJ = m.missingCov; }
          else if (sel[l].second == pj.k) { abK += m.abundance; if (missK == -1 || m.missingCov < missK) missK = m.missingCov; }
        }
        double score = 0;
        for (size_t w = 0; w < BW; ++w) {
          uint64_t x = bits[w];
          while (x) { score += adjW[(w << 6) + (size_t)__builtin_ctzll(x)]; x &= x - 1; }
        }
        if (T > 3 || missJ >= 10 || missK >= 10) {
          auto itJ = tw.find(missJ), itK = tw.find(missK);  // (every type's smallest missing coverage is a key: typeWeight was built from the same lists)
          double wJ = itJ != tw.end() ? itJ->second : 0.0, wK = itK != tw.end() ? itK->second : 0.0;
          if (T <= 3) {
            if (wJ >= 1) wJ = log(wJ) / log(10.0);
            if (wK >= 1) wK = log(wK) / log(10.0);
          }
          score = score - missJ * wJ * readLength / 150.0 - missK * wK * readLength / 150.0 + (R.al[sel[pj.lastJ].first].weight);
        }
        pj.score = score; pj.product = abJ * abK;
      };
      if (entries + jobs.size() * BW > (size_t)4 << 20) parallelForDynamic(jobs.size(), 1, scoreJob);
      else for (size_t q = 0; q < jobs.size(); ++q) scoreJob(q);
      double bestCover = 0, bestProduct = 0;
      std::vector<std::pair<int, int>> bestPairs;
      for (const PairJob &pj : jobs) {
        if (bestPairs.empty() || pj.score > bestCover || (pj.score == bestCover && pj.product > bestProduct)) {
          bestCover = pj.score; bestProduct = pj.product;
          bestPairs.clear();
          bestPairs.push_back({pj.j, pj.k});
        } else if (pj.score == bestCover) bestPairs.push_back({pj.j, pj.k});
      }
      const std::pair<int, int> win = bestPairs[0];
      if (win.first != 0 || win.second != 1) {
        ++changed;
        for (auto &s : sel) {
          int r;
          if (s.second == win.first) r = 0;
          else if (s.second == win.second) r = 1;
          else if (s.second < win.first) r = s.second + 2;
          else if (s.second < win.second) r = s.second + 1;
          else continue;
          s.second = r;
          R.al[s.first].rank = r;
        }
      }
      usedEc.clear();
      forTopTwo(g, usedEc, +1);
    }
    if (!changed) break;
  }
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k host] select: classes by abundance %.1f ms (%zu rejected), rescue %.1f ms, type pairs %.1f ms\n", ts1 - ts0, rejected.size(), ts2 - ts1, hostNowMs() - ts2);
  // genotype quality (2010-2085)
  std::vector<double> selAbund(nGenes, 0);
  for (int g = 0; g < nGenes; ++g)
    for (auto &s : selected[g]) selAbund[g] += R.al[s.first].abundance;
  const double crossAllele = 0.01;
  for (int g = 0; g < nGenes; ++g) {
    const int T = geneTypes(g);
    std::vector<double> ab(T, 0);
    for (auto &s : selected[g]) ab[s.second] += R.al[s.first].abundance;
    double noise = 0;
    for (int o = 0; o < nGenes; ++o)
      if (o != g) noise += prm.cross_gene_rate * R.geneSim[o][g] * selAbund[o];
    for (int t = 0; t < T; ++t) {
      double nullMean = (selAbund[g] - ab[t]) * crossAllele + noise;
      double score = 0;
      if (ab[t]) score = -log(normalTail(2 * (sqrt(ab[t]) - sqrt(nullMean)), true)) / log(double(10.0));
      if (score > 60) score = 60;
      if (score < 0) score = 0;
      if (ab[t] < prm.filter_cov) score = 0;
      for (auto &s : selected[g])
        if (s.second == t && R.al[s.first].quality > 0) R.al[s.first].quality = (int)score;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// text output (Genotyper::GetAlleleDescription 2103-2178 + Genotyper.cpp:660-670; OutputRepresentativeAlleles 2180-2229)
// ------------------------------------------------------------------------------------------------------------------
std::string Genotyper::geneLine(int gene) const {
  const RefSet &R = *ref;
  std::vector<char> shown(R.majorName.size(), 0);
  int firstTwoQual[2] = {-1, -1};
  int called = 0;
  int T = std::max(2, geneTypes(gene));
  std::string field[3];
  char sep = '\t';
  char num[64];
  for (int type = 0; type < T; ++type) {
    std::string &buf = field[type < 2 ? type : 2];
    if (type > 1) sep = ';';
    buf.clear();  // the reference resets the buffer for every type, so only the last secondary type survives
    double abundance = 0;
    bool any = false;
    int qual = -1;
    if (type == 1 && firstTwoQual[0] == 0) std::fill(shown.begin(), shown.end(), 0);
    for (auto &s : selected[gene]) {
      if (s.second != type) continue;
      const AlleleMeta &m = R.al[s.first];
      abundance += m.abundance;
      if (shown[m.major]) continue;
      qual = m.quality;
      if (type <= 1) called = type + 1;
      if (any) buf += ",";
      buf += R.majorName[m.major];
      any = true;
      shown[m.major] = 1;
    }
    if (qual >= 0) { snprintf(num, sizeof(num), "%c%lf%c%d", sep, abundance, sep, qual); buf += num; }
    else if (type <= 1) buf += ".\t0\t-1";
    if (type <= 1) firstTwoQual[type] = qual;
  }
  std::string line = R.geneName[gene] + "\t" + std::to_string(called);
  for (int j = 0; j < 3; ++j) line += "\t" + field[j];
  line += "\n";
  return line;
}

std::string Genotyper::alleleLines() const {
  const RefSet &R = *ref;
  std::string out;
  for (size_t g = 0; g < R.geneName.size(); ++g) {
    int rep[2] = {-1, -1};
    for (auto &s : selected[g]) {
      const int t = s.second, a = s.first;
      if (t > 1 || R.al[a].quality < 1) continue;
      if (rep[t] == -1 || R.al[rep[t]].ecAbundance < R.al[a].ecAbundance || (R.al[rep[t]].ecAbundance == R.al[a].ecAbundance && rep[t] > a)) rep[t] = a;
    }
    if (rep[1] == -1 && rep[0] != -1) {  // two alleles of one major allele (2201-2221)
      double top = -1;
      int pick = -1;
      for (auto &s : selected[g]) {
        const int a = s.first;
        if (s.second != 0 || R.al[a].ec == R.al[rep[0]].ec) continue;
        std::string g1, m1, g2, m2;
        R.splitName(R.al[a].name, g1, m1, 1);
        R.splitName(R.al[rep[0]].name, g2, m2, 1);
        if (m1 == m2) continue;
        if (R.al[a].ecAbundance > top || (R.al[a].ecAbundance == top && a < pick)) { top = R.al[a].ecAbundance; pick = a; }
      }
      if (top != -1) rep[1] = pick;
    }
    for (int j = 0; j < 2; ++j)
      if (rep[j] != -1) out += R.al[rep[j]].name + " " + std::to_string(R.al[rep[j]].quality) + "\n";
  }
  return out;
}

}  // namespace t1k
