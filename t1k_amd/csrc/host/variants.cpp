// t1k_amd/csrc/host/variants.cpp -- novel-variant calling of the analyzer stage, on the host ("VariantCaller stays CPU").
//
// What the reference's VariantCaller (VariantCaller.hpp:92-1311) computes from the fragments' assignment lists and the edit strings of their
// read-ends (SeqSet::AddOverlapAlignmentInfo, SeqSet.hpp:2657-2681: one global alignment of the read window against the allele window per
// overlap -- in this build t1k_align_batch on the GPU), written from its behaviour:
//
//   1. per-base tables (ComputeVariant 987-1001 -> UpdateBaseVariantFromFragmentOverlap 273-305 -> UpdateBaseVariantFromOverlap 103-173):
//      a first sweep over every assignment records, per allele base and read nucleotide, the best match count seen (and adds 1 to the
//      nucleotide's count); a second sweep adds 1 again wherever the overlap is "good" for the base (match count within 4 of the best of
//      all four nucleotides, IsGoodAssignment 47-54) and keeps a "unique" count for fragments whose abundance share is exactly 1.
//      The reference's walk does NOT advance its two positions when it skips a column (the `continue`s of 134-137 leave the loop body
//      before 165-168): every later column of that overlap is booked one base early.  Kept, column for column (bookOverlap below).
//   2. root candidates (FindCandidateVariants 307-345): bases where a nucleotide other than the allele's own has count >= 5 and at least
//      half the count of the allele's own.
//   3. expansion to the other alleles a fragment is assigned to, until nothing is added (ExpandCandidateVariantsFromFragmentOverlap
//      347-571, loop 1049-1070): the read-end's assignments are walked side by side, read position by read position; where one good
//      assignment sits on a candidate base, the bases the other good assignments hold at that read position become candidates too, and
//      every ordered pair of candidates met together gets an edge weight.
//   4. groups (BuildCandidateVariantGroup 573-593): components of root candidates over the edges that carry >= 15 % of either end's depth.
//   5. fragment <-> candidate edges (BuildFragmentCandidateVarGraph 595-687): the nucleotide a fragment shows at a candidate base.
//   6. per group of at most --varMaxGroup candidates on distinct alleles with an exonic member: the nucleotide choice that covers most
//      fragments with fewest changes, by enumeration (EnumerateVariants 689-820, SolveVariantGroup 822-976).
//   7. <prefix>_allele.vcf (OutputAlleleVCF 1202-1227) and, for the per-barcode table, the assignments of a fragment that explain most
//      of its mismatches by called variants (AdjustFragmentAssignment 1229-1311).
//
// Data structures are this build's own: one flat cell table over all alleles, candidate edges in per-candidate vectors (the reference
// threads linked lists through SimpleVectors; list order never reaches a result: weights are integer counts, groups are components, a
// leaf of the enumeration only counts), the enumeration as an odometer instead of a recursion.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <unordered_map>

#include "t1k_host.h"

namespace t1k {

namespace {

enum { OP_MATCH = 0, OP_MISMATCH = 1, OP_INSERT = 2, OP_DELETE = 3 };  // AlignAlgo.hpp:7-10

inline int nucIndex(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }  // nucToNum, Analyzer.cpp:34-37
const char kNuc[4] = {'A', 'C', 'G', 'T'};

// a read as an overlap sees it: as it is, or reverse-complemented (SeqSet::ReverseComplement, SeqSet.hpp:2103-2114: what is not A/C/G/T stays N)
struct ReadView {
  const char *p;
  uint32_t n;
  bool rev;
  char at(int i) const {
    if (!rev) return p[i];
    const int b = nucIndex(p[n - 1 - (uint32_t)i]);
    return b >= 0 ? kNuc[3 - b] : 'N';
  }
};

}  // namespace

struct VariantCaller::Cell {
  double count[4] = {0, 0, 0, 0};  // _baseVariant::count; unweightedCount (every statement that adds to one adds the same to the other: 142-143)
  double uniq[4] = {0, 0, 0, 0};   // uniqCount
  int bestMatch[4] = {0, 0, 0, 0}; // alignInfo[].a (the similarity beside it, alignInfo[].b, is written and never read)
  int bestOfAll = 0;               // the largest of the four
  int cand = -1;                   // candidateId
  double depth() const { return count[0] + count[1] + count[2] + count[3]; }
  bool good(int matchCnt) const { return matchCnt >= bestOfAll - 4; }  // IsGoodAssignment (47-54): not more than 4 below any nucleotide's best
};

VariantCaller::~VariantCaller() = default;

VariantCaller::VariantCaller(const RefSet &ref, const std::vector<double> &abundance, int maxGroup) : ref_(ref), abundance_(abundance), maxGroup_(maxGroup) {
  const size_t A = ref.seqs.size();
  base_.assign(A + 1, 0);
  for (size_t a = 0; a < A; ++a) base_[a + 1] = base_[a] + ref.seqs[a].size();
  cells_.reset(new std::vector<Cell>(base_[A]));
  // seqCopy (SetSeqAbundance 260-264): how many of the loaded alleles belong to the allele's gene
  std::map<int, int> perGene;
  for (size_t a = 0; a < A; ++a) ++perGene[ref.al[a].gene];
  copies_.resize(A);
  for (size_t a = 0; a < A; ++a) copies_[a] = perGene[ref.al[a].gene];
  candAt_.assign(A, {});
}

VariantCaller::Cell &VariantCaller::cell(int allele, int pos) const { return (*cells_)[base_[allele] + (size_t)pos]; }

// the overlap a sweep over read-end k of a fragment reads (SelectOverlapFromFragmentOverlap 201-208)
static inline const t1k_overlap &endOverlap(const t1k_frag_assignment &a, int k) { return k == 1 ? a.o2 : a.o1; }
static inline void endOps(const t1k_frag_assignment &a, int k, const int8_t *ops, const int8_t *&p, uint32_t &n) {
  p = ops + (k == 1 ? a.ops2 : a.ops1);
  n = k == 1 ? a.n_ops2 : a.n_ops1;
}

// UpdateBaseVariantFromOverlap (103-173).  filter = false: the sweep that only learns the best match counts (updateType 1: weight 0)
// table: where the counts are added; best: where good() reads the best match counts (both the job's cells, or -- the sweeps threaded over the
// FRAGMENTS -- a thread's private table and the job's cells, whose best match counts the first sweep has made final)
void VariantCaller::bookOverlap(const char *read, uint32_t len, const t1k_overlap &o, const int8_t *ops, uint32_t nOps, double weight, bool filter, Cell *table, const Cell *best) {
  if (o.seq_idx == -1) return;
  const ReadView r{read, len, o.strand == -1};
  const int L = (int)ref_.seqs[o.seq_idx].size();
  Cell *cells = table + base_[o.seq_idx];
  const Cell *bestCells = best + base_[o.seq_idx];
  int refPos = o.seq_start, readPos = o.read_start;
  for (uint32_t k = 0; k < nOps; ++k) {
    const int op = ops[k];
    if (op == OP_MATCH || op == OP_MISMATCH) {
      if (refPos >= L || readPos >= (int)len) break;  // (cannot happen with the edit string of this overlap; the reference does not look)
      Cell &c = cells[refPos];
      if (filter && !bestCells[refPos].good(o.match_cnt)) continue;  // 134-135: leaves the column WITHOUT moving on (see the header)
      const int b = nucIndex(r.at(readPos));
      if (b < 0) continue;                           // 136-137 ('N'): the same
      if (weight == 1) c.uniq[b] += weight;
      c.count[b] += 1;
      if (o.match_cnt > c.bestMatch[b]) { c.bestMatch[b] = o.match_cnt; if (o.match_cnt > c.bestOfAll) c.bestOfAll = o.match_cnt; }
    }
    if (op != OP_INSERT) ++refPos;
    if (op != OP_DELETE) ++readPos;
  }
}

// UpdateBaseVariantFromFragmentOverlap (273-305).  An overlap books into the cells of its own allele only, and what it reads there (good())
// was written by overlaps on the same allele: thread `part` of `parts` takes the assignments of the alleles it owns, in fragment order, and
// every allele's cells see exactly the sequence of updates of a single sweep.
void VariantCaller::bookFragment(const Fragment &f, const int8_t *ops, bool first, int part, int parts, Cell *table, const Cell *best) {
  double total = 0;
  for (uint32_t i = 0; i < f.n; ++i) total += abundance_[f.asg[i].allele_idx];
  for (uint32_t i = 0; i < f.n; ++i) {
    const t1k_frag_assignment &a = f.asg[i];
    if (a.allele_idx % parts != part) continue;
    const double w = first ? 0.0 : abundance_[a.allele_idx] / total;
    if (a.has_mate_pair) {
      bookOverlap(f.r1, f.l1, a.o1, ops + a.ops1, a.n_ops1, w, !first, table, best);
      bookOverlap(f.r2, f.l2, a.o2, ops + a.ops2, a.n_ops2, w, !first, table, best);
    } else if (!a.o1_from_r2) bookOverlap(f.r1, f.l1, a.o1, ops + a.ops1, a.n_ops1, w, !first, table, best);
    else bookOverlap(f.r2, f.l2, a.o1, ops + a.ops1, a.n_ops1, w, !first, table, best);
  }
}

// candidates of an allele inside [from, to]?  (what the reference's accumulated counts were meant for, 182-199, 214-226)
bool VariantCaller::candidateIn(int allele, int from, int to) const {
  const std::vector<int> &v = candAt_[allele];
  auto it = std::lower_bound(v.begin(), v.end(), from);
  return it != v.end() && *it <= to;
}

int VariantCaller::newCandidate(int allele, int pos, bool root) {
  const int id = (int)cand_.size();
  cand_.push_back({allele, pos});
  root_.push_back(root ? 1 : 0);
  group_.push_back(-1);
  cell(allele, pos).cand = id;
  std::vector<int> &v = candAt_[allele];
  v.insert(std::upper_bound(v.begin(), v.end(), pos), pos);
  return id;
}

// FindCandidateVariants (307-345).  An allele base that is not A/C/G/T (the N separators of a genomic reference) reads count[-1] there:
// the eight bytes in front of the array, which are the zeroed tail of the previous table entry -- an own count of 0, and no nucleotide
// is "the allele's own".
void VariantCaller::findRoots() {
  for (size_t a = 0; a < ref_.seqs.size(); ++a) {
    const std::string &s = ref_.seqs[a];
    for (size_t j = 0; j < s.size(); ++j) {
      Cell &c = cell((int)a, (int)j);
      const int own = nucIndex(s[j]);
      const double ownCount = own >= 0 ? c.count[own] : 0.0;
      for (int k = 0; k < 4; ++k)
        if (c.count[k] >= 5 && c.count[k] >= ownCount * 0.5 && k != own) { newCandidate((int)a, (int)j, true); break; }
    }
  }
}

// ExpandCandidateVariantsFromFragmentOverlap (347-571) for one fragment.  `edges_[c]` = (other candidate, times met together).
void VariantCaller::expandFragment(const Fragment &f, const int8_t *ops) {
  const uint32_t n = f.n;
  if (!n) return;
  std::vector<int> refPos(n), readPos(n);
  std::vector<uint32_t> at(n);
  std::vector<char> valid(n);
  for (int k = 0; k <= 1; ++k) {
    if (k == 1 && !f.asg[0].has_mate_pair) break;
    // (362-371: the test for candidates under the overlaps ends in `;` and its loop always leaves at the first assignment -- no fragment is skipped)
    const uint32_t len = (k == 1 || f.asg[0].o1_from_r2) ? f.l2 : f.l1;  // 376-380: the read of the FIRST assignment decides for all
    for (uint32_t i = 0; i < n; ++i) {
      const t1k_overlap &o = endOverlap(f.asg[i], k);
      refPos[i] = o.seq_start;
      readPos[i] = o.read_start;
    }
    bool sameStart = true;
    for (uint32_t i = 1; i < n; ++i)
      if (readPos[i] != readPos[0]) { sameStart = false; break; }
    if (!sameStart) continue;  // 398-404
    // The walk below stands on the bases [seq_start, seq_end + 1] of every assignment and does something only where one of them is a
    // candidate: a read-end none of whose windows holds a candidate (as of now -- candidates are added while the fragments are swept) is done.
    bool any = false;
    for (uint32_t i = 0; i < n && !any; ++i) {
      const t1k_overlap &o = endOverlap(f.asg[i], k);
      any = candidateIn(o.seq_idx, o.seq_start, o.seq_end + 1);
    }
    if (!any) continue;
    std::fill(at.begin(), at.end(), 0u);
    for (uint32_t j = 0; j < len; ++j) {  // the read position is the anchor (407)
      bool onCandidate = false;
      for (uint32_t i = 0; i < n; ++i) {
        const t1k_overlap &o = endOverlap(f.asg[i], k);
        valid[i] = refPos[i] < (int)ref_.seqs[o.seq_idx].size() && cell(o.seq_idx, refPos[i]).good(o.match_cnt);
      }
      for (uint32_t i = 0; i < n && !onCandidate; ++i)
        if (valid[i] && cell(endOverlap(f.asg[i], k).seq_idx, refPos[i]).cand != -1) onCandidate = true;
      if (onCandidate) {
        for (uint32_t i = 0; i < n; ++i) {
          if (!valid[i]) continue;
          const t1k_overlap &o = endOverlap(f.asg[i], k);
          const int8_t *e; uint32_t ne;
          endOps(f.asg[i], k, ops, e, ne);
          Cell &c = cell(o.seq_idx, refPos[i]);
          if (c.cand == -1 && at[i] < ne && (e[at[i]] == OP_MATCH || e[at[i]] == OP_MISMATCH)) {
            newCandidate(o.seq_idx, refPos[i], false);
            edges_.emplace_back();
          }
        }
        for (uint32_t i = 0; i < n; ++i) {  // 507-548: every ordered pair of candidates held at this read position
          if (!valid[i]) continue;
          const int ci = cell(endOverlap(f.asg[i], k).seq_idx, refPos[i]).cand;
          if (ci == -1) continue;
          for (uint32_t l = 0; l < n; ++l) {
            if (l == i || !valid[l]) continue;
            const int cl = cell(endOverlap(f.asg[l], k).seq_idx, refPos[l]).cand;
            if (cl == -1) continue;
            std::vector<std::pair<int, double>> &ev = edges_[ci];
            size_t q = 0;
            while (q < ev.size() && ev[q].first != cl) ++q;
            if (q < ev.size()) ev[q].second += 1;
            else ev.emplace_back(cl, 1.0);
          }
        }
      }
      for (uint32_t i = 0; i < n; ++i) {  // 551-567: on to the next read position
        const int8_t *e; uint32_t ne;
        endOps(f.asg[i], k, ops, e, ne);
        while (at[i] < ne && readPos[i] <= (int)j) {
          if (e[at[i]] != OP_INSERT) ++refPos[i];
          if (e[at[i]] != OP_DELETE) ++readPos[i];
          ++at[i];
        }
      }
    }
  }
}

// BuildFragmentCandidateVarGraph (595-687): seen_[c] = (fragment, nucleotide) pairs, each once.  Fragments arrive in order, so the pairs
// of the current fragment are the tail of the vector.  A candidate's list is written by the assignments on its allele only: thread `part` of
// `parts` takes the alleles it owns.
void VariantCaller::linkFragment(const Fragment &f, uint32_t fragIdx, const int8_t *ops, int part, int parts) {
  if (!f.n) return;
  for (int k = 0; k <= 1; ++k) {
    if (k == 1 && !f.asg[0].has_mate_pair) break;
    const bool second = k == 1 || f.asg[0].o1_from_r2;
    const char *read = second ? f.r2 : f.r1;
    const uint32_t len = second ? f.l2 : f.l1;
    for (uint32_t i = 0; i < f.n; ++i) {
      const int allele = f.asg[i].allele_idx;
      if (allele % parts != part) continue;
      const t1k_overlap &o = endOverlap(f.asg[i], k);
      if (!candidateIn(allele, o.seq_start, o.seq_end + 1)) continue;  // (no column of this overlap stands on a candidate)
      const ReadView r{read, len, o.strand == -1};
      const int L = (int)ref_.seqs[allele].size();
      const int8_t *e; uint32_t ne;
      endOps(f.asg[i], k, ops, e, ne);
      int refPos = o.seq_start, readPos = o.read_start;
      for (uint32_t j = 0; j < ne; ++j) {  // every column, gaps included (637)
        const int c = refPos < L ? cell(allele, refPos).cand : -1;
        if (c != -1) {
          const char nuc = readPos < (int)len ? r.at(readPos) : '\0';  // (behind the read's last base the reference reads its terminator)
          std::vector<std::pair<uint32_t, char>> &sv = seen_[c];
          bool have = false;
          for (size_t q = sv.size(); q > 0 && sv[q - 1].first == fragIdx; --q)
            if (sv[q - 1].second == nuc) { have = true; break; }
          if (!have) sv.emplace_back(fragIdx, nuc);
        }
        if (e[j] != OP_INSERT) ++refPos;
        if (e[j] != OP_DELETE) ++readPos;
      }
    }
  }
}

// SolveVariantGroup (822-976) with EnumerateVariants (689-820) as an odometer over the 4^n nucleotide choices, first candidate slowest
void VariantCaller::solveGroup(const std::vector<int> &vars, int groupId) {
  const int n = (int)vars.size();
  if (maxGroup_ >= 0 && n > maxGroup_) return;
  {
    bool exonic = false;
    std::map<int, int> perAllele;
    for (int v : vars) {
      if (ref_.exon[cand_[v].first][cand_[v].second]) exonic = true;
      if (++perAllele[cand_[v].first] > 1) return;  // two candidates on one allele: not resolved (844-849)
    }
    if (!exonic) return;
  }
  // the fragments that show a nucleotide at one of the group's bases, numbered densely
  std::unordered_map<uint32_t, uint32_t> local;
  for (int v : vars)
    for (auto &fn : seen_[v]) local.emplace(fn.first, (uint32_t)local.size());
  const uint32_t nFrag = (uint32_t)local.size();
  std::vector<std::vector<std::pair<uint32_t, char>>> at(n);
  for (int i = 0; i < n; ++i)
    for (auto &fn : seen_[vars[i]]) at[i].emplace_back(local[fn.first], fn.second);
  std::vector<char> own(n);
  for (int i = 0; i < n; ++i) own[i] = ref_.seqs[cand_[vars[i]].first][cand_[vars[i]].second];

  double bestCover = -1;
  int bestUsed = n + 1;
  std::vector<char> choice(n, 'A'), best, equalBest;
  std::vector<int> digit(n, 0);
  std::vector<uint8_t> covered(nFrag);
  for (;;) {
    for (int i = 0; i < n; ++i) choice[i] = kNuc[digit[i]];
    std::fill(covered.begin(), covered.end(), 0);
    for (int i = 0; i < n; ++i) {
      if (n <= 1 && copies_[cand_[vars[i]].first] <= 1 && choice[i] != own[i]) continue;  // 708-710
      for (auto &fn : at[i])
        if (fn.second == choice[i]) covered[fn.first] = 1;
    }
    if (n <= 1) {  // 732-780: a lone candidate on a gene with a single loaded allele
      const int i = 0;
      if (copies_[cand_[vars[i]].first] == 1 && choice[i] != own[i]) {
        int ownSeen = 0, altSeen = 0;
        for (auto &fn : at[i]) {
          if (fn.second == choice[i]) ++altSeen;
          else if (fn.second == own[i]) ++ownSeen;
        }
        const Cell &c = cell(cand_[vars[i]].first, cand_[vars[i]].second);
        const bool withAlt = ((altSeen >= 2 && c.uniq[nucIndex(choice[i])] > 0) || altSeen >= 10) && altSeen > 0.15 * ownSeen;
        for (auto &fn : at[i])
          if (fn.second == own[i] || (fn.second == choice[i] && withAlt))
            if (!covered[fn.first]) covered[fn.first] = 2;
      }
    }
    double cover = 0;
    for (uint32_t q = 0; q < nFrag; ++q)
      if (covered[q]) ++cover;
    int used = 0;
    for (int i = 0; i < n; ++i)
      if (own[i] != choice[i]) ++used;
    if (cover > bestCover || (cover == bestCover && used < bestUsed)) {
      bestCover = cover; bestUsed = used; best = choice; equalBest.clear();
    } else if (cover == bestCover && used == bestUsed) equalBest = choice;
    int d = n - 1;  // next choice: the last candidate runs fastest (the recursion's innermost loop)
    while (d >= 0 && ++digit[d] == 4) digit[d--] = 0;
    if (d < 0) break;
  }
  const bool unique = equalBest.empty();
  for (int pass = 0; pass < (unique ? 1 : 2); ++pass) {
    const std::vector<char> &pick = pass ? equalBest : best;
    for (int i = 0; i < n; ++i) {
      const int a = cand_[vars[i]].first, p = cand_[vars[i]].second;
      if (!ref_.exon[a][p] || own[i] == pick[i]) continue;
      const Cell &c = cell(a, p);
      VariantRec v;
      v.allele = a; v.refPos = p; v.ref = own[i]; v.var = pick[i];
      v.allSupport = c.depth();
      v.varSupport = c.count[nucIndex(pick[i])];
      v.varUniqSupport = c.uniq[nucIndex(pick[i])];
      v.group = groupId; v.outputGroup = pass; v.qual = unique ? 60 : 0;
      variants.push_back(v);
    }
  }
}

// ComputeVariant (978-1140)
void VariantCaller::compute(const std::vector<Fragment> &frags, const int8_t *ops) {
  if (maxGroup_ == 0) return;  // 980-981
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  // the sweeps that touch one allele's cells (or one allele's candidates) per assignment run on threads that own the alleles
  int parts = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16, ref_.seqs.size()});
  if (frags.size() < 20000) parts = 1;
  if (const char *e = getenv("T1K_VARIANTS_THREADS")) parts = std::max(1, std::min(64, atoi(e)));  // (tests: the threaded sweeps on small inputs)
  auto sweep = [&](auto fn) {
    if (parts == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < parts; ++t) th.emplace_back([&, t] { fn(t); });
    for (auto &x : th) x.join();
  };
  // The two booking sweeps (UpdateBaseVariantFromFragmentOverlap with updateType 1, then 0).  What they add to a cell are whole numbers (count += 1,
  // uniq += 1 for a weight of exactly 1) and maxima (best match counts), and the second sweep's filter reads best match counts the first has made
  // final: the result does not depend on the order of the fragments.  Round 6: threads take contiguous pieces of the FRAGMENTS, each into a table of its
  // own, and the tables are folded into the job's cells (sums of whole numbers in doubles below 2^53 and maxima: exact) -- threads that own alleles
  // (round 5) all walk every fragment and wait for the one that owns the sample's most abundant allele.  Tables beyond 1 GB in all: ownership as before.
  Cell *const mine = cells_->data();
  const size_t nCells = cells_->size();
  int bookT = (int)std::min<size_t>((size_t)std::max(1u, std::thread::hardware_concurrency()), 16);
  if (frags.size() < 20000) bookT = 1;
  if (const char *e = getenv("T1K_VARIANTS_THREADS")) bookT = std::max(1, std::min(64, atoi(e)));
  if (bookT > 1 && nCells * sizeof(Cell) * (size_t)bookT <= ((size_t)1 << 30)) {
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<std::vector<Cell>> priv((size_t)bookT);
      std::vector<std::thread> th;
      for (int t = 0; t < bookT; ++t)
        th.emplace_back([&, t] {
          priv[t].assign(nCells, Cell());
          const size_t lo = frags.size() * (size_t)t / bookT, hi = frags.size() * (size_t)(t + 1) / bookT;
          for (size_t i = lo; i < hi; ++i) bookFragment(frags[i], ops, pass == 0, 0, 1, priv[t].data(), mine);
        });
      for (auto &x : th) x.join();
      th.clear();
      for (int t = 0; t < bookT; ++t)   // the fold, by cell ranges
        th.emplace_back([&, t] {
          const size_t lo = nCells * (size_t)t / bookT, hi = nCells * (size_t)(t + 1) / bookT;
          for (int u = 0; u < bookT; ++u) {
            const Cell *src = priv[u].data();
            for (size_t c = lo; c < hi; ++c) {
              Cell &d = mine[c];
              const Cell &q = src[c];
              for (int b = 0; b < 4; ++b) { d.count[b] += q.count[b]; d.uniq[b] += q.uniq[b]; if (q.bestMatch[b] > d.bestMatch[b]) d.bestMatch[b] = q.bestMatch[b]; }
              if (q.bestOfAll > d.bestOfAll) d.bestOfAll = q.bestOfAll;
            }
          }
        });
      for (auto &x : th) x.join();
    }
  } else {
    sweep([&](int t) { for (const Fragment &f : frags) bookFragment(f, ops, true, t, parts, mine, mine); });
    sweep([&](int t) { for (const Fragment &f : frags) bookFragment(f, ops, false, t, parts, mine, mine); });
  }
  findRoots();
  const double t1 = now();
  int rounds = 0;
  const size_t nRoot = cand_.size();
  edges_.assign(nRoot, {});
  for (;;) {  // 1049-1070: the edge weights are counted afresh in every round, over the candidates of the round before
    const size_t before = cand_.size();
    for (auto &e : edges_) e.clear();
    for (const Fragment &f : frags) expandFragment(f, ops);
    ++rounds;
    if (cand_.size() == before) break;
  }
  const double t2 = now();
  // groups: components reached from the root candidates, numbered in the order of their first root (1080-1088)
  int nGroups = 0;
  std::vector<int> stack;
  for (size_t r = 0; r < cand_.size(); ++r) {
    if (!root_[r] || group_[r] != -1) continue;
    stack.assign(1, (int)r);
    group_[r] = nGroups;
    while (!stack.empty()) {
      const int from = stack.back();
      stack.pop_back();
      const double depthFrom = cell(cand_[from].first, cand_[from].second).depth();
      for (auto &e : edges_[from]) {
        if (group_[e.first] != -1) continue;
        const double depthTo = cell(cand_[e.first].first, cand_[e.first].second).depth();
        if (e.second >= depthFrom * 0.15 || e.second >= depthTo * 0.15) { group_[e.first] = nGroups; stack.push_back(e.first); }
      }
    }
    ++nGroups;
  }
  seen_.assign(cand_.size(), {});
  sweep([&](int t) { for (size_t i = 0; i < frags.size(); ++i) linkFragment(frags[i], (uint32_t)i, ops, t, parts); });
  const double t3 = now();
  std::vector<std::vector<int>> members(nGroups);
  for (size_t c = 0; c < cand_.size(); ++c)
    if (group_[c] != -1) members[group_[c]].push_back((int)c);
  for (int g = 0; g < nGroups; ++g) solveGroup(members[g], g);
  for (size_t v = 0; v < variants.size(); ++v) calledAt_[base_[variants[v].allele] + (size_t)variants[v].refPos].push_back((int)v);
  if (getenv("T1K_DEBUG_PHASES"))
    fprintf(stderr, "[t1k variants] base tables %.1f ms, %d expansion rounds %.1f ms (%zu candidates, %zu of them roots), fragment links %.1f ms, %d groups solved in %.1f ms\n", t1 - t0, rounds,
            t2 - t1, cand_.size(), (size_t)std::count(root_.begin(), root_.end(), (char)1), t3 - t2, nGroups, now() - t3);
}

// SeqSet::GetExonicPosition (SeqSet.hpp:2808-2828) for a base inside an exon: the exonic bases in front of it (the exons of a record are
// listed in order and do not overlap, SeqSet.hpp:935-958)
static int exonicPosition(const RefSet &ref, int allele, int pos) {
  if (!ref.exon[allele][pos]) return -1;
  int n = 0;
  for (int i = 0; i < pos; ++i) n += ref.exon[allele][i] ? 1 : 0;
  return n;
}

// OutputAlleleVCF (1202-1227)
std::string VariantCaller::vcfText() const {
  std::string out;
  char line[1024];
  for (const VariantRec &v : variants) {
    snprintf(line, sizeof line, "%s %d . %c %c . %s %lf %lf %lf %d %d\n", ref_.al[v.allele].name.c_str(), exonicPosition(ref_, v.allele, v.refPos) + 1, v.ref, v.var,
             v.qual > 0 ? "PASS" : "FAIL", v.varSupport, v.allSupport, v.varUniqSupport, v.refPos, v.outputGroup);
    out += line;
  }
  return out;
}

// AdjustFragmentAssignment (1229-1311): keep[i] = 1 for the assignments with the most mismatches that are called variants
void VariantCaller::adjust(const Fragment &f, const int8_t *ops, uint8_t *keep) const {
  std::vector<double> score(f.n, 0.0);
  for (uint32_t i = 0; i < f.n; ++i) {
    const t1k_frag_assignment &a = f.asg[i];
    for (int k = 0; k < 2; ++k) {
      if (k == 1 && !a.has_mate_pair) continue;
      const bool second = k == 1 || a.o1_from_r2;
      const char *read = second ? f.r2 : f.r1;
      const uint32_t len = second ? f.l2 : f.l1;
      const t1k_overlap &o = endOverlap(a, k);
      const ReadView r{read, len, o.strand == -1};
      const int8_t *e; uint32_t ne;
      endOps(a, k, ops, e, ne);
      int refPos = o.seq_start, readPos = o.read_start;
      for (uint32_t j = 0; j < ne; ++j) {
        if (e[j] == OP_MISMATCH && !calledAt_.empty()) {
          auto it = calledAt_.find(base_[o.seq_idx] + (size_t)refPos);
          if (it != calledAt_.end())
            for (int v : it->second)
              if (variants[v].var == r.at(readPos)) { score[i] += 1; break; }
        }
        if (e[j] != OP_INSERT) ++refPos;
        if (e[j] != OP_DELETE) ++readPos;
      }
    }
  }
  double top = -1;
  for (uint32_t i = 0; i < f.n; ++i) top = std::max(top, score[i]);
  for (uint32_t i = 0; i < f.n; ++i) keep[i] = score[i] == top ? 1 : 0;
}

// ---- which overlaps stand behind a fragment's assignment -----------------------------------------------------------------------------
// The fragment rows of the device path (k_pair) keep, per assigned allele, the fragment's window and weight -- what the genotyper reads.
// The variant caller also reads the two read-ends' own overlaps (_fragmentOverlap::overlap1 / overlap2, SeqSet.hpp:158-159), so the
// choice SeqSet::ReadAssignmentToFragmentAssignment makes per allele (SeqSet.hpp:2310-2458) is taken again here, on the host, from the
// read-ends' final overlap lists (t1k_overlaps_download: AssignRead's order), for the alleles the device kept and in the device's order.

// _overlap::operator< (SeqSet.hpp:103-127)
static bool overlapRanksBefore(const t1k_overlap &a, const t1k_overlap &b) {
  if (a.match_cnt != b.match_cnt) return a.match_cnt > b.match_cnt;
  if (a.similarity != b.similarity) return a.similarity > b.similarity;
  if (a.read_end - a.read_start != b.read_end - b.read_start) return a.read_end - a.read_start > b.read_end - b.read_start;
  if (a.seq_idx != b.seq_idx) return a.seq_idx < b.seq_idx;
  if (a.strand != b.strand) return a.strand < b.strand;
  if (a.read_start != b.read_start) return a.read_start < b.read_start;
  if (a.read_end != b.read_end) return a.read_end < b.read_end;
  if (a.seq_start != b.seq_start) return a.seq_start < b.seq_start;
  return a.seq_end < b.seq_end;
}

bool fragmentDetails(const t1k_overlap *l1, uint32_t n1, const t1k_overlap *l2, uint32_t n2, bool paired, const int32_t *alleles, uint32_t nAlleles, t1k_frag_assignment *out) {
  struct Pick { int i = -1, j = -1, matchCnt = 0; double sim = 0; };
  for (uint32_t q = 0; q < nAlleles; ++q) {
    const int A = alleles[q];
    Pick best;
    bool have = false;
    auto offer = [&](const Pick &c) {
      if (!have) { best = c; have = true; return; }
      // _fragmentOverlap::operator< (SeqSet.hpp:164-171): a later candidate replaces the kept one only when it ranks strictly before it
      const t1k_overlap &oc = c.i >= 0 ? l1[c.i] : l2[c.j], &ob = best.i >= 0 ? l1[best.i] : l2[best.j];
      bool before;
      if (c.matchCnt != best.matchCnt) before = c.matchCnt > best.matchCnt;
      else if (c.sim != best.sim) before = c.sim > best.sim;
      else before = overlapRanksBefore(oc, ob);
      if (before) best = c;
    };
    if (!paired || n1 == 0 || n2 == 0) {  // single-end run, or one mate without an overlap: every overlap on its own (2321-2346)
      for (uint32_t i = 0; i < n1; ++i)
        if (l1[i].seq_idx == A) { Pick c; c.i = (int)i; c.matchCnt = l1[i].match_cnt; c.sim = l1[i].similarity; offer(c); }
      if (paired)
        for (uint32_t j = 0; j < n2; ++j)
          if (l2[j].seq_idx == A) { Pick c; c.j = (int)j; c.matchCnt = l2[j].match_cnt; c.sim = l2[j].similarity; offer(c); }
    } else {
      for (uint32_t i = 0; i < n1; ++i) {
        if (l1[i].seq_idx != A) continue;
        const t1k_overlap &o = l1[i];
        for (uint32_t j = 0; j < n2; ++j) {
          const t1k_overlap &o2 = l2[j];
          if (o2.seq_idx != A || o.strand == o2.strand) continue;  // 2364-2368
          if (!((o.strand == 1 && o.seq_start < o2.seq_start) || (o.strand == -1 && o.seq_start > o2.seq_start))) continue;
          Pick c;
          c.i = (int)i; c.j = (int)j;
          c.matchCnt = o.match_cnt + o2.match_cnt;
          c.sim = (double)c.matchCnt / (o.read_end - o.read_start + 1 + o2.read_end - o2.read_start + 1 + o.seq_end - o.seq_start + 1 + o2.seq_end - o2.seq_start + 1 +
                                         2 * o.left_clip + 2 * o.right_clip + 2 * o2.left_clip + 2 * o2.right_clip);  // 2411-2414
          offer(c);
        }
      }
    }
    if (!have) return false;  // the device kept an allele the lists do not explain
    t1k_frag_assignment &a = out[q];
    memset(&a, 0, sizeof a);
    a.allele_idx = A;
    if (best.i >= 0) {
      a.o1 = l1[best.i];
      if (best.j >= 0) { a.has_mate_pair = 1; a.o2 = l2[best.j]; }
    } else {
      a.o1 = l2[best.j];
      a.o1_from_r2 = 1;
    }
  }
  return true;
}

}  // namespace t1k
