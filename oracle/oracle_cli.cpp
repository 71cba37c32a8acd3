// oracle/oracle_cli.cpp -- TEST INFRASTRUCTURE ONLY.
// Command-line driver around the CPU restatement, following the order of operations of the reference's
// genotyper main() (Genotyper.cpp:194-738) through the EM, the pruning, the selection and the two tables, and dumping every intermediate the
// parity tests compare: <o>_assign.tsv (same format as the reference's --outputReadAssignment, Genotyper.cpp:555-562),
// <o>_overlaps.tsv (per read-end overlap lists), <o>_groups.tsv, <o>_em.tsv, <o>_cov.tsv, <o>_stats.json, and the reference's own
// <o>_genotype.tsv / <o>_allele.tsv.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>

#include "oracle_core.hpp"

using namespace t1k_oracle;

int main(int argc, char **argv) {
  // --dumpReads file...: "id<TAB>seq" per record of the files read back to back, nothing else (the restated reader against the
  // reference's: tests/test_host_reads_cpu.py)
  if (argc >= 2 && !strcmp(argv[1], "--dumpReads")) {
    for (int i = 2; i < argc; ++i) {
      std::vector<SeqRecord> recs;
      if (!readAllRecords(argv[i], recs)) { fprintf(stderr, "cannot read %s\n", argv[i]); return 1; }
      for (auto &r : recs) printf("%s\t%s\n", r.id.c_str(), r.seq.c_str());
    }
    return 0;
  }
  std::string ref, f1, f2, fbc, out = "oracle";
  Oracle orc;
  bool dumpOverlaps = false, noEM = false, fragDump = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
    if (a == "-f") ref = next();
    else if (a == "-1" || a == "-u") f1 = next();
    else if (a == "-2") f2 = next();
    else if (a == "-o") out = next();
    else if (a == "-s") orc.prm.refSeqSimilarity = atof(next().c_str());
    else if (a == "-n") orc.prm.maxAssignCnt = atoi(next().c_str());
    else if (a == "-t") next();
    else if (a == "--frac") orc.prm.filterFrac = atof(next().c_str());
    else if (a == "--relaxIntronAlign") orc.prm.relaxIntronAlign = true;
    else if (a == "--alleleDigitUnits") orc.prm.alleleDigitUnits = atoi(next().c_str());
    else if (a == "--alleleDelimiter") orc.prm.alleleDelimiter = next()[0];
    else if (a == "--squaremMinAlpha") orc.prm.minSquaremAlpha = atof(next().c_str());
    else if (a == "--barcode") fbc = next();
    else if (a == "--dumpOverlaps") dumpOverlaps = true;
    else if (a == "--noEM") noEM = true;
    else if (a == "--fragDump") fragDump = true;  // <o>_fragdump.tsv: what the reference's analyzer hands its VariantCaller (Analyzer.cpp:560-571, 611-684)
    else if (a == "--cov") orc.prm.filterCov = atof(next().c_str());
    else if (a == "--crossGeneRate") orc.prm.crossGeneRate = atof(next().c_str());
    else if (a == "--outputReadAssignment") {}
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
  }
  if (ref.empty() || f1.empty()) { fprintf(stderr, "usage: t1k_oracle_cli -f ref.fa -1 a.fq [-2 b.fq] [-s S] [--relaxIntronAlign] -o prefix\n"); return 1; }
  if (orc.loadReference(ref) <= 0) { fprintf(stderr, "cannot load reference %s\n", ref.c_str()); return 1; }
  std::vector<SeqRecord> r1, r2;
  if (!readAllRecords(f1, r1)) { fprintf(stderr, "cannot read %s\n", f1.c_str()); return 1; }
  bool hasMate = !f2.empty();
  if (hasMate && !readAllRecords(f2, r2)) { fprintf(stderr, "cannot read %s\n", f2.c_str()); return 1; }
  if (!fbc.empty()) {  // fragments without a barcode are not loaded at all (Genotyper.cpp:372-381); the barcode itself plays no part in the assignment
    std::vector<SeqRecord> bc;
    if (!readAllRecords(fbc, bc) || bc.size() != r1.size()) { fprintf(stderr, "cannot read %s\n", fbc.c_str()); return 1; }
    size_t kept = 0;
    for (size_t i = 0; i < r1.size(); ++i) {
      if (bc[i].seq == "missing_barcode") continue;
      if (kept != i) { r1[kept] = r1[i]; if (hasMate) r2[kept] = r2[i]; }
      ++kept;
    }
    r1.resize(kept);
    if (hasMate) r2.resize(kept);
  }
  size_t F = r1.size();
  auto t0 = std::chrono::steady_clock::now();
  // concat mates, sort by sequence, one AssignRead per distinct sequence with weight = multiplicity (Genotyper.cpp:451-480)
  std::vector<const std::string *> ends;
  for (auto &r : r1) ends.push_back(&r.seq);
  for (auto &r : r2) ends.push_back(&r.seq);
  std::vector<uint32_t> ord(ends.size());
  std::iota(ord.begin(), ord.end(), 0u);
  std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return *ends[a] < *ends[b]; });
  std::vector<int> listOf(ends.size());
  std::vector<std::vector<Overlap>> lists;
  for (size_t i = 0; i < ord.size();) {
    size_t j = i + 1;
    while (j < ord.size() && *ends[ord[j]] == *ends[ord[i]]) ++j;
    lists.emplace_back();
    orc.assignRead(*ends[ord[i]], (int)(j - i), lists.back());
    for (size_t q = i; q < j; ++q) listOf[ord[q]] = (int)lists.size() - 1;
    i = j;
  }
  auto t1 = std::chrono::steady_clock::now();
  if (dumpOverlaps) {
    FILE *fo = fopen((out + "_overlaps.tsv").c_str(), "w");
    for (size_t e = 0; e < ends.size(); ++e) {
      const char *id = e < F ? r1[e].id.c_str() : r2[e - F].id.c_str();
      for (auto &o : lists[listOf[e]])
        fprintf(fo, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.17g\t%d\t%d\t%d\n", id, e < F ? 1 : 2, o.seqIdx, o.readStart, o.readEnd, o.seqStart,
                o.seqEnd, o.strand, o.matchCnt, o.similarity, o.leftClip, o.rightClip, o.relaxedMatchCnt);
    }
    fclose(fo);
  }
  FILE *fa = fopen((out + "_assign.tsv").c_str(), "w");
  FILE *fal = fopen((out + "_aligned_ids.txt").c_str(), "w");
  std::vector<FragmentOverlap> frag;
  std::vector<RowEntry> row;
  size_t assignedFragments = 0;
  FILE *fd = fragDump ? fopen((out + "_fragdump.tsv").c_str(), "w") : nullptr;
  // one overlap with the edit string SeqSet::AddOverlapAlignmentInfo gives it (SeqSet.hpp:2657-2681)
  auto dumpOverlap = [&](const Overlap &o, const std::string &read) {
    const std::string r = o.strand == -1 ? reverseComplement(read) : read;
    std::vector<int8_t> ops;
    globalAlignment(orc.alleles[o.seqIdx].seq.c_str() + o.seqStart, o.seqEnd - o.seqStart + 1, r.c_str() + o.readStart, o.readEnd - o.readStart + 1, ops);
    std::string e;
    for (int8_t c : ops) e += (char)('0' + c);
    fprintf(fd, "\t%d\t%d\t%d\t%d\t%d\t%d\t%.17g\t%d\t%d\t%d\t%s", o.readStart, o.readEnd, o.seqStart, o.seqEnd, o.strand, o.matchCnt, o.similarity, o.leftClip, o.rightClip,
            o.relaxedMatchCnt, e.empty() ? "-" : e.c_str());
  };
  for (size_t i = 0; i < F; ++i) {
    bool hasN = r1[i].seq.find('N') != std::string::npos || (hasMate && r2[i].seq.find('N') != std::string::npos);
    orc.pairFragments(lists[listOf[i]], hasMate ? &lists[listOf[F + i]] : nullptr, hasN, frag);
    orc.fragmentToRow(frag, row);
    for (auto &e : row) fprintf(fa, "%s\t%s\t%d\t%d\n", r1[i].id.c_str(), orc.alleles[e.alleleIdx].name.c_str(), e.start, e.end);
    if (!frag.empty()) fprintf(fal, "%s\n", r1[i].id.c_str());  // fragmentAssigned (Genotyper.cpp:564-565)
    if (!row.empty()) ++assignedFragments;
    orc.coalesceRow(row);
    if (fd)
      for (auto &fo : frag) {  // the list itself, before the -n / separator drops of SetReadAssignments (Analyzer.cpp:571)
        fprintf(fd, "%zu\t%d\t%d\t%d", i, fo.seqIdx, fo.hasMatePair ? 1 : 0, fo.o1FromR2 ? 1 : 0);
        dumpOverlap(fo.o1, (fo.o1FromR2 && !fo.hasMatePair) ? r2[i].seq : r1[i].seq);
        if (fo.hasMatePair) dumpOverlap(fo.o2, r2[i].seq);
        fprintf(fd, "\n");
      }
  }
  if (fd) fclose(fd);
  fclose(fa);
  fclose(fal);
  auto t2 = std::chrono::steady_clock::now();
  orc.finalizeGroups();
  {
    FILE *fg = fopen((out + "_groups.tsv").c_str(), "w");
    for (size_t g = 0; g < orc.groups.size(); ++g) {
      fprintf(fg, "%zu\t%zu", g, orc.groups[g].size());
      for (auto &e : orc.groups[g]) fprintf(fg, "\t%d:%d:%d:%.9g:%.9g", e.alleleIdx, e.start, e.end, e.weight, e.adjustWeight);
      fprintf(fg, "\n");
    }
    fclose(fg);
    FILE *fc = fopen((out + "_cov.tsv").c_str(), "w");
    for (size_t a = 0; a < orc.alleles.size(); ++a)
      fprintf(fc, "%s\t%d\t%d\t%d\t%d\n", orc.alleles[a].name.c_str(), orc.alleles[a].missingCoverage, orc.alleles[a].ec, orc.alleles[a].effectiveLen, orc.alleles[a].weight);
    fclose(fc);
  }
  int iters = 0;
  if (!noEM) {
    iters = orc.quantify();
    FILE *fe = fopen((out + "_em.tsv").c_str(), "w");
    fprintf(fe, "#iterations\t%d\n", iters);
    for (size_t i = 0; i < orc.ecAlleles.size(); ++i) {
      fprintf(fe, "%zu\t", i);
      for (size_t j = 0; j < orc.ecAlleles[i].size(); ++j) fprintf(fe, "%s%s", j ? "," : "", orc.alleles[orc.ecAlleles[i][j]].name.c_str());
      fprintf(fe, "\t%d\t%.17g\t%.17g\n", orc.ecLength[i], orc.ecReadCountFinal[i], orc.ecAbundanceFinal[i]);
    }
    fclose(fe);
    FILE *fb = fopen((out + "_abundance.tsv").c_str(), "w");
    for (auto &a : orc.alleles) fprintf(fb, "%s\t%.17g\t%.17g\n", a.name.c_str(), a.abundance, a.ecAbundance);
    fclose(fb);
    // pruning, selection and the two tables, as main() goes on after the EM (Genotyper.cpp:647-676)
    for (auto &r : r1) orc.readLength = std::max(orc.readLength, (int)r.seq.size());  // maxReadLength (Genotyper.cpp:412-443)
    for (auto &r : r2) orc.readLength = std::max(orc.readLength, (int)r.seq.size());
    orc.removeLowLikelihoodAlleles();
    orc.selectAllelesForGenes();
    FILE *fgt = fopen((out + "_genotype.tsv").c_str(), "w");
    fputs(orc.genotypeText().c_str(), fgt);
    fclose(fgt);
    FILE *fat = fopen((out + "_allele.tsv").c_str(), "w");
    fputs(orc.alleleText().c_str(), fat);
    fclose(fat);
  }
  auto t3 = std::chrono::steady_clock::now();
  auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
  FILE *fs = fopen((out + "_stats.json").c_str(), "w");
  fprintf(fs,
          "{\"fragments\": %zu, \"assigned_fragments\": %zu, \"distinct_read_ends\": %zu, \"alleles\": %zu, \"groups\": %zu, \"ecs\": %zu, "
          "\"em_iterations\": %d, \"lookups\": %llu, \"postings\": %llu, \"candidates\": %llu, \"extended\": %llu, \"near_best\": %llu, "
          "\"ga_calls\": %llu, \"ga_cells\": %llu, \"t_assign\": %.4f, \"t_pair_coalesce\": %.4f, \"t_em\": %.4f}\n",
          F, assignedFragments, lists.size(), orc.alleles.size(), orc.groups.size(), orc.ecAlleles.size(), iters,
          (unsigned long long)orc.stats.lookups, (unsigned long long)orc.stats.postings, (unsigned long long)orc.stats.candidates,
          (unsigned long long)orc.stats.extended, (unsigned long long)orc.stats.nearBest, (unsigned long long)orc.stats.gaCalls,
          (unsigned long long)orc.stats.gaCells, sec(t0, t1), sec(t1, t2), sec(t2, t3));
  fclose(fs);
  fprintf(stderr, "oracle: %zu fragments, %zu distinct read-ends, %zu groups, %zu ECs, %d EM iterations; assign %.2fs pair+coalesce %.2fs em %.2fs\n", F,
          lists.size(), orc.groups.size(), orc.ecAlleles.size(), iters, sec(t0, t1), sec(t1, t2), sec(t2, t3));
  return 0;
}
