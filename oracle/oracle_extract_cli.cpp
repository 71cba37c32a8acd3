// oracle/oracle_extract_cli.cpp -- TEST INFRASTRUCTURE ONLY.
// Driver around the extraction restatement with the option set and output files of the reference's fastq-extractor
// (FastqExtractor.cpp:260-626; barcode options are not restated).  --flags FILE additionally writes one 0/1 per fragment.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "oracle_core.hpp"

using namespace t1k_oracle;

static void outputSeq(FILE *fp, const std::string &name, const std::string &seq, const std::string &qual, int start, int end) {  // FastqExtractor.cpp:120-154
  std::string s = seq, q = qual;
  if (!(start == 0 && end == -1)) {
    int e = end == -1 ? (int)seq.size() - 1 : end;
    s = seq.substr(start, e - start + 1);
    if (!qual.empty()) q = qual.substr(start, e - start + 1);
  }
  if (!qual.empty()) fprintf(fp, "@%s\n%s\n+\n%s\n", name.c_str(), s.c_str(), q.c_str());
  else fprintf(fp, ">%s\n%s\n", name.c_str(), s.c_str());
}

int main(int argc, char **argv) {
  std::string ref, prefix = "toassemble", flagsPath;
  std::vector<std::string> f1, f2;
  bool hasMate = false, interleaved = false;
  int threads = 1, r1s = 0, r1e = -1, r2s = 0, r2e = -1;
  Oracle orc;
  orc.prm.k = 9;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
    if (a == "-f") ref = next();
    else if (a == "-u") f1.push_back(next());
    else if (a == "-1") { f1.push_back(next()); hasMate = true; }
    else if (a == "-2") { f2.push_back(next()); hasMate = true; }
    else if (a == "-i") { f1.push_back(next()); hasMate = true; interleaved = true; }
    else if (a == "-o") prefix = next();
    else if (a == "-t") threads = atoi(next().c_str());
    else if (a == "-s") orc.prm.refSeqSimilarity = atof(next().c_str());
    else if (a == "--read1Start") r1s = atoi(next().c_str());
    else if (a == "--read1End") r1e = atoi(next().c_str());
    else if (a == "--read2Start") r2s = atoi(next().c_str());
    else if (a == "--read2End") r2e = atoi(next().c_str());
    else if (a == "--flags") flagsPath = next();
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
  }
  if (ref.empty() || f1.empty()) { fprintf(stderr, "usage: t1k_oracle_extract -f ref.fa (-u r.fq | -1 a.fq -2 b.fq | -i il.fq) [-s S] [-t T] -o prefix\n"); return 1; }
  if (orc.loadReferenceFa(ref) <= 0) { fprintf(stderr, "cannot load %s\n", ref.c_str()); return 1; }
  std::vector<SeqRecord> r1, r2;
  for (auto &p : f1) if (!readAllRecords(p, r1)) { fprintf(stderr, "cannot read %s\n", p.c_str()); return 1; }
  for (auto &p : f2) if (!readAllRecords(p, r2)) { fprintf(stderr, "cannot read %s\n", p.c_str()); return 1; }
  if (interleaved) {
    std::vector<SeqRecord> a, b;
    for (size_t i = 0; i + 1 < r1.size(); i += 2) { a.push_back(r1[i]); b.push_back(r1[i + 1]); }
    r1.swap(a); r2.swap(b);
  }
  if (r1.empty()) { fprintf(stderr, "Read file is empty.\n"); return 1; }
  if (hasMate && r1.size() != r2.size()) { fprintf(stderr, "The two mate-pair read files have different number of reads.\n"); return 1; }
  // FastqExtractor.cpp:383-416
  int hitLenRequired = hasMate ? 27 : 23, len = 0, n = 0;
  for (; n < 1000 && n < (int)r1.size(); ++n) len += (int)r1[n].seq.size();
  if (len / (n * 5) > hitLenRequired) hitLenRequired = len / (n * 5);
  int k = orc.inferKmerLength();
  if (k > 9) {
    if (k > hitLenRequired) hitLenRequired = k;
    orc.setKmerLength(k);
  }
  orc.prm.hitLenRequired = hitLenRequired;
  FILE *fp1 = fopen((prefix + (hasMate ? "_1.fq" : ".fq")).c_str(), "w"), *fp2 = hasMate ? fopen((prefix + "_2.fq").c_str(), "w") : nullptr;
  FILE *ff = flagsPath.empty() ? nullptr : fopen(flagsPath.c_str(), "w");
  for (size_t i = 0; i < r1.size(); ++i) {
    bool good = orc.isGoodCandidate(r1[i].seq) || (hasMate && orc.isGoodCandidate(r2[i].seq));
    if (ff) fputc(good ? '1' : '0', ff);
    if (!good) continue;
    // the single-thread loop prints ReadFiles::Next()'s id (/1 /2 stripped), the batch loop NextWithBuffer()'s raw name (FastqExtractor.cpp:446-476 vs 529-545)
    const std::string &id = threads == 1 ? r1[i].id : r1[i].rawId;
    outputSeq(fp1, id, r1[i].seq, r1[i].qual, r1s, r1e);
    if (hasMate) outputSeq(fp2, id, r2[i].seq, r2[i].qual, r2s, r2e);
  }
  fclose(fp1);
  if (fp2) fclose(fp2);
  if (ff) fclose(ff);
  fprintf(stderr, "oracle extract: k=%d hitLenRequired=%d\n", orc.prm.k, orc.prm.hitLenRequired);
  return 0;
}
