// tests/harness/host_tables_harness.cpp -- TEST INFRASTRUCTURE (CPU): the product's host side after the device loop
// (t1k_amd/csrc/host/refset.cpp + genotype.cpp, plain C++) fed with the intermediates the oracle CLI dumps:
//   reference   RefSet::load against <oracle>_cov.tsv (names, merged weights, effective lengths)
//   classes     Genotyper::finalize on the read groups of <oracle>_groups.tsv against the class ids of <oracle>_cov.tsv
//   abundances  Genotyper::setAbundance with the EM result of <oracle>_em.tsv (the E-step itself is a device stage: not run here)
//   pruning, selection, quality, the two tables  ->  <out>_genotype.tsv, <out>_allele.tsv (compared by the test)
//   host_tables_harness ref.fa oraclePrefix readLength frac cov crossGeneRate outPrefix [alleleDigitUnits [alleleDelimiter]]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include "../../t1k_amd/csrc/host/t1k_host.h"

// Genotyper::quantify drives the device E-step; this harness never calls it
extern "C" {
int t1k_em_setup(t1k_ctx *, const uint64_t *, const uint32_t *, const double *, const int32_t *, uint32_t, uint32_t, t1k_allreduce_fn, void *) { abort(); }
int t1k_em_update(t1k_ctx *, const double *, double *, double *, double *) { abort(); }
int t1k_em_shard(t1k_ctx *, uint32_t, uint32_t, t1k_comm *) { abort(); }
void t1k_em_times(const t1k_ctx *, double *) { abort(); }
int t1k_comm_size(const t1k_comm *) { abort(); }
int t1k_comm_rank(const t1k_comm *) { abort(); }
const char *t1k_last_error(const t1k_ctx *) { return "no device in this harness"; }
}

static std::vector<std::string> split(const std::string &s, char c) {
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    size_t e = s.find(c, b);
    out.push_back(s.substr(b, e == std::string::npos ? e : e - b));
    if (e == std::string::npos) break;
    b = e + 1;
  }
  return out;
}

int main(int argc, char **argv) {
  if (argc < 8) { fprintf(stderr, "usage: host_tables_harness ref.fa oraclePrefix readLength frac cov crossGeneRate outPrefix\n"); return 2; }
  const std::string refFa = argv[1], orc = argv[2], out = argv[7];
  std::string err;
  t1k::RefSet R;
  const int digitUnits = argc > 8 ? atoi(argv[8]) : -1;
  const char delimiter = argc > 9 ? argv[9][0] : 0;
  if (!R.load(refFa, digitUnits, delimiter, err)) { fprintf(stderr, "reference: %s\n", err.c_str()); return 1; }
  // ---- reference + class ids expected by the oracle ----
  std::vector<int> missing, wantEc;
  {
    std::ifstream f(orc + "_cov.tsv");
    std::string line;
    size_t a = 0;
    while (std::getline(f, line)) {
      const auto c = split(line, '\t');
      if (a >= R.al.size()) { fprintf(stderr, "the oracle holds more alleles than the %zu loaded here\n", R.al.size()); return 1; }
      if (c[0] != R.al[a].name || atoi(c[3].c_str()) != R.al[a].effLen || atoi(c[4].c_str()) != R.al[a].weight) {
        fprintf(stderr, "allele %zu: %s effLen %d weight %d here, %s\n", a, R.al[a].name.c_str(), R.al[a].effLen, R.al[a].weight, line.c_str());
        return 1;
      }
      missing.push_back(atoi(c[1].c_str()));
      wantEc.push_back(atoi(c[2].c_str()));
      ++a;
    }
    if (a != R.al.size()) { fprintf(stderr, "%zu alleles loaded, the oracle holds %zu\n", R.al.size(), a); return 1; }
  }
  // ---- read groups ----
  t1k::Genotyper g;
  g.ref = &R;
  memset(&g.prm, 0, sizeof(g.prm));  // (t1k_job_params_default lives with the job, which needs the device library)
  g.prm.allele_digit_units = digitUnits;
  g.prm.allele_delimiter = delimiter;
  g.readLength = atoi(argv[3]);
  g.prm.filter_frac = atof(argv[4]);
  g.prm.filter_cov = atof(argv[5]);
  g.prm.cross_gene_rate = atof(argv[6]);
  {
    std::ifstream f(orc + "_groups.tsv");
    std::string line;
    while (std::getline(f, line)) {
      const auto c = split(line, '\t');
      for (size_t j = 2; j < c.size(); ++j) {
        const auto e = split(c[j], ':');
        t1k::GroupEntry ge;
        ge.allele = atoi(e[0].c_str()); ge.start = atoi(e[1].c_str()); ge.end = atoi(e[2].c_str());
        ge.weight = strtof(e[3].c_str(), nullptr); ge.adjustWeight = strtof(e[4].c_str(), nullptr);
        g.groupEnt.push_back(ge);
      }
      g.groupPtr.push_back(g.groupEnt.size());
    }
  }
  std::vector<int32_t> miss32(missing.begin(), missing.end());
  g.finalize(miss32);
  for (size_t a = 0; a < R.al.size(); ++a)
    if (R.al[a].ec != wantEc[a]) { fprintf(stderr, "allele %s: class %d here, %d in the oracle\n", R.al[a].name.c_str(), R.al[a].ec, wantEc[a]); return 1; }
  // ---- EM result ----
  std::vector<double> n;
  std::vector<int> ecLen;
  {
    std::ifstream f(orc + "_em.tsv");
    std::string line;
    while (std::getline(f, line)) {
      if (line.empty() || line[0] == '#') continue;
      const auto c = split(line, '\t');
      ecLen.push_back(atoi(c[2].c_str()));
      n.push_back(strtod(c[3].c_str(), nullptr));
    }
    if (n.size() != g.ecAlleles.size()) { fprintf(stderr, "%zu classes here, %zu in the oracle\n", g.ecAlleles.size(), n.size()); return 1; }
  }
  g.setAbundance(n.data(), ecLen);
  g.dropUnlikely();
  g.select();
  {
    std::ofstream f(out + "_genotype.tsv");
    for (size_t gene = 0; gene < R.geneName.size(); ++gene) f << g.geneLine((int)gene);
  }
  {
    std::ofstream f(out + "_allele.tsv");
    f << g.alleleLines();
  }
  fprintf(stderr, "%zu alleles, %zu groups, %zu classes\n", R.al.size(), g.nGroups(), g.ecAlleles.size());
  return 0;
}
