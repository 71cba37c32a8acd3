// t1k_synth -- seeded synthetic reference + read generator for the genotyper hot path.
//
// The real allele databases named in BASELINE.json (hlaidx_rna_seq.fa, kiridx_dna_seq.fa) cannot be
// downloaded here (reference t1k-build.pl:117-136 uses curl), so benchmarks and large parity cases run
// on references with the same *shape*: the FASTA format written by the reference's ParseDatFile.pl
// (header ">NAME nExon s0 e0 s1 e1 ...", rna = exons concatenated, dna = exons + introns padded +-200
// around a single 'N' separator, ParseDatFile.pl:43,297-328), many near-identical alleles per gene,
// identical sequences under different names, a few indel alleles.
//
//   t1k_synth ref-rna  --seed S --genes G --scale X            > ref.fa
//   t1k_synth ref-dna  --seed S --genes G --scale X            > ref.fa
//   t1k_synth reads    --ref ref.fa --seed S --pairs F --len L --out PFX [--fasta] [--sub r] [--indel r]
//                      [--nrate r] [--bg r] [--barcodes N]
//     writes PFX_1.fq PFX_2.fq (and PFX_bc.fa with --barcodes) and PFX_truth.tsv
//
// Everything is driven by one xoshiro256** stream seeded through splitmix64, so a (seed, parameters)
// pair reproduces byte-identical files on any box.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <string>
#include <vector>
#include <algorithm>
#include <map>

struct Rng {
  uint64_t s[4];
  static uint64_t splitmix(uint64_t &x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  int below(int n) { return (int)(uni() * n); }            // [0,n)
  int range(int a, int b) { return a + below(b - a + 1); }  // [a,b]
  double normal() {
    double u1 = uni(), u2 = uni();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

static const char NUC[5] = "ACGT";

static std::string randomSeq(Rng &r, int n) {
  std::string s(n, 'A');
  for (int i = 0; i < n; ++i) s[i] = NUC[r.below(4)];
  return s;
}
static char otherBase(Rng &r, char c) {
  char o;
  do { o = NUC[r.below(4)]; } while (o == c);
  return o;
}
static std::string mutate(Rng &r, const std::string &s, double rate) {
  std::string o = s;
  for (auto &c : o) if (r.uni() < rate) c = otherBase(r, c);
  return o;
}
static std::string revcomp(const std::string &s) {
  std::string o(s.rbegin(), s.rend());
  for (auto &c : o) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
  return o;
}

struct Args {
  std::map<std::string, std::string> kv;
  bool has(const char *k) const { return kv.count(k) > 0; }
  std::string str(const char *k, const char *d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
  double num(const char *k, double d) const { auto it = kv.find(k); return it == kv.end() ? d : atof(it->second.c_str()); }
};

// ---------------------------------------------------------------------------------------------------
// reference generators
// ---------------------------------------------------------------------------------------------------
static const char *GENE_NAMES[] = {"A", "B", "C", "DRB1", "DQB1", "DPB1", "DQA1", "DPA1", "DRB3", "DRB4", "DRB5",
                                   "E", "F", "G", "DMA", "DMB", "DOA", "DOB", "DRA", "MICA", "MICB", "TAP1", "TAP2", "HFE"};

// alleles per gene: heavy-tailed list from SURVEY 8d (3 x 6000, 3 x 2000, rest 20..500), multiplied by scale
static int allelesForGene(int g, double scale, Rng &r) {
  int base;
  if (g < 3) base = 6000;
  else if (g < 6) base = 2000;
  else base = 20 + r.below(481);
  int n = (int)(base * scale + 0.5);
  return n < 2 ? 2 : n;
}

static void emitExonHeader(const std::string &name, const std::vector<std::pair<int, int>> &exons) {
  printf(">%s %d", name.c_str(), (int)exons.size());
  for (auto &e : exons) printf(" %d %d", e.first, e.second);
  printf("\n");
}

static int cmdRefRna(const Args &a) {
  Rng r((uint64_t)a.num("seed", 20250614));
  int G = (int)a.num("genes", 24);
  double scale = a.num("scale", 1.0);
  if (G > 24) G = 24;
  std::string ancestor = randomSeq(r, 1300);
  std::string sub = mutate(r, ancestor, 0.20);  // class-I-like sub-ancestor
  for (int g = 0; g < G; ++g) {
    int len = r.range(1000, 1200);
    std::string root = g < 3 ? mutate(r, sub, 0.05) : mutate(r, ancestor, 0.15 + 0.15 * r.uni());
    root.resize(len);
    // exon layout: 8 contiguous exons
    std::vector<int> cuts;
    for (int i = 0; i < 7; ++i) cuts.push_back(r.range(60, len - 60));
    std::sort(cuts.begin(), cuts.end());
    // polymorphic-site pool: position -> alternative base
    int poolSize = 300;
    std::vector<std::pair<int, char>> pool;
    for (int i = 0; i < poolSize; ++i) {
      int p = r.below(len);
      pool.push_back({p, otherBase(r, root[p])});
    }
    int nAll = allelesForGene(g, scale, r);
    std::vector<std::string> seqs;
    int field1 = 1, field2 = 1;
    for (int i = 0; i < nAll; ++i) {
      std::string s;
      char name[128];
      bool dup = i > 0 && r.uni() < 0.20;
      if (dup) {  // identical exon sequence, different (4th-field) name: merged by the reference (Genotyper.hpp:718-721)
        int src = r.below(i);
        s = seqs[src];
      } else {
        s = root;
        int k = r.range(3, 14);
        for (int j = 0; j < k; ++j) {
          auto &pv = pool[r.below(poolSize)];
          s[pv.first] = pv.second;
        }
        if (r.uni() < 0.02) {  // small indel allele
          int p = r.range(50, len - 50), l = r.range(1, 3);
          if (r.uni() < 0.5) s.erase(p, l);
          else s.insert(p, randomSeq(r, l));
        }
      }
      seqs.push_back(s);
      snprintf(name, sizeof(name), "HLA-%s*%02d:%02d:%02d:%02d", GENE_NAMES[g], field1, field2, 1 + r.below(3), 1 + (i % 97));
      if (++field2 > 40) { field2 = 1; ++field1; }
      std::vector<std::pair<int, int>> exons;
      int prev = 0, L = (int)s.size();
      for (int c : cuts) { int cc = std::min(c, L - 1); if (cc > prev) { exons.push_back({prev, cc - 1}); prev = cc; } }
      exons.push_back({prev, L - 1});
      emitExonHeader(name, exons);
      printf("%s\n", s.c_str());
    }
  }
  return 0;
}

static int cmdRefDna(const Args &a) {
  Rng r((uint64_t)a.num("seed", 20250615));
  int G = (int)a.num("genes", 17);
  double scale = a.num("scale", 1.0);
  std::string exAnc = randomSeq(r, 1600);
  for (int g = 0; g < G; ++g) {
    int nEx = 9;
    // exon + intron-flank layout of the gene root
    std::vector<std::string> exon(nEx), lflank(nEx), rflank(nEx);
    std::string exRoot = mutate(r, exAnc, 0.05 + 0.10 * r.uni());
    int pos = 0;
    for (int e = 0; e < nEx; ++e) {
      int l = e == 0 ? 40 : r.range(50, 300);
      if (pos + l > (int)exRoot.size()) l = (int)exRoot.size() - pos;
      exon[e] = exRoot.substr(pos, l);
      pos += l;
      lflank[e] = randomSeq(r, 200);
      rflank[e] = randomSeq(r, 200);
    }
    int nAll = std::max(2, (int)((30 + r.below(171)) * scale + 0.5));
    std::vector<std::string> seqs;
    for (int i = 0; i < nAll; ++i) {
      // per-allele SNPs in exons (rate 0.2-1%) and intron flanks (0.5%)
      std::string s;
      std::vector<std::pair<int, int>> exons;
      bool dup = i > 0 && r.uni() < 0.15;
      if (dup) {
        // same exons as an earlier allele, different introns -> same exon-only sequence (SeqSet.hpp:1008-1029)
      }
      double er = 0.002 + 0.008 * r.uni();
      for (int e = 0; e < nEx; ++e) {
        if (e > 0) s += mutate(r, lflank[e], 0.005);
        int st = (int)s.size();
        s += dup ? exon[e] : mutate(r, exon[e], er);
        exons.push_back({st, (int)s.size() - 1});
        if (e + 1 < nEx) { s += mutate(r, rflank[e], 0.005); s += 'N'; }
      }
      char name[128];
      snprintf(name, sizeof(name), "KIR%dDL%d*%03d%02d%02d", 2 + g % 2, 1 + g, 1 + i / 20, 1 + i % 20, 1 + r.below(3));
      emitExonHeader(name, exons);
      printf("%s\n", s.c_str());
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// read simulator
// ---------------------------------------------------------------------------------------------------
struct Allele { std::string name, gene, seq; };

static std::vector<Allele> loadFasta(const std::string &path) {
  std::vector<Allele> out;
  FILE *fp = fopen(path.c_str(), "r");
  if (!fp) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(1); }
  char *line = NULL; size_t cap = 0; ssize_t n;
  while ((n = getline(&line, &cap, fp)) > 0) {
    while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
    if (line[0] == '>') {
      Allele al;
      char *sp = strchr(line, ' ');
      al.name = sp ? std::string(line + 1, sp) : std::string(line + 1);
      size_t star = al.name.find('*');
      al.gene = star == std::string::npos ? al.name : al.name.substr(0, star);
      out.push_back(al);
    } else if (!out.empty()) out.back().seq += line;
  }
  free(line);
  fclose(fp);
  return out;
}

static int cmdReads(const Args &a) {
  Rng r((uint64_t)a.num("seed", 2));
  std::vector<Allele> ref = loadFasta(a.str("ref", ""));
  long F = (long)a.num("pairs", 1000);
  int L = (int)a.num("len", 150);
  double subR = a.num("sub", 0.002), indelR = a.num("indel", 0.00005), nR = a.num("nrate", 0.0001), bgR = a.num("bg", 0.01);
  double fragMean = a.num("fragmean", 350), fragSd = a.num("fragsd", 40);
  int nBarcodes = (int)a.num("barcodes", 0);
  bool fasta = a.has("fasta");
  std::string pfx = a.str("out", "synth");
  // group alleles by gene (file order)
  std::vector<std::string> genes;
  std::map<std::string, std::vector<int>> byGene;
  for (int i = 0; i < (int)ref.size(); ++i) {
    if (!byGene.count(ref[i].gene)) genes.push_back(ref[i].gene);
    byGene[ref[i].gene].push_back(i);
  }
  // sample: 2 alleles per gene (10 % homozygous), expression log-uniform over 2 decades
  struct Hap { int allele; double w; std::string genome; };
  std::vector<Hap> haps;
  FILE *ft = fopen((pfx + "_truth.tsv").c_str(), "w");
  for (auto &g : genes) {
    auto &v = byGene[g];
    int a1 = v[r.below((int)v.size())];
    int a2 = r.uni() < 0.10 ? a1 : v[r.below((int)v.size())];
    double expr = std::pow(10.0, 2.0 * r.uni());
    for (int al : {a1, a2}) {
      Hap h; h.allele = al; h.w = expr;
      // 'N' separators stand for unsequenced intron middles: a real fragment carries real intron there
      const std::string &s = ref[al].seq;
      for (char c : s) { if (c == 'N') h.genome += randomSeq(r, r.range(300, 900)); else h.genome += c; }
      haps.push_back(h);
    }
    fprintf(ft, "%s\t%s\t%s\t%.4f\n", g.c_str(), ref[a1].name.c_str(), ref[a2].name.c_str(), expr);
  }
  fclose(ft);
  std::vector<double> cum;
  double tot = 0;
  for (auto &h : haps) { tot += h.w * h.genome.size(); cum.push_back(tot); }
  FILE *f1 = fopen((pfx + (fasta ? "_1.fa" : "_1.fq")).c_str(), "w");
  FILE *f2 = fopen((pfx + (fasta ? "_2.fa" : "_2.fq")).c_str(), "w");
  FILE *fb = nBarcodes > 0 ? fopen((pfx + "_bc.fa").c_str(), "w") : NULL;
  std::vector<std::string> barcodes;
  for (int i = 0; i < nBarcodes; ++i) barcodes.push_back(randomSeq(r, 16));
  std::string qual(L, 'I');
  auto sequencing = [&](std::string s) {
    std::string o;
    for (size_t i = 0; i < s.size(); ++i) {
      double u = r.uni();
      if (u < indelR) { if (r.uni() < 0.5) continue; o += NUC[r.below(4)]; }
      char c = s[i];
      if (r.uni() < subR) c = otherBase(r, c);
      if (r.uni() < nR) c = 'N';
      o += c;
    }
    if ((int)o.size() > L) o.resize(L);
    while ((int)o.size() < L) o += NUC[r.below(4)];
    return o;
  };
  for (long i = 0; i < F; ++i) {
    std::string m1, m2;
    if (r.uni() < bgR) {
      m1 = randomSeq(r, L); m2 = randomSeq(r, L);
    } else {
      double u = r.uni() * tot;
      int hi = (int)(std::lower_bound(cum.begin(), cum.end(), u) - cum.begin());
      if (hi >= (int)haps.size()) hi = (int)haps.size() - 1;
      const std::string &g = haps[hi].genome;
      int flen = (int)(fragMean + fragSd * r.normal() + 0.5);
      if (flen < L) flen = L;
      if (flen > (int)g.size()) flen = (int)g.size();
      int st = r.below((int)g.size() - flen + 1);
      std::string frag = g.substr(st, flen);
      std::string e1 = frag.substr(0, std::min(L + 4, flen));
      std::string e2 = revcomp(frag).substr(0, std::min(L + 4, flen));
      m1 = sequencing(e1); m2 = sequencing(e2);
      if (r.uni() < 0.5) std::swap(m1, m2);
    }
    if (fasta) {
      fprintf(f1, ">r%ld/1\n%s\n", i, m1.c_str());
      fprintf(f2, ">r%ld/2\n%s\n", i, m2.c_str());
    } else {
      fprintf(f1, "@r%ld/1\n%s\n+\n%s\n", i, m1.c_str(), qual.c_str());
      fprintf(f2, "@r%ld/2\n%s\n+\n%s\n", i, m2.c_str(), qual.c_str());
    }
    if (fb) {
      // Zipf(s=1)-like barcode choice; 0.5 % missing
      if (r.uni() < 0.005) fprintf(fb, ">r%ld\nmissing_barcode\n", i);
      else {
        int b = (int)(std::pow((double)nBarcodes, r.uni())) - 1;
        if (b < 0) b = 0; if (b >= nBarcodes) b = nBarcodes - 1;
        fprintf(fb, ">r%ld\n%s\n", i, barcodes[b].c_str());
      }
    }
  }
  fclose(f1); fclose(f2);
  if (fb) fclose(fb);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: t1k_synth ref-rna|ref-dna|reads [--key value ...]\n"); return 1; }
  Args a;
  for (int i = 2; i < argc; ++i) {
    if (strncmp(argv[i], "--", 2) == 0) {
      std::string k = argv[i] + 2;
      if (i + 1 < argc && strncmp(argv[i + 1], "--", 2) != 0) { a.kv[k] = argv[i + 1]; ++i; }
      else a.kv[k] = "1";
    }
  }
  std::string cmd = argv[1];
  if (cmd == "ref-rna") return cmdRefRna(a);
  if (cmd == "ref-dna") return cmdRefDna(a);
  if (cmd == "reads") return cmdReads(a);
  fprintf(stderr, "unknown command %s\n", cmd.c_str());
  return 1;
}
