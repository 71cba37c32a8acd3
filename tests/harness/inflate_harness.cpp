// tests/harness/inflate_harness.cpp -- command-line front of the product's gzip decoder (t1k_amd/csrc/host/inflate.cpp) for the CPU tests:
//   inflate_harness in.gz out.bin [reps]   -> writes the text, prints "<bytes> <members> <crc of last member, hex> <MB/s of the best repetition>"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../t1k_amd/csrc/host/t1k_host.h"
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) return 2;
  std::vector<uint8_t> src;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) src.insert(src.end(), buf, buf + n);
  fclose(fp);
  const size_t cap = argc > 4 ? (size_t)atoll(argv[4]) : src.size() * 1200 + (1 << 20);
  std::vector<uint8_t> dst(cap + 16);
  const int reps = argc > 3 ? atoi(argv[3]) : 1;
  size_t outLen = 0, members = 0;
  uint32_t crc = 0;
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    t1k::GzProgress pg;
    std::string err;
    auto t0 = std::chrono::steady_clock::now();
    const int rc = t1k::gzInflateAll(src.data(), src.size(), dst.data(), cap, &pg, &outLen, &crc, &members, err);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) { printf("ERROR %s\n", err.c_str()); return 1; }
    if (pg.produced.load() != outLen || pg.state.load() != 1) { printf("ERROR progress %llu of %zu, state %d\n", (unsigned long long)pg.produced.load(), outLen, pg.state.load()); return 1; }
    best = s < best ? s : best;
  }
  fp = fopen(argv[2], "wb");
  if (!fp || fwrite(dst.data(), 1, outLen, fp) != outLen) return 2;
  fclose(fp);
  printf("%zu %zu %08x %.0f\n", outLen, members, crc, outLen / best / 1e6);
  return 0;
}
