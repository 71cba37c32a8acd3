// smallest HIP process: what a fresh process pays before and after any work of its own (tools/cold_r05.sh).
//   hip_hello [GB [free]]   allocates and touches that much device memory in 1.5 GB blocks first (and hands it back before leaving with `free`):
//   what does the end of a process cost per GB of device memory it held?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include <sys/mman.h>
__global__ void k(int *p) { *p = 1; }
int main(int argc, char **argv) {
  auto t0 = std::chrono::steady_clock::now();
  int *d = nullptr;
  (void)hipSetDevice(0);
  (void)hipMalloc(&d, 1 << 20);
  auto t1 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
  (void)hipDeviceSynchronize();
  auto t2 = std::chrono::steady_clock::now();
  const double gb = argc > 1 ? atof(argv[1]) : 0;
  std::vector<void *> blocks;
  for (double have = 0; have < gb; have += 1.5) {
    void *p = nullptr;
    if (hipMalloc(&p, (size_t)1536 << 20) != hipSuccess) break;
    (void)hipMemsetAsync(p, 1, (size_t)1536 << 20, 0);
    blocks.push_back(p);
  }
  (void)hipDeviceSynchronize();
  // HH_STREAMS=n: a kernel on n streams of its own (what does the end of a process cost per hardware queue?); HH_ANON_GB=g: g GB of host memory touched
  // page by page (... per resident GB of its own address space?); HH_MAPS=1: the largest resident mappings at the end of main
  if (const char *e = getenv("HH_STREAMS")) {
    std::vector<hipStream_t> st(atoi(e));
    for (auto &x : st) { (void)hipStreamCreate(&x); hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, x, d); }
    (void)hipDeviceSynchronize();
  }
  if (const char *e = getenv("HH_ANON_GB")) {
    const size_t n = (size_t)(atof(e) * 1073741824.0);
    volatile char *m = (volatile char *)malloc(n);
    for (size_t i = 0; m && i < n; i += 4096) m[i] = 1;
    auto ta = std::chrono::steady_clock::now();
    if (const char *z = getenv("HH_ZAP")) {  // hand the pages back before leaving, from several threads (MADV_DONTNEED takes the memory-map lock for reading only)
      const int T = atoi(z) > 0 ? atoi(z) : 1;
      std::vector<std::thread> th;
      const size_t lo = ((size_t)m + 4095) & ~(size_t)4095, hi = ((size_t)m + n) & ~(size_t)4095, piece = ((hi - lo) / T) & ~(size_t)4095;
      for (int t = 0; t < T; ++t) th.emplace_back([=] { (void)madvise((void *)(lo + t * piece), t == T - 1 ? hi - (lo + t * piece) : piece, MADV_DONTNEED); });
      for (auto &x : th) x.join();
      fprintf(stderr, "  %.1f GB handed back by %d threads in %.1f ms\n", n / 1073741824.0, T, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count());
    }
  }
  if (getenv("HH_MAPS")) {
    FILE *fp = fopen("/proc/self/smaps", "r");
    char line[512], head[512] = "";
    while (fp && fgets(line, sizeof line, fp)) {
      unsigned long a, b;
      if (sscanf(line, "%lx-%lx ", &a, &b) == 2) strcpy(head, line);
      else if (!strncmp(line, "Rss:", 4) && strtoul(line + 4, nullptr, 10) >= 32768) fprintf(stderr, "  rss %s kB  %s", strtok(line + 4, " k\n"), head);
    }
    if (fp) fclose(fp);
  }
  auto t3 = std::chrono::steady_clock::now();
  if (argc > 2 && !strcmp(argv[2], "free")) for (void *p : blocks) (void)hipFree(p);
  auto t4 = std::chrono::steady_clock::now();
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  fprintf(stderr, "hip_hello: first hipMalloc %.1f ms, first kernel %.1f ms, %.1f GB allocated + set in %.1f ms, freed in %.1f ms, main %.1f ms\n", ms(t0, t1), ms(t1, t2), blocks.size() * 1.5,
          ms(t2, t3), ms(t3, t4), ms(t0, t4));
  return 0;
}
