// tools/fetch_calib.hip -- calibration of the FETCH_SIZE / WRITE_SIZE counters for 8-byte loads (k_seed_groups reads its postings as
// 64 consecutive 8-byte words per wavefront load).  MI355X_MICROARCH.md prescribes a x2 correction of FETCH_SIZE on gfx950, established
// with 16-byte-per-lane streams; this streams a KNOWN number of bytes once with 8-byte (and, for comparison, 16-byte) loads per lane, so
// the factor can be read off the counter for the access width the seeding kernel uses:
//   rocprofv3 --pmc FETCH_SIZE -- tools/fetch_calib          factor = bytes streamed / (FETCH_SIZE x 1024)   (profiles/r03_fetch_calibration.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <class T>
__global__ void k_stream(const T *p, size_t n, unsigned long long *out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = p[i];  // consecutive lanes, consecutive elements: 64 x sizeof(T) contiguous bytes per wavefront load
    const unsigned long long *w = (const unsigned long long *)&v;
    for (unsigned k = 0; k < sizeof(T) / 8; ++k) acc += w[k];
  }
  if (acc == 0x1234567ull) out[0] = acc;  // (keeps the loads alive)
}
int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 4096) << 20;  // MiB, default 4 GiB: far beyond the 256 MiB Infinity Cache
  void *buf; unsigned long long *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&out, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms8 = 0, ms16 = 0;
  hipEventRecord(a); hipLaunchKernelGGL(k_stream<unsigned long long>, dim3(256 * 32), dim3(256), 0, 0, (const unsigned long long *)buf, bytes / 8, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms8, a, b);
  hipEventRecord(a); hipLaunchKernelGGL(k_stream<ulonglong2>, dim3(256 * 32), dim3(256), 0, 0, (const ulonglong2 *)buf, bytes / 16, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms16, a, b);
  printf("{\"bytes_per_kernel\": %zu, \"k_stream_u64_ms\": %.3f, \"k_stream_u64_GBs\": %.1f, \"k_stream_u128_ms\": %.3f, \"k_stream_u128_GBs\": %.1f}\n", bytes, ms8, bytes / ms8 / 1e6, ms16, bytes / ms16 / 1e6);
  return 0;
}
