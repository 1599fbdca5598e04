// Calibrates rocprofv3's FETCH_SIZE for THIS kernel's access pattern (MI355X_MICROARCH.md, HBM section: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern"): every lane gathers one 64-byte record (4 x 16 B, like a
// DNode<float>) at a pseudo-random index of a table far larger than L2 + Infinity Cache, so every record is a miss all the way to DRAM.
//   ./fetch_calib [table GiB] [records per lane]      known bytes = lanes x records x 64 (x 2 if the L2 fills whole 128-byte lines)
// Prints the record rate; run under `rocprofv3 --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum) for the counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_gather64(const float4 *table, unsigned long long n_rec, int per_lane, float *out) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  float acc = 0.f;
  for (int k = 0; k < per_lane; ++k) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;                 // a different record per lane and step
    const float4 *p = table + 4 * (x % n_rec);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.z + d.w;
  }
  out[i] = acc;
}
// The same gather with 128-byte records (two adjacent 64-byte halves of one 128-byte-aligned line): if the memory-side request
// count per record doubles, requests are 64-byte sectors and a 64-byte gather moves 64 bytes; if it stays 1, a request is a whole
// 128-byte line tallied at 64 bytes (the guide's streaming-read finding) and every count has to be doubled.
__global__ void k_gather128(const float4 *table, unsigned long long n_rec, int per_lane, float *out) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  float acc = 0.f;
  for (int k = 0; k < per_lane; ++k) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const float4 *p = table + 8 * (x % (n_rec / 2));
    const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
    acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
  }
  out[i] = acc;
}
int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  const int per_lane = argc > 2 ? atoi(argv[2]) : 16;
  const unsigned long long n_rec = (unsigned long long)(gib * (1ull << 30)) / 64;
  const int blocks = 256 * 16, threads = 256;
  float4 *table; float *out;
  CHECK(hipMalloc((void **)&table, n_rec * 64));
  CHECK(hipMemset(table, 0, n_rec * 64));
  CHECK(hipMalloc((void **)&out, (size_t)blocks * threads * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(threads), 0, 0, table, n_rec, per_lane, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * threads * per_lane;
    printf("table %.1f GiB, %.0f records of 64 B per launch (%.3f GB at 64 B each): %.3f ms = %.1f GB/s counted at 64 B per record\n", gib, recs, recs * 64 / 1e9, ms, recs * 64 / 1e9 / (ms * 1e-3));
  }
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather128, dim3(blocks), dim3(threads), 0, 0, table, n_rec, per_lane, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * threads * per_lane;
    printf("table %.1f GiB, %.0f records of 128 B per launch (%.3f GB): %.3f ms = %.1f GB/s\n", gib, recs, recs * 128 / 1e9, ms, recs * 128 / 1e9 / (ms * 1e-3));
  }
  return 0;
}
