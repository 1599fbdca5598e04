// Issue rate of a few VALU instructions on gfx950, measured: every wave runs ITER x 16 instances of one instruction on 8 independent
// register chains (so latency hides), 16 waves per CU on every CU; cycles per wave-level instruction per SIMD = elapsed SIMD cycles /
// (instructions per wave x waves per SIMD).      hipcc --offload-arch=gfx950 -O2 tools/calib/inst_rate.hip -o tools/calib/inst_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, TYPE, INIT, ASM)                                                       \
  __global__ __launch_bounds__(256) void NAME(TYPE *out, TYPE seed) {                       \
    TYPE a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    TYPE b = INIT;                                                                          \
    for (int i = 0; i < ITER; ++i) {                                                        \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));                \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;     \
  }
#define A_ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define A_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
#define A_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define A_FMA32(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n"
#define A_FMA64(k) "v_fma_f64 %" #k ", %" #k ", %8, %8\n"
#define A_MUL64(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define A_ADD64(k) "v_add_f64 %" #k ", %" #k ", %8\n"
#define A_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %8\n"
#define A_MAD64(k) "v_mad_u64_u32 %" #k ", vcc, %8, %8, %" #k "\n"
#define A_MIN32(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define A_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define A_RCP64(k) "v_rcp_f64 %" #k ", %" #k "\n"
#define A_SQRT64(k) "v_sqrt_f64 %" #k ", %" #k "\n"
#define A_CVT(k) "v_cvt_f64_f32 %" #k ", %8\n"
KERNEL(k_add, uint32_t, 3u, A_ADD)
KERNEL(k_mullo, uint32_t, 3u, A_MULLO)
KERNEL(k_mulhi, uint32_t, 3u, A_MULHI)
KERNEL(k_xor, uint32_t, 3u, A_XOR)
KERNEL(k_fma32, float, 1.0001f, A_FMA32)
KERNEL(k_min32, float, 1.0001f, A_MIN32)
KERNEL(k_fma64, double, 1.0001, A_FMA64)
KERNEL(k_mul64, double, 1.0001, A_MUL64)
KERNEL(k_add64, double, 1.0001, A_ADD64)
KERNEL(k_pkfma, double, 1.0001, A_PKFMA)
__global__ __launch_bounds__(256) void k_mad64(uint64_t *out, uint64_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t b = 3u + threadIdx.x;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(A_MAD64(0) A_MAD64(1) A_MAD64(2) A_MAD64(3) A_MAD64(4) A_MAD64(5) A_MAD64(6) A_MAD64(7) A_MAD64(0) A_MAD64(1) A_MAD64(2) A_MAD64(3) A_MAD64(4) A_MAD64(5) A_MAD64(6) A_MAD64(7)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
KERNEL(k_rcp64, double, 1.0001, A_RCP64)
KERNEL(k_sqrt64, double, 1.0001, A_SQRT64)

template <typename T, typename K> void run(const char *name, K kern, int cus, double ghz) {
  T *out;
  const int blocks = cus * 4, threads = 256;            // 4 blocks of 4 waves per CU = 16 waves per CU = 4 per SIMD
  if (hipMalloc(&out, sizeof(T) * blocks * threads) != hipSuccess) { printf("hipMalloc failed\n"); return; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, (T)1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, (T)1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_wave = (double)ITER * 16, waves_per_simd = 4;
  printf("%-14s %8.3f ms  -> %5.2f cycles per wave-level instruction per SIMD (at %.2f GHz)\n", name, ms, ms * 1e-3 * ghz * 1e9 / (inst_per_wave * waves_per_simd), ghz);
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("%s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
  run<uint32_t>("v_add_u32", k_add, cus, ghz);
  run<uint32_t>("v_xor_b32", k_xor, cus, ghz);
  run<uint32_t>("v_mul_lo_u32", k_mullo, cus, ghz);
  run<uint32_t>("v_mul_hi_u32", k_mulhi, cus, ghz);
  run<uint64_t>("v_mad_u64_u32", k_mad64, cus, ghz);
  run<float>("v_fma_f32", k_fma32, cus, ghz);
  run<float>("v_min_f32", k_min32, cus, ghz);
  run<double>("v_pk_fma_f32", k_pkfma, cus, ghz);
  run<double>("v_fma_f64", k_fma64, cus, ghz);
  run<double>("v_mul_f64", k_mul64, cus, ghz);
  run<double>("v_add_f64", k_add64, cus, ghz);
  run<double>("v_rcp_f64", k_rcp64, cus, ghz);
  run<double>("v_sqrt_f64", k_sqrt64, cus, ghz);
  return 0;
}
