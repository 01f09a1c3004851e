// fp32 / bf16 MFMA peak probe: N back-to-back MFMAs per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int NACC> __global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float av = a + threadIdx.x, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> static void run(const char* name, int wgs_per_cu) {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const int iters = 2000, blocks = 256 * wgs_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k_f32<NACC><<<blocks, 256>>>(out, 10, 1.f, 2.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k_f32<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("%s: %d WG/CU (4 waves each), NACC=%d: %.3f ms, %.1f TFLOP/s\n", name, wgs_per_cu, NACC, ms, flops / ms / 1e9);
  (void)hipFree(out);
}
int main() {
  run<4>("f32 32x32x2", 1);
  run<4>("f32 32x32x2", 2);
  run<1>("f32 32x32x2 dependent", 1);
  run<1>("f32 32x32x2 dependent", 2);
  run<2>("f32 32x32x2", 2);
  return 0;
}
