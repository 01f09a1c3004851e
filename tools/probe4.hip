// What does a wave's MFMA burst cost the OTHER wave of its SIMD?  8-wave workgroup (waves w and w+4 share a SIMD):
// waves 0-3 issue back-to-back MFMAs (fp32 32x32x2, bf16 32x32x16, or nothing); waves 4-7 run a fixed amount of
// non-MFMA work (LDS reads / LDS-DMA / VALU) and report their own elapsed cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MF, int WORK, int FLIP, int PRIO> __global__ __launch_bounds__(512) void k(float* out, const float* src, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  int w = __builtin_amdgcn_readfirstlane(t >> 6);
  if (FLIP) w ^= 4;   // waves 4-7 take the MFMA role
  if (w >= 4 && PRIO) __builtin_amdgcn_s_setprio(PRIO);
  if (w < 4) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    if (MF == 1) {
      float a = 1.f + lane, b = 2.f;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
    } else if (MF == 2) {
      bf16x8 a, b;
      for (int j = 0; j < 8; ++j) a[j] = (__bf16)(1.f + lane), b[j] = (__bf16)2.f;
      for (int it = 0; it < 2 * iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + t] = s;
    return;
  }
  char* ring = smem + (w - 4) * 16384;
  const unsigned ad = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 8;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float accv = 0.f;
  const int n = iters / 8;
  if (WORK == 0) {   // 8 ds_read_b64 + wait, n times
    for (int it = 0; it < n; ++it) {
      f32x2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(ad) : "memory");
      accv += v0[0] + v7[1];
    }
  } else if (WORK == 1) {   // 4 LDS-DMA of 1 KiB (L2-resident source) + vmcnt(0), n times
    const char* s0 = (const char*)src + (blockIdx.x & 63) * 65536 + lane * 16;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s0 + ((it * 4 + q) & 15) * 1024),
                                         (__attribute__((address_space(3))) void*)(ring + q * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {   // 64 dependent VALU adds, n times
    float x = lane;
    for (int it = 0; it < n; ++it)
#pragma unroll
      for (int u = 0; u < 64; ++u) x = x * 1.0001f + 0.5f;
    accv = x;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + (w - 4)] = t1 - t0;
  out[blockIdx.x * 512 + t] = accv;
}

template <int MF, int WORK, int FLIP = 0, int PRIO = 0> static void run(const char* name, float* out, const float* src, unsigned long long* cyc) {
  const int iters = 400;
  (void)hipFuncSetAttribute((const void*)k<MF, WORK, FLIP, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<MF, WORK, FLIP, PRIO><<<256, 512, 65536>>>(out, src, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
  printf("%-34s partner cycles per unit of work: %8.0f\n", name, s / 1024 / (iters / 8));
}
int main() {
  float *out, *src; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&src, 64 * 65536 + 65536); (void)hipMalloc(&cyc, 1024 * 8);
  (void)hipMemset(src, 0, 64 * 65536 + 65536);
  run<0, 0>("LDS reads    | partner idle", out, src, cyc);
  run<1, 0>("LDS reads    | partner f32 MFMA", out, src, cyc);
  run<2, 0>("LDS reads    | partner bf16 MFMA", out, src, cyc);
  run<0, 1>("LDS-DMA      | partner idle", out, src, cyc);
  run<1, 1>("LDS-DMA      | partner f32 MFMA", out, src, cyc);
  run<2, 1>("LDS-DMA      | partner bf16 MFMA", out, src, cyc);
  run<0, 2>("64 VALU fma  | partner idle", out, src, cyc);
  run<1, 2>("64 VALU fma  | partner f32 MFMA", out, src, cyc);
  run<2, 2>("64 VALU fma  | partner bf16 MFMA", out, src, cyc);
  printf("-- roles flipped (waves 4-7 run the MFMAs, waves 0-3 the work)\n");
  run<1, 0, 1>("LDS reads    | partner f32 MFMA", out, src, cyc);
  run<1, 1, 1>("LDS-DMA      | partner f32 MFMA", out, src, cyc);
  run<2, 1, 1>("LDS-DMA      | partner bf16 MFMA", out, src, cyc);
  run<1, 2, 1>("64 VALU fma  | partner f32 MFMA", out, src, cyc);
  printf("-- worker waves at s_setprio 3\n");
  run<1, 0, 0, 3>("LDS reads    | partner f32 MFMA", out, src, cyc);
  run<1, 1, 0, 3>("LDS-DMA      | partner f32 MFMA", out, src, cyc);
  run<2, 1, 0, 3>("LDS-DMA      | partner bf16 MFMA", out, src, cyc);
  run<1, 2, 0, 3>("64 VALU fma  | partner f32 MFMA", out, src, cyc);
  printf("-- roles flipped + worker waves at s_setprio 3\n");
  run<1, 0, 1, 3>("LDS reads    | partner f32 MFMA", out, src, cyc);
  run<1, 1, 1, 3>("LDS-DMA      | partner f32 MFMA", out, src, cyc);
  return 0;
}
