// What one LDS-DMA instruction costs the wave that issues it, and what it costs that wave's MFMA stream.
// One wave per SIMD (256 threads, 1 workgroup per CU).  Each iteration = NM independent 32x32x16 bf16 MFMAs + ND DMAs
// (1 KiB each, L2-resident source), 4 DMAs kept in flight.  Address forms of the DMA:
//   V = global_load_lds_dwordx4 with a 64-bit VGPR address pair
//   S = global_load_lds_dwordx4 with an SGPR base + 32-bit VGPR offset
//   B = buffer_load_dwordx4 ... lds (buffer resource + 32-bit VGPR offset)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int FORM, int NM, int ND> __global__ __launch_bounds__(256, 1) void k(const char* src, unsigned long long* cyc, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  char* ring = smem + w * 8192;
  const char* ubase = src + (size_t)(blockIdx.x & 31) * (1 << 20) + (size_t)w * 65536;   // wave-uniform
  const uint32_t voff = (uint32_t)((lane >> 2) * 4096 + (lane & 3) * 16);               // 16 rows x 64 B
  const char* vptr = ubase + voff;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ubase, 0, 1 << 20, 0x00020000);
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) fa[i] = (__bf16)(float)(lane + i), fb[i] = (__bf16)(float)(lane - i);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const uint32_t step = (uint32_t)(((it * ND + d) & 7) * 64);
      char* dst = ring + ((it * ND + d) & 7) * 1024;
      if constexpr (FORM == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vptr + step),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      } else if constexpr (FORM == 1) {
        const char* p = ubase + (size_t)(uint32_t)(voff + step);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)(voff + step), 0, 0, 0);
      }
    }
    if (ND > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][lane & 15];
  sink[blockIdx.x * 256 + t] = s + ((float*)ring)[lane];
}
template <int FORM, int NM, int ND> static void run(const char* src, unsigned long long* cyc, float* sink, const char* name) {
  const int iters = 2000;
  (void)hipFuncSetAttribute((const void*)k<FORM, NM, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  k<FORM, NM, ND><<<256, 256, 32768>>>(src, cyc, 50, sink);
  k<FORM, NM, ND><<<256, 256, 32768>>>(src, cyc, iters, sink);
  (void)hipDeviceSynchronize();
  static unsigned long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
  printf("%s  %2d MFMA + %d DMA per iteration: %7.1f cycles/iteration\n", name, NM, ND, s / 1024 / iters);
}
int main() {
  char* src; unsigned long long* cyc; float* sink;
  (void)hipMalloc(&src, 33u << 20); (void)hipMemset(src, 0, 33u << 20);
  (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&sink, 256 * 256 * 4);
  run<0, 4, 0>(src, cyc, sink, "-");
  run<0, 0, 1>(src, cyc, sink, "V"); run<1, 0, 1>(src, cyc, sink, "S"); run<2, 0, 1>(src, cyc, sink, "B");
  run<0, 4, 1>(src, cyc, sink, "V"); run<1, 4, 1>(src, cyc, sink, "S"); run<2, 4, 1>(src, cyc, sink, "B");
  run<0, 8, 1>(src, cyc, sink, "V"); run<1, 8, 1>(src, cyc, sink, "S"); run<2, 8, 1>(src, cyc, sink, "B");
  run<0, 8, 2>(src, cyc, sink, "V"); run<2, 8, 2>(src, cyc, sink, "B");
  return 0;
}
