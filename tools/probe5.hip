// LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) cost as a function of the source shape: the same
// 1 KiB as 1 x 1024 B, 4 x 256 B, 8 x 128 B or 16 x 64 B row pieces (row pitch 4 KiB, L2-resident source).
// 8 waves per CU (2 per SIMD), every wave streams pieces back to back with 4 in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ROWS> __global__ __launch_bounds__(512) void k(const char* src, unsigned long long* cyc, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  char* ring = smem + w * 8192;
  constexpr int PIECE = 1024 / ROWS;          // bytes per row piece
  constexpr int LPR = PIECE / 16;             // lanes per row
  const int row = lane / LPR, c = lane % LPR;
  const char* base = src + (size_t)(blockIdx.x & 31) * (1 << 20) + (size_t)w * 65536 + (size_t)row * 4096 + c * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const char* s = base + (size_t)((it & 7) * PIECE);   // walk along the rows, stay L2-resident
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                     (__attribute__((address_space(3))) void*)(ring + (it & 7) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
  sink[blockIdx.x * 512 + t] = ((float*)ring)[lane];
}
template <int ROWS> static void run(const char* src, unsigned long long* cyc, float* sink) {
  const int iters = 2000;
  (void)hipFuncSetAttribute((const void*)k<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<ROWS><<<256, 512, 65536>>>(src, cyc, 50, sink);
  k<ROWS><<<256, 512, 65536>>>(src, cyc, iters, sink);
  (void)hipDeviceSynchronize();
  static unsigned long long h[2048];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
  const double per = s / 2048 / iters;
  printf("1 KiB as %2d x %4d B: %6.1f cycles per DMA per wave (8 waves/CU) -> %5.1f B/clk/CU\n", ROWS, 1024 / ROWS, per, 8 * 1024.0 / per);
}
int main() {
  char* src; unsigned long long* cyc; float* sink;
  (void)hipMalloc(&src, 33u << 20); (void)hipMemset(src, 0, 33u << 20);
  (void)hipMalloc(&cyc, 2048 * 8); (void)hipMalloc(&sink, 256 * 512 * 4);
  run<1>(src, cyc, sink); run<4>(src, cyc, sink); run<8>(src, cyc, sink); run<16>(src, cyc, sink);
  return 0;
}
