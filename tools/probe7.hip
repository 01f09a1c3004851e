// probe7: HBM write bandwidth vs the width of the contiguous piece each row receives per step.
// 512 resident workgroups (2 per CU) x 256 threads; workgroup b owns token rows 64 b .. 64 b + 63 of a [T, D] bf16
// matrix (like one block of the chain kernel) and writes it column slice by column slice, W columns (2 W bytes per row)
// at a time -- W = 64 is what one phase-2 step of chain2_kernel writes.  8 rotating buffers, hipEvents.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int NT> __global__ __launch_bounds__(256) void wr(uint16_t* Y, int T, int D, int W) {
  const int per_row = W / 8;
  const u32x4 v = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  for (int blk = blockIdx.x; blk < T / 64; blk += gridDim.x) {
    const int64_t row0 = (int64_t)blk * 64;
    for (int c0 = 0; c0 < D; c0 += W) {
      const int w = D - c0 < W ? D - c0 : W;
      const int pr = w / 8;
      for (int idx = threadIdx.x; idx < 64 * pr; idx += 256) {
        const int r = idx / pr, c = idx % pr;
        uint16_t* dst = Y + (row0 + r) * D + c0 + c * 8;
        if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
        else *(u32x4*)dst = v;
      }
    }
  }
  (void)per_row;
}

int main() {
  const int T = 32768, NB = 8;
  for (int D : {512, 1376, 2048}) {
    uint16_t* buf[NB];
    for (int i = 0; i < NB; ++i) hipMalloc(&buf[i], (size_t)T * D * 2);
    for (int nt = 0; nt < 2; ++nt)
      for (int W : {64, 128, 256, 512, 4096}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        auto run = [&]() {
          for (int i = 0; i < NB; ++i) {
            if (nt) hipLaunchKernelGGL(wr<1>, dim3(512), dim3(256), 0, 0, buf[i], T, D, W);
            else hipLaunchKernelGGL(wr<0>, dim3(512), dim3(256), 0, 0, buf[i], T, D, W);
          }
        };
        run();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 5; ++rep) run();
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (5 * NB), mb = (double)T * D * 2 / 1e6;
        printf("D=%5d nt=%d W=%4d: %7.1f us  %6.0f GB/s\n", D, nt, W, us, mb / us * 1e3);
      }
    for (int i = 0; i < NB; ++i) hipFree(buf[i]);
  }
  return 0;
}
