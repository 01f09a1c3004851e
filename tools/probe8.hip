// probe8: HBM READ bandwidth of the skinny-TN access pattern vs the width of the contiguous piece a workgroup reads per
// token row.  256 resident workgroups (one per CU) x 8 waves; workgroup (cg, slab) streams columns [cg*CW, (cg+1)*CW) of
// the token rows of its slab of a [T, D] bf16 matrix through wave-private LDS-DMA rings (4 stages of 4 KiB per wave, as
// tn_partial_dma_wide_kernel does with 6-KiB stages), optionally together with the 128-byte rows of a [T, 64] side
// matrix S that every column group re-reads (h / dh).  Nothing is computed: this is the ceiling of the read stream.
//   order 0: block b -> cg = b / NS (all slabs of a column group are neighbours)   [what the kernel does]
//   order 1: block b -> cg = b % NCG (the column groups of a slab are neighbours)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int WAVES = 8, STAGE = 2048;   // M bytes per wave stage; ring depth 8 (M only) or 4 (M + S): 128 KiB of LDS

__device__ __forceinline__ void dma16(const void* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_newer(int newer) {   // newer stages of N instructions each
  switch (newer) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * N) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * N) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * N) : "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * N) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * N) : "memory"); break;
  }
}

// CWB = bytes per row piece (128 .. 2048); rows per stage = 2048 / CWB; 2 DMA instructions per stage for M, 1-2 for S
template <int CWB, bool WITH_S> __global__ __launch_bounds__(64 * WAVES, 1) void rd(const uint16_t* M, const uint16_t* S, int T, int D,
                                                                                     int NS, int NCG, int order, int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = STAGE / CWB;          // token rows per stage
  constexpr int CPR = CWB / 16;              // 16-byte chunks per row
  constexpr int NI = WITH_S ? 2 + (ROWS * 128 + 1023) / 1024 : 2;
  constexpr int DEPTH = WITH_S ? 4 : 8;
  constexpr int SLOT = WITH_S ? 4096 : 2048;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  char* ring = smem + w * (DEPTH * SLOT);
  const int slab_len = T / NS;
  for (int b = blockIdx.x; b < total; b += gridDim.x) {
    const int cg = order ? b % NCG : b / NS, slab = order ? b / NCG : b % NS;
    const int64_t t0 = (int64_t)slab * slab_len;
    const int ngroups = slab_len / ROWS, nw = (ngroups - w + WAVES - 1) / WAVES;
    auto issue = [&](int i) {
      const int64_t tt = t0 + (int64_t)(w + WAVES * i) * ROWS;
      char* slot = ring + (i % DEPTH) * SLOT;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int idx = q * 64 + lane, row = idx / CPR, c = idx % CPR;
        dma16((const char*)(M + (tt + row) * D) + (int64_t)cg * CWB + c * 16, slot + q * 1024);
      }
      if (WITH_S) {
#pragma unroll
        for (int q = 0; q < NI - 2; ++q) {
          const int idx = q * 64 + lane, row = idx >> 3, c = idx & 7;
          if (row < ROWS) dma16((const char*)(S + (tt + row) * 64) + c * 16, slot + STAGE + q * 1024);
        }
      }
    };
    const int pre = nw < DEPTH ? nw : DEPTH;
    for (int i = 0; i < pre; ++i) issue(i);
    for (int i = 0; i < nw; ++i) {
      const int newer = (nw - 1 - i) < (DEPTH - 1) ? (nw - 1 - i) : (DEPTH - 1);
      wait_newer<NI>(newer);
      if (i + DEPTH < nw) issue(i + DEPTH);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <int CWB, bool WS> float run(const uint16_t* const* M, const uint16_t* const* S, int NB, int T, int D, int NS, int order) {
  const int NCG = D * 2 / CWB, total = NCG * NS;
  const int lds = 128 * 1024;
  hipFuncSetAttribute((const void*)rd<CWB, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  auto go = [&]() {
    for (int i = 0; i < NB; ++i)
      hipLaunchKernelGGL((rd<CWB, WS>), dim3(total < 256 ? total : 256), dim3(64 * WAVES), lds, 0, M[i], S[i], T, D, NS, NCG, order, total);
  };
  go();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 5; ++rep) go();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (5 * NB);
}

int main() {
  const int T = 32768, NB = 8;
  for (int D : {512, 1024, 1376 + 32 /* 1408: a multiple of 128 near the ffn width */, 2048}) {
    uint16_t *M[NB], *S[NB];
    for (int i = 0; i < NB; ++i) {
      hipMalloc(&M[i], (size_t)T * D * 2 + 4096);
      hipMalloc(&S[i], (size_t)T * 128 + 4096);
      hipMemset(M[i], 0, (size_t)T * D * 2);
      hipMemset(S[i], 0, (size_t)T * 128);
    }
    const double mb = (double)T * D * 2 / 1e6;
    for (int ws = 0; ws < 2; ++ws)
      for (int order = 0; order < 2; ++order) {
        for (int cwb : {128, 256, 512, 1024, 2048}) {
          if ((D * 2) % cwb) continue;
          const int ncg = D * 2 / cwb;
          int ns = 256 / ncg;   // one resident round
          if (ns < 1) ns = 1;
          while (T % ns || (T / ns) % (STAGE / cwb)) --ns;
          float us = 0;
#define RUN(C)                                                                    \
  us = ws ? run<C, true>(M, S, NB, T, D, ns, order) : run<C, false>(M, S, NB, T, D, ns, order)
          if (cwb == 128) RUN(128);
          else if (cwb == 256) RUN(256);
          else if (cwb == 512) RUN(512);
          else if (cwb == 1024) RUN(1024);
          else RUN(2048);
          const double smb = ws ? (double)T * 128 * ncg / 1e6 : 0.0;
          printf("D=%5d S=%d order=%d piece=%4d B (%2d col groups x %3d slabs): %7.1f us  M %6.0f GB/s  M+S(re-read) %6.0f GB/s\n", D, ws,
                 order, cwb, ncg, ns, us, mb / us * 1e3, (mb + smb) / us * 1e3);
          fflush(stdout);
        }
      }
    for (int i = 0; i < NB; ++i) hipFree(M[i]), hipFree(S[i]);
  }
  return 0;
}
