// bf16 streaming GEMM, one wave per SIMD (gfx950) -- experimental successor of gemm2.hip, same contract:
//     C[M,N] = alpha * (A[M,K] . op(B) + A2[M,64] . op(B2)) + beta * C + bias[N]
//
// gemm2.hip runs 8 waves (two per SIMD) that all issue their DMAs, read their fragments and multiply in the same
// order; on a SIMD the matrix pipe and every other instruction's issue are exclusive BETWEEN waves (tools/probe4.hip),
// so the pipe idles ~50 % of a stage.  Inside ONE wave an MFMA runs asynchronously: the wave can issue LDS reads and
// DMAs while its previous MFMA occupies the pipe.  Here a workgroup is 4 waves (2 x 2), each owning a 128 x 128
// sub-tile (16 accumulator tiles = 256 registers, the AGPR half of the 512 a lone wave may use), and the k-loop is
// software-pipelined by hand: while the 16 MFMAs of one k-step run, the wave reads the 8 fragments of the next k-step
// and issues 4 of its 8 DMA instructions of a stage three ahead.  One raw s_barrier per 32-wide stage.
#include "kernels.hpp"
#include "epilogue.hpp"
#include "lds_dma.hpp"
#include <cstdlib>

namespace sow {

constexpr int G3_BM = 256, G3_BN = 256, G3_BK = 32;
constexpr int G3_THREADS = 256;
constexpr int G3_NSLOT = 4;
constexpr int G3_PIECE = 256 * G3_BK * 2;       // 16 KiB
constexpr int G3_STAGE = 2 * G3_PIECE;          // 32 KiB
constexpr int G3_LDS = G3_NSLOT * G3_STAGE;     // 128 KiB

struct Gemm3Params {
  const bf16_t* A;
  const bf16_t* B;
  const bf16_t* A2;   // [M, 64] or nullptr
  const bf16_t* B2;   // NT: [N, 64]; NN: [k2, N]
  bf16_t* C;
  const bf16_t* bias;
  int64_t M, lda, ldb, lda2, ldb2, ldc;
  int N, K, k2;
  float alpha, beta;
};

__device__ __forceinline__ void g3_vm_wait(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
  }
  __builtin_amdgcn_sched_barrier(0);
}

#define G3_SB() __builtin_amdgcn_sched_barrier(0)
#define G3_MF(i, FA, FB) \
  acc[(i) >> 2][(i)&3] = mfma32(as_bf16x8(FA[(i) >> 2]), as_bf16x8(FB[(i)&3]), acc[(i) >> 2][(i)&3])

template <bool NT> __global__ __launch_bounds__(G3_THREADS, 1) void gemm3_kernel(const Gemm3Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = (p.N + G3_BN - 1) / G3_BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(lid / tiles_n) * G3_BM;
  const int n0 = (lid % tiles_n) * G3_BN;
  const int K = p.K, N = p.N;
  const int64_t M = p.M;
  const int s_main = (K + G3_BK - 1) / G3_BK;
  const int S = s_main + (p.A2 ? 2 : 0);
  const char* zp = zero_page_for(lane);

  // ---------------------------------------------------------------- DMA: instruction q = 0..3 -> A rows, 4..7 -> B
  const int crow = 16 * (4 * w) + (lane >> 2);   // k-contiguous pieces: instruction 4w + ii covers rows 16(4w+ii) ..+15
  const int cpc = lane & 3;
  const int krow = 2 * (4 * w) + (lane >> 5);    // k-major B piece: instruction 4w + ii covers k rows 2(4w+ii), +1
  const int kpc = lane & 31;
  auto issue_slow = [&](int s, int q) {
    char* slot = smem + (s % G3_NSLOT) * G3_STAGE;
    const bool ext = s >= s_main;
    const int k0 = ext ? (s - s_main) * G3_BK : s * G3_BK;
    const int ii = q & 3;
    if (q < 4) {
      const bf16_t* Ap = ext ? p.A2 : p.A;
      const int64_t lda = ext ? p.lda2 : p.lda;
      const int klim = ext ? 64 : K;
      const int row = crow + 16 * ii;
      const int lc = cpc ^ ((row >> 2) & 3);
      const int64_t gr = m0 + row;
      const void* src = (gr < M && k0 + 8 * lc < klim) ? (const void*)(Ap + gr * lda + k0 + 8 * lc) : (const void*)zp;
      dma16(src, slot + (4 * w + ii) * 1024);
    } else {
      const bf16_t* Bp = ext ? p.B2 : p.B;
      const int64_t ldb = ext ? p.ldb2 : p.ldb;
      if constexpr (NT) {
        const int klim = ext ? 64 : K;
        const int row = crow + 16 * ii;
        const int lc = cpc ^ ((row >> 2) & 3);
        const int gn = n0 + row;
        const void* src = (gn < N && k0 + 8 * lc < klim) ? (const void*)(Bp + (int64_t)gn * ldb + k0 + 8 * lc) : (const void*)zp;
        dma16(src, slot + G3_PIECE + (4 * w + ii) * 1024);
      } else {
        const int krows = ext ? p.k2 : K;
        const int row = krow + 2 * ii;
        const int lc = kpc ^ ((row & 3) << 2);
        const int gk = k0 + row, gn = n0 + 8 * lc;
        const void* src = (gk < krows && gn < N) ? (const void*)(Bp + (int64_t)gk * ldb + gn) : (const void*)zp;
        dma16(src, slot + G3_PIECE + (4 * w + ii) * 1024);
      }
    }
  };

  // Fast path for the main stages before the last one: running per-lane source pointers (row validity folded in as
  // pointer = zero page, stride = 0), one 64-bit add per DMA -- the generic address arithmetic above costs ~30 VALU
  // instructions per DMA, which would sit between the MFMAs it is interleaved with.
  const char* pa[4];
  const char* pb[4];
  int sa[4], sbs[4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int row = crow + 16 * ii;
    const int lc = cpc ^ ((row >> 2) & 3);
    const int64_t gr = m0 + row;
    const bool va = gr < M;
    pa[ii] = va ? (const char*)(p.A + gr * p.lda + 8 * lc) : zp;
    sa[ii] = va ? 64 : 0;
    if constexpr (NT) {
      const int gn = n0 + row;
      const bool vb = gn < N;
      pb[ii] = vb ? (const char*)(p.B + (int64_t)gn * p.ldb + 8 * lc) : zp;
      sbs[ii] = vb ? 64 : 0;
    } else {
      const int rowk = krow + 2 * ii;
      const int lck = kpc ^ ((rowk & 3) << 2);
      const int gn = n0 + 8 * lck;
      const bool vb = gn < N;
      pb[ii] = vb ? (const char*)(p.B + (int64_t)rowk * p.ldb + gn) : zp;
      sbs[ii] = vb ? (int)(64 * p.ldb) : 0;
    }
  }
  auto issue_one = [&](int s, int q) {
    if (s >= s_main - 1) {
      issue_slow(s, q);
      return;
    }
    char* slot = smem + (s % G3_NSLOT) * G3_STAGE;
    const int ii = q & 3;
    if (q < 4) {
      dma16((const void*)pa[ii], slot + (4 * w + ii) * 1024);
      pa[ii] += sa[ii];
    } else {
      dma16((const void*)pb[ii], slot + G3_PIECE + (4 * w + ii) * 1024);
      pb[ii] += sbs[ii];
    }
  };

  // ---------------------------------------------------------------- fragment addresses (per lane)
  const uint32_t base = lds_addr(smem);
  const int fsw = (li >> 2) & 3;
  const uint32_t a_off = (uint32_t)((wm * 128 + li) * 64);              // + mi * 2048
  uint32_t b_off[4];
  if constexpr (NT) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b_off[ni] = (uint32_t)(G3_PIECE + (wn * 128 + ni * 32 + li) * 64);
  } else {
    const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3;
    const int r1 = 8 * (g >> 1) + q;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = wn * 128 + ni * 32 + 16 * (g & 1) + 4 * pp;
      b_off[ni] = (uint32_t)(G3_PIECE + r1 * 512 + (((col >> 3) ^ ((r1 & 3) << 2)) * 16) + (col & 7) * 2);
    }
  }
  const uint32_t ch0 = (uint32_t)(((0 + lh) ^ fsw) * 16), ch1 = (uint32_t)(((2 + lh) ^ fsw) * 16);

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // fragment reads of k-step ks of the stage in slot address sb: A tile mi / B tile ni
  u32x4 fa0[4], fb0[4], fa1[4], fb1[4];
  u32x2 tl[4], th[4];   // NN: halves of the transposed B reads, joined after the wait
#define G3_RD_A(dst, sb, ks, mi) DS_READ_B128(dst[mi], (sb) + a_off + ((ks) ? ch1 : ch0), (mi)*2048)
#define G3_RD_B(dst, sb, ks, ni)                                              \
  do {                                                                        \
    if constexpr (NT) {                                                       \
      DS_READ_B128(dst[ni], (sb) + b_off[ni] + ((ks) ? ch1 : ch0), 0);        \
    } else {                                                                  \
      if (ks) {                                                               \
        DS_READ_TR(tl[ni], (sb) + b_off[ni], 8192);                           \
        DS_READ_TR(th[ni], (sb) + b_off[ni], 8192 + 2048);                    \
      } else {                                                                \
        DS_READ_TR(tl[ni], (sb) + b_off[ni], 0);                              \
        DS_READ_TR(th[ni], (sb) + b_off[ni], 2048);                           \
      }                                                                       \
    }                                                                         \
  } while (0)
#define G3_JOIN(dst)                                                           \
  do {                                                                         \
    if constexpr (!NT) {                                                       \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) dst[ni] = join2(tl[ni], th[ni]); \
    }                                                                          \
  } while (0)

  // ---------------------------------------------------------------- prologue: stages 0..2 and half of stage 3
  for (int s = 0; s < 3 && s < S; ++s)
    for (int q = 0; q < 8; ++q) issue_one(s, q);
  if (3 < S)
    for (int q = 0; q < 4; ++q) issue_one(3, q);
  g3_vm_wait((1 < S ? 8 : 0) + (2 < S ? 8 : 0) + (3 < S ? 4 : 0));
  raw_barrier();   // stage 0 landed for everyone
  {
    G3_RD_A(fa0, base, 0, 0); G3_RD_A(fa0, base, 0, 1); G3_RD_A(fa0, base, 0, 2); G3_RD_A(fa0, base, 0, 3);
    G3_RD_B(fb0, base, 0, 0); G3_RD_B(fb0, base, 0, 1); G3_RD_B(fb0, base, 0, 2); G3_RD_B(fb0, base, 0, 3);
    LGKM_WAIT0();
    G3_JOIN(fb0);
  }

#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    const uint32_t sb = base + (uint32_t)((s % G3_NSLOT) * G3_STAGE);
    const uint32_t sn = base + (uint32_t)(((s + 1) % G3_NSLOT) * G3_STAGE);
    const bool d3 = s + 3 < S, d4 = s + 4 < S, nx = s + 1 < S;
    // ---- first k-step: multiply (s, 0); fetch (s, 1); second half of the DMAs of stage s+3 (slot of stage s-1,
    //      free since the barrier of stage s-1)
    G3_MF(0, fa0, fb0); G3_RD_A(fa1, sb, 1, 0); G3_SB();
    G3_MF(1, fa0, fb0); G3_RD_A(fa1, sb, 1, 1); G3_SB();
    G3_MF(2, fa0, fb0); G3_RD_A(fa1, sb, 1, 2); G3_SB();
    G3_MF(3, fa0, fb0); G3_RD_A(fa1, sb, 1, 3); if (d3) issue_one(s + 3, 4); G3_SB();
    G3_MF(4, fa0, fb0); G3_RD_B(fb1, sb, 1, 0); G3_SB();
    G3_MF(5, fa0, fb0); G3_RD_B(fb1, sb, 1, 1); G3_SB();
    G3_MF(6, fa0, fb0); G3_RD_B(fb1, sb, 1, 2); G3_SB();
    G3_MF(7, fa0, fb0); G3_RD_B(fb1, sb, 1, 3); if (d3) issue_one(s + 3, 5); G3_SB();
    G3_MF(8, fa0, fb0); G3_MF(9, fa0, fb0); G3_MF(10, fa0, fb0); G3_MF(11, fa0, fb0); if (d3) issue_one(s + 3, 6); G3_SB();
    G3_MF(12, fa0, fb0); G3_MF(13, fa0, fb0); G3_MF(14, fa0, fb0); G3_MF(15, fa0, fb0); if (d3) issue_one(s + 3, 7); G3_SB();
    LGKM_WAIT0();
    G3_JOIN(fb1);
    // own pieces of stage s+1 complete: stages s+2 and s+3 may stay in flight
    g3_vm_wait((s + 2 < S ? 8 : 0) + (d3 ? 8 : 0));
    raw_barrier();   // stage s+1 landed for everyone; everyone has finished reading stage s
    // ---- second k-step: multiply (s, 1); fetch (s+1, 0); first half of the DMAs of stage s+4 (slot of stage s)
    G3_MF(0, fa1, fb1); if (nx) G3_RD_A(fa0, sn, 0, 0); G3_SB();
    G3_MF(1, fa1, fb1); if (nx) G3_RD_A(fa0, sn, 0, 1); G3_SB();
    G3_MF(2, fa1, fb1); if (nx) G3_RD_A(fa0, sn, 0, 2); G3_SB();
    G3_MF(3, fa1, fb1); if (nx) G3_RD_A(fa0, sn, 0, 3); if (d4) issue_one(s + 4, 0); G3_SB();
    G3_MF(4, fa1, fb1); if (nx) G3_RD_B(fb0, sn, 0, 0); G3_SB();
    G3_MF(5, fa1, fb1); if (nx) G3_RD_B(fb0, sn, 0, 1); G3_SB();
    G3_MF(6, fa1, fb1); if (nx) G3_RD_B(fb0, sn, 0, 2); G3_SB();
    G3_MF(7, fa1, fb1); if (nx) G3_RD_B(fb0, sn, 0, 3); if (d4) issue_one(s + 4, 1); G3_SB();
    G3_MF(8, fa1, fb1); G3_MF(9, fa1, fb1); G3_MF(10, fa1, fb1); G3_MF(11, fa1, fb1); if (d4) issue_one(s + 4, 2); G3_SB();
    G3_MF(12, fa1, fb1); G3_MF(13, fa1, fb1); G3_MF(14, fa1, fb1); G3_MF(15, fa1, fb1); if (d4) issue_one(s + 4, 3); G3_SB();
    LGKM_WAIT0();
    if (nx) G3_JOIN(fb0);
  }

  // ---------------------------------------------------------------- epilogue
  raw_barrier();   // every fragment read is done: the ring becomes the per-wave transpose scratch
  if ((m0 + wm * 128 < M) && (n0 + wn * 128 < N)) {
    float* scratch = (float*)(smem + w * (EpiScratch<2>::FLOATS * 4));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      wave_store_tiles<bf16_t, 2, true>(&acc[mi][0], scratch, p.C, p.ldc, m0 + wm * 128 + mi * 32, n0 + wn * 128, M, N, p.alpha,
                                        p.beta, p.bias, lane, SOW_GEMM_NT(M));
      wave_store_tiles<bf16_t, 2, true>(&acc[mi][2], scratch, p.C, p.ldc, m0 + wm * 128 + mi * 32, n0 + wn * 128 + 64, M, N,
                                        p.alpha, p.beta, p.bias, lane, SOW_GEMM_NT(M));
    }
  }
}

int launch_gemm3(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                 const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                 float alpha, float beta, hipStream_t stream) {
  Gemm3Params p;
  p.A = (const bf16_t*)A, p.B = (const bf16_t*)B, p.A2 = (const bf16_t*)A2, p.B2 = (const bf16_t*)B2;
  p.C = (bf16_t*)C, p.bias = (const bf16_t*)bias;
  p.M = M, p.lda = lda, p.ldb = ldb, p.lda2 = lda2, p.ldb2 = ldb2, p.ldc = ldc;
  p.N = N, p.K = K, p.k2 = k2 < 64 ? k2 : 64;
  p.alpha = alpha, p.beta = beta;
  const int64_t tiles = (int64_t)ceil_div(M, G3_BM) * ceil_div(N, G3_BN);
  if (tiles <= 0) return SOW_OK;
  if (tiles > 0x7fffffff) return SOW_ERR_SHAPE;
  if (nt) {
    SOW_SET_MAX_LDS_ONCE(G3_LDS, gemm3_kernel<true>);
    hipLaunchKernelGGL(gemm3_kernel<true>, dim3((unsigned)tiles), dim3(G3_THREADS), G3_LDS, stream, p);
  } else {
    SOW_SET_MAX_LDS_ONCE(G3_LDS, gemm3_kernel<false>);
    hipLaunchKernelGGL(gemm3_kernel<false>, dim3((unsigned)tiles), dim3(G3_THREADS), G3_LDS, stream, p);
  }
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
