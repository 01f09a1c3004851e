// bf16 streaming GEMM for short M (few row tiles): 128 x 128 workgroup tiles, one wave per SIMD.  Same contract as
// gemm2.hip / gemm3.hip:
//     C[M,N] = alpha * (A[M,K] . op(B) + A2[M,64] . op(B2)) + beta * C + bias[N]
//
// The finetune shapes of BASELINE config 5 (T = 1024 tokens, 4096 -> 4096 / 11008 -> 4096) give 16 - 64 tiles of
// 256 x 256: a quarter of the chip.  With 128 x 128 tiles they are 128 - 256 workgroups.  Structure = gemm3.hip with
// half the tile edge: 4 waves (2 x 2), 64 x 64 per wave (4 accumulator tiles), a stage = 64 k as two 32-wide slices
// [A0 | B0 | A1 | B1] (the 64-byte-row LDS images of gemm2.hip with 128 rows, 32 KiB a stage, 4 slots), 8 DMA
// instructions per wave per stage.  Four k-steps per stage; the fragments of k-step j+2 are fetched while k-step j
// multiplies (four fragment sets, 64 VGPRs -- the accumulators are only 64), so every read has two k-steps = 8 MFMAs
// to land; two DMAs ride in every k-step.  One raw s_barrier per stage, after k-step 1: by then every read of the
// stage has completed (its slot can be refilled) and the next stage is about to be read.
// The instruction mix is worse than the 256-tile kernels' (1.5 memory instructions per MFMA against 0.75 / 1.1), so
// this kernel is used only where those leave most of the chip idle.
#include "kernels.hpp"
#include "epilogue.hpp"
#include "lds_dma.hpp"
#include <cstdlib>

namespace sow {

constexpr int GS_BM = 128, GS_BN = 128, GS_BK = 64;
constexpr int GS_THREADS = 256;
constexpr int GS_NSLOT = 4;
constexpr int GS_PIECE = 128 * 32 * 2;          // 8 KiB: one operand's 32-wide slice
constexpr int GS_SLICE = 2 * GS_PIECE;          // 16 KiB: [A | B] of one slice
constexpr int GS_STAGE = 2 * GS_SLICE;          // 32 KiB
constexpr int GS_LDS = GS_NSLOT * GS_STAGE;     // 128 KiB

struct Gemm3sParams {
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

__device__ __forceinline__ void gs_vm_wait(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
  }
  __builtin_amdgcn_sched_barrier(0);
}
#define GS_SB() __builtin_amdgcn_sched_barrier(0)

template <bool NT> __global__ __launch_bounds__(GS_THREADS, 1) void gemm3s_kernel(const Gemm3sParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = (p.N + GS_BN - 1) / GS_BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(lid / tiles_n) * GS_BM;
  const int n0 = (lid % tiles_n) * GS_BN;
  const int K = p.K, N = p.N;
  const int64_t M = p.M;
  const int s_main = (K + GS_BK - 1) / GS_BK;
  const int S = s_main + (p.A2 ? 1 : 0);     // the 64-wide extension is one stage
  const char* zp = zero_page_for(lane);

  // ---------------------------------------------------------------- DMA: q = 4*slice + 2*operand + ii, instruction 2w + ii
  const int crow = 16 * (2 * w) + (lane >> 2);   // k-contiguous pieces [128 rows][32 k]: instruction i covers rows 16i ..+15
  const int cpc = lane & 3;
  const int krow = 4 * (2 * w) + (lane >> 4);    // k-major B piece [32 k][128 n], 256-byte rows: instruction i covers k rows 4i ..+3
  const int kpc = lane & 15;
  auto issue_slow = [&](int s, int q) {
    char* slot = smem + (s % GS_NSLOT) * GS_STAGE + (q >> 2) * GS_SLICE;
    const bool ext = s >= s_main;
    const int k0 = (ext ? 0 : s * GS_BK) + (q >> 2) * 32;
    const int ii = q & 1;
    if (!(q & 2)) {
      const bf16_t* Ap = ext ? p.A2 : p.A;
      const int64_t lda = ext ? p.lda2 : p.lda;
      const int klim = ext ? 64 : K;
      const int row = crow + 16 * ii;
      const int lc = cpc ^ ((row >> 2) & 3);
      const int64_t gr = m0 + row;
      const void* src = (gr < M && k0 + 8 * lc < klim) ? (const void*)(Ap + gr * lda + k0 + 8 * lc) : (const void*)zp;
      dma16(src, slot + (2 * w + ii) * 1024);
    } else {
      const bf16_t* Bp = ext ? p.B2 : p.B;
      const int64_t ldb = ext ? p.ldb2 : p.ldb;
      if constexpr (NT) {
        const int klim = ext ? 64 : K;
        const int row = crow + 16 * ii;
        const int lc = cpc ^ ((row >> 2) & 3);
        const int gn = n0 + row;
        const void* src = (gn < N && k0 + 8 * lc < klim) ? (const void*)(Bp + (int64_t)gn * ldb + k0 + 8 * lc) : (const void*)zp;
        dma16(src, slot + GS_PIECE + (2 * w + ii) * 1024);
      } else {
        const int krows = ext ? p.k2 : K;
        const int row = krow + 4 * ii;
        const int lc = kpc ^ ((row & 3) << 2);
        const int gk = k0 + row, gn = n0 + 8 * lc;
        const void* src = (gk < krows && gn < N) ? (const void*)(Bp + (int64_t)gk * ldb + gn) : (const void*)zp;
        dma16(src, slot + GS_PIECE + (2 * w + ii) * 1024);
      }
    }
  };
  // fast path (main stages before the last one): running per-lane pointers, one 64-bit add per DMA
  const char* ptr[8];
  int stride[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int sl = q >> 2, ii = q & 1;
    if (!(q & 2)) {
      const int row = crow + 16 * ii;
      const int lc = cpc ^ ((row >> 2) & 3);
      const int64_t gr = m0 + row;
      const bool v = gr < M;
      ptr[q] = v ? (const char*)(p.A + gr * p.lda + sl * 32 + 8 * lc) : zp;
      stride[q] = v ? 2 * GS_BK : 0;
    } else if constexpr (NT) {
      const int row = crow + 16 * ii;
      const int lc = cpc ^ ((row >> 2) & 3);
      const int gn = n0 + row;
      const bool v = gn < N;
      ptr[q] = v ? (const char*)(p.B + (int64_t)gn * p.ldb + sl * 32 + 8 * lc) : zp;
      stride[q] = v ? 2 * GS_BK : 0;
    } else {
      const int row = krow + 4 * ii;
      const int lc = kpc ^ ((row & 3) << 2);
      const int gn = n0 + 8 * lc;
      const bool v = gn < N;
      ptr[q] = v ? (const char*)(p.B + (int64_t)(sl * 32 + row) * p.ldb + gn) : zp;
      stride[q] = v ? (int)(2 * GS_BK * p.ldb) : 0;
    }
  }
  auto issue_one = [&](int s, int q) {
    if (s >= s_main - 1) {
      issue_slow(s, q);
      return;
    }
    char* slot = smem + (s % GS_NSLOT) * GS_STAGE + (q >> 2) * GS_SLICE + ((q & 2) ? GS_PIECE : 0);
    dma16((const void*)ptr[q], slot + (2 * w + (q & 1)) * 1024);
    ptr[q] += stride[q];
  };

  // ---------------------------------------------------------------- fragment addresses (per lane)
  const uint32_t base = lds_addr(smem);
  const int fsw = (li >> 2) & 3;
  const uint32_t a_off = (uint32_t)((wm * 64 + li) * 64);              // + mi * 2048
  uint32_t b_off[2];
  if constexpr (NT) {
    b_off[0] = (uint32_t)(GS_PIECE + (wn * 64 + li) * 64);
    b_off[1] = b_off[0] + 2048;
  } else {
    const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3;
    const int r1 = 8 * (g >> 1) + q;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = wn * 64 + ni * 32 + 16 * (g & 1) + 4 * pp;
      b_off[ni] = (uint32_t)(GS_PIECE + r1 * 256 + (((col >> 3) ^ ((r1 & 3) << 2)) * 16) + (col & 7) * 2);
    }
  }
  const uint32_t ch0 = (uint32_t)(((0 + lh) ^ fsw) * 16), ch1 = (uint32_t)(((2 + lh) ^ fsw) * 16);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // fragment set j = k-step j of a stage (slice j >> 1, half j & 1)
  u32x4 fa[4][2], fb[4][2];
  u32x2 tl[4][2], th[4][2];   // NN: halves of the transposed B reads, joined before use
  // reads of k-step J of the stage at LDS address SB into set J (4 instructions NT, 6 NN)
#define GS_READ(SB, J)                                                                   \
  do {                                                                                   \
    const uint32_t sl_ = (SB) + ((J) >> 1) * GS_SLICE;                                   \
    const uint32_t ch_ = ((J)&1) ? ch1 : ch0;                                            \
    DS_READ_B128(fa[J][0], sl_ + a_off + ch_, 0);                                        \
    DS_READ_B128(fa[J][1], sl_ + a_off + ch_, 2048);                                     \
    if constexpr (NT) {                                                                  \
      DS_READ_B128(fb[J][0], sl_ + b_off[0] + ch_, 0);                                   \
      DS_READ_B128(fb[J][1], sl_ + b_off[1] + ch_, 0);                                   \
    } else {                                                                             \
      if ((J)&1) {                                                                       \
        DS_READ_TR(tl[J][0], sl_ + b_off[0], 4096);                                      \
        DS_READ_TR(th[J][0], sl_ + b_off[0], 4096 + 1024);                               \
        DS_READ_TR(tl[J][1], sl_ + b_off[1], 4096);                                      \
        DS_READ_TR(th[J][1], sl_ + b_off[1], 4096 + 1024);                               \
      } else {                                                                           \
        DS_READ_TR(tl[J][0], sl_ + b_off[0], 0);                                         \
        DS_READ_TR(th[J][0], sl_ + b_off[0], 1024);                                      \
        DS_READ_TR(tl[J][1], sl_ + b_off[1], 0);                                         \
        DS_READ_TR(th[J][1], sl_ + b_off[1], 1024);                                      \
      }                                                                                  \
    }                                                                                    \
  } while (0)
  // wait until the reads of set J have landed: the reads issued after them (one more set) may stay outstanding
#define GS_WAIT_SET(J, MORE)                                                             \
  do {                                                                                   \
    if (MORE) {                                                                          \
      if constexpr (NT) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");               \
      else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");                            \
    } else {                                                                             \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
    }                                                                                    \
    GS_SB();                                                                             \
    if constexpr (!NT) {                                                                 \
      fb[J][0] = join2(tl[J][0], th[J][0]);                                              \
      fb[J][1] = join2(tl[J][1], th[J][1]);                                              \
    }                                                                                    \
  } while (0)
#define GS_MF(J, mi, ni) acc[mi][ni] = mfma32(as_bf16x8(fa[J][mi]), as_bf16x8(fb[J][ni]), acc[mi][ni])

  // ---------------------------------------------------------------- prologue: stages 0..2 and half of stage 3
  for (int s = 0; s < 3 && s < S; ++s)
    for (int q = 0; q < 8; ++q) issue_one(s, q);
  if (3 < S)
    for (int q = 0; q < 4; ++q) issue_one(3, q);
  gs_vm_wait((1 < S ? 8 : 0) + (2 < S ? 8 : 0) + (3 < S ? 4 : 0));
  raw_barrier();   // stage 0 landed for everyone
  GS_READ(base, 0);
  GS_READ(base, 1);

#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    const uint32_t sb = base + (uint32_t)((s % GS_NSLOT) * GS_STAGE);
    const uint32_t sn = base + (uint32_t)(((s + 1) % GS_NSLOT) * GS_STAGE);
    const bool d3 = s + 3 < S, d4 = s + 4 < S, nx = s + 1 < S;
    // k-step 0: sets 0 (landing) and 1 are in flight; fetch set 2; DMAs 4, 5 of stage s+3 (slot of stage s-1)
    GS_WAIT_SET(0, true);
    GS_MF(0, 0, 0); GS_MF(0, 0, 1); GS_SB();
    GS_READ(sb, 2); GS_SB();
    GS_MF(0, 1, 0); if (d3) issue_one(s + 3, 4); GS_SB();
    GS_MF(0, 1, 1); if (d3) issue_one(s + 3, 5); GS_SB();
    // k-step 1: fetch set 3; DMAs 6, 7
    GS_WAIT_SET(1, true);
    GS_MF(1, 0, 0); GS_MF(1, 0, 1); GS_SB();
    GS_READ(sb, 3); GS_SB();
    GS_MF(1, 1, 0); if (d3) issue_one(s + 3, 6); GS_SB();
    GS_MF(1, 1, 1); if (d3) issue_one(s + 3, 7); GS_SB();
    // every read of stage s has been issued; wait for them (sets 2, 3) and for the own pieces of stage s+1
    GS_WAIT_SET(2, false);
    if constexpr (!NT) {
      fb[3][0] = join2(tl[3][0], th[3][0]);
      fb[3][1] = join2(tl[3][1], th[3][1]);
    }
    gs_vm_wait((s + 2 < S ? 8 : 0) + (d3 ? 8 : 0));
    raw_barrier();   // stage s+1 landed for everyone; everyone has finished reading stage s
    // k-step 2: fetch set 0 of stage s+1; DMAs 0, 1 of stage s+4 (slot of stage s)
    GS_MF(2, 0, 0); GS_MF(2, 0, 1); GS_SB();
    if (nx) GS_READ(sn, 0);
    GS_SB();
    GS_MF(2, 1, 0); if (d4) issue_one(s + 4, 0); GS_SB();
    GS_MF(2, 1, 1); if (d4) issue_one(s + 4, 1); GS_SB();
    // k-step 3: fetch set 1 of stage s+1; DMAs 2, 3
    GS_MF(3, 0, 0); GS_MF(3, 0, 1); GS_SB();
    if (nx) GS_READ(sn, 1);
    GS_SB();
    GS_MF(3, 1, 0); if (d4) issue_one(s + 4, 2); GS_SB();
    GS_MF(3, 1, 1); if (d4) issue_one(s + 4, 3); GS_SB();
  }

  // ---------------------------------------------------------------- epilogue
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();   // every fragment read is done: the ring becomes the per-wave transpose scratch
  if ((m0 + wm * 64 < M) && (n0 + wn * 64 < N)) {
    float* scratch = (float*)(smem + w * (EpiScratch<2>::FLOATS * 4));
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      wave_store_tiles<bf16_t, 2, true>(&acc[mi][0], scratch, p.C, p.ldc, m0 + wm * 64 + mi * 32, n0 + wn * 64, M, N, p.alpha,
                                        p.beta, p.bias, lane, SOW_GEMM_NT(M));
  }
}

int launch_gemm3s(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                  const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                  float alpha, float beta, hipStream_t stream) {
  Gemm3sParams p;
  p.A = (const bf16_t*)A, p.B = (const bf16_t*)B, p.A2 = (const bf16_t*)A2, p.B2 = (const bf16_t*)B2;
  p.C = (bf16_t*)C, p.bias = (const bf16_t*)bias;
  p.M = M, p.lda = lda, p.ldb = ldb, p.lda2 = lda2, p.ldb2 = ldb2, p.ldc = ldc;
  p.N = N, p.K = K, p.k2 = k2 < 64 ? k2 : 64;
  p.alpha = alpha, p.beta = beta;
  const int64_t tiles = (int64_t)ceil_div(M, GS_BM) * ceil_div(N, GS_BN);
  if (tiles <= 0) return SOW_OK;
  if (tiles > 0x7fffffff) return SOW_ERR_SHAPE;
  if (nt) {
    SOW_SET_MAX_LDS_ONCE(GS_LDS, gemm3s_kernel<true>);
    hipLaunchKernelGGL(gemm3s_kernel<true>, dim3((unsigned)tiles), dim3(GS_THREADS), GS_LDS, stream, p);
  } else {
    SOW_SET_MAX_LDS_ONCE(GS_LDS, gemm3s_kernel<false>);
    hipLaunchKernelGGL(gemm3s_kernel<false>, dim3((unsigned)tiles), dim3(GS_THREADS), GS_LDS, stream, p);
  }
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
