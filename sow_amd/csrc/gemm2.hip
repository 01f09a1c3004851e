// bf16 streaming GEMM with an optional K-extension (gfx950):
//     C[M,N] = alpha * (A[M,K] . op(B) + A2[M,64] . op(B2)) + beta * C + bias[N]
//
// This is the dense-accumulator form of the SoW layer: after the first accumulate() the
// reference computes  y = x @ W_acc + scale * (x @ A) @ B  (tn_gradient/layer/sow.py:109-121) as two
// products and an add; here the low-rank term rides along as 64 more K columns of ONE product,
//     y  = [x , h ] . [W_acc ; B]          h  = scale * x  . A    (h_save, written by the chain kernel)
//     dX = [dY, dh] . [W_acc^T ; A^T]      dh = scale * dY . B^T
// so y / dX are written once, in fp32-accumulated form, instead of being read-modify-written by a
// second kernel.  op(B): NT = B given as [N, K] (k-contiguous, the backward case, W_acc is [d_in, d_out]);
// NN = B given as [K, N] (the forward case).
//
// Shape of the problem: M = tokens (32768), N, K in {512, 1376}: arithmetic intensity ~ the bf16 ridge,
// i.e. the kernel is HBM-bound unless it keeps enough bytes in flight.  Design:
//   * 256 x 256 output tile per workgroup, 8 waves (2 x 4), 128 x 64 per wave = 4 x 2 MFMA 32x32x16
//     tiles (128 accumulator registers);
//   * K advances in 32-wide stages; one stage = A piece [256][32] + B piece (16 KiB each); a ring of
//     4 stage slots (128 KiB LDS), 3 stages (96 KiB) in flight by LDS-DMA, one raw s_barrier per
//     stage, counted s_waitcnt vmcnt -- never 0 inside the loop;
//   * k-contiguous pieces are [rows][32] images with 64-byte rows read by ds_read_b128, 16-byte chunk
//     c of a row at physical chunk c ^ ((row >> 2) & 3); the k-major B piece is a [32][256] image read
//     by ds_read_b64_tr_b16, chunk c of a row at c ^ ((row & 3) << 2).  Both conflict-free; the XOR is
//     applied to the per-lane SOURCE address (DMA writes LDS lane-linearly);
//   * out-of-range rows / K tails / rows >= k2 of the extension read a zero page instead (exact zeros);
//   * XCD-aware tile order: the column tiles of one row panel are neighbours on one XCD (shared L2).
#include "kernels.hpp"
#include "epilogue.hpp"
#include "lds_dma.hpp"
#include <cstdlib>

namespace sow {

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 32;
constexpr int G2_THREADS = 512;
constexpr int G2_NSLOT = 4;
constexpr int G2_PIECE = 256 * G2_BK * 2;       // 16 KiB: one operand's piece of a stage
constexpr int G2_STAGE = 2 * G2_PIECE;          // 32 KiB
constexpr int G2_LDS = G2_NSLOT * G2_STAGE;     // 128 KiB
constexpr int G2_DPW = 4;                       // DMA instructions per wave per stage (2 A + 2 B)


struct Gemm2Params {
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

template <bool NT> __global__ __launch_bounds__(G2_THREADS, 1) void gemm2_kernel(const Gemm2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 2, wn = w & 3, li = lane & 31, lh = lane >> 5;
  const int tiles_n = (p.N + G2_BN - 1) / G2_BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(lid / tiles_n) * G2_BM;
  const int n0 = (lid % tiles_n) * G2_BN;
  const int K = p.K, N = p.N;
  const int64_t M = p.M;
  const int s_main = (K + G2_BK - 1) / G2_BK;
  const int S = s_main + (p.A2 ? 2 : 0);
  const char* zp = zero_page_for(lane);

  // ---------------------------------------------------------------- DMA sources (per lane)
  // k-contiguous pieces ([256 rows][32 k], 64-byte rows): instruction i = 2w + ii covers rows 16i .. 16i+15
  const int crow = 16 * (2 * w) + (lane >> 2);          // row of instruction ii = 0 (ii = 1: +16)
  const int cpc = lane & 3;
  // k-major B piece ([32 k][256 n], 512-byte rows): instruction i covers k rows 2i, 2i+1
  const int krow = 2 * (2 * w) + (lane >> 5);           // k row of instruction ii = 0 (ii = 1: +2)
  const int kpc = lane & 31;
  // DMA instruction q of stage s: q = 0, 1 -> this wave's two A pieces; q = 2, 3 -> its two B pieces
  auto issue_one = [&](int s, int q) {
    char* slot = smem + (s % G2_NSLOT) * G2_STAGE;
    const bool ext = s >= s_main;
    const int k0 = ext ? (s - s_main) * G2_BK : s * G2_BK;
    const int ii = q & 1;
    if (q < 2) {
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
        dma16(src, slot + G2_PIECE + (2 * w + ii) * 1024);
      } else {
        const int krows = ext ? p.k2 : K;
        const int row = krow + 2 * ii;
        const int lc = kpc ^ ((row & 3) << 2);
        const int gk = k0 + row, gn = n0 + 8 * lc;
        const void* src = (gk < krows && gn < N) ? (const void*)(Bp + (int64_t)gk * ldb + gn) : (const void*)zp;
        dma16(src, slot + G2_PIECE + (2 * w + ii) * 1024);
      }
    }
  };
  auto issue = [&](int s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_one(s, q);
  };

  // ---------------------------------------------------------------- fragment addresses (per lane)
  const uint32_t base = lds_addr(smem);
  const int fsw = (li >> 2) & 3;                                        // b128 row swizzle (64-byte rows)
  const uint32_t a_off = (uint32_t)((wm * 128 + li) * 64);              // + mi * 2048
  uint32_t b_off[2];
  if constexpr (NT) {
    b_off[0] = (uint32_t)(G2_PIECE + (wn * 64 + li) * 64);              // + ni * 2048
    b_off[1] = b_off[0] + 2048;
  } else {
    const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3;
    const int r1 = 8 * (g >> 1) + q;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = wn * 64 + ni * 32 + 16 * (g & 1) + 4 * pp;
      b_off[ni] = (uint32_t)(G2_PIECE + r1 * 512 + (((col >> 3) ^ ((r1 & 3) << 2)) * 16) + (col & 7) * 2);
    }
  }
  const uint32_t ch0 = (uint32_t)(((0 + lh) ^ fsw) * 16), ch1 = (uint32_t)(((2 + lh) ^ fsw) * 16);

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // waves whose whole 128 x 64 sub-tile is outside the matrix only help with the DMA and the barriers
  const bool live = (m0 + wm * 128 < M) && (n0 + wn * 64 < N);

  const int pre = S < (G2_NSLOT - 1) ? S : (G2_NSLOT - 1);
  for (int s = 0; s < pre; ++s) issue(s);

  // One barrier per stage.  (A software-pipelined variant -- barrier between the two k-steps, fragment
  // reads one k-step ahead, DMA issues interleaved with the MFMAs -- measured 3-8 % SLOWER: the two
  // waves of a SIMD already cover each other's read latency.)
#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    const int newer = (S - 1 - s) < (G2_NSLOT - 2) ? (S - 1 - s) : (G2_NSLOT - 2);
    wait_groups<G2_DPW>(newer);   // this wave's pieces of stage s have landed
    raw_barrier();                // ... everyone's have, and everyone is done with stage s-1
    if (s + G2_NSLOT - 1 < S) issue(s + G2_NSLOT - 1);   // into the slot of stage s-1
    if (live) {
      const uint32_t sb = base + (uint32_t)((s % G2_NSLOT) * G2_STAGE);
      u32x4 af[2][4], bf[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint32_t aa = sb + a_off + (ks ? ch1 : ch0);
        DS_READ_B128(af[ks][0], aa, 0);
        DS_READ_B128(af[ks][1], aa, 2048);
        DS_READ_B128(af[ks][2], aa, 4096);
        DS_READ_B128(af[ks][3], aa, 6144);
        if constexpr (NT) {
          const uint32_t bb = sb + (ks ? ch1 : ch0);
          DS_READ_B128(bf[ks][0], bb + b_off[0], 0);
          DS_READ_B128(bf[ks][1], bb + b_off[1], 0);
        }
      }
      if constexpr (!NT) {
        u32x2 bl[2][2], bh[2][2];
        DS_READ_TR(bl[0][0], sb + b_off[0], 0);
        DS_READ_TR(bh[0][0], sb + b_off[0], 2048);
        DS_READ_TR(bl[0][1], sb + b_off[1], 0);
        DS_READ_TR(bh[0][1], sb + b_off[1], 2048);
        DS_READ_TR(bl[1][0], sb + b_off[0], 8192);
        DS_READ_TR(bh[1][0], sb + b_off[0], 8192 + 2048);
        DS_READ_TR(bl[1][1], sb + b_off[1], 8192);
        DS_READ_TR(bh[1][1], sb + b_off[1], 8192 + 2048);
        LGKM_WAIT0();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bf[ks][ni] = join2(bl[ks][ni], bh[ks][ni]);
      } else {
        LGKM_WAIT0();
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = mfma32(as_bf16x8(af[ks][mi]), as_bf16x8(bf[ks][ni]), acc[mi][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---------------------------------------------------------------- epilogue
  raw_barrier();   // every fragment read is done: the ring becomes the per-wave transpose scratch
  if (live) {
    float* scratch = (float*)(smem + w * (EpiScratch<2>::FLOATS * 4));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      wave_store_tiles<bf16_t, 2, true>(acc[mi], scratch, p.C, p.ldc, m0 + wm * 128 + mi * 32, n0 + wn * 64, M, N,
                                        p.alpha, p.beta, p.bias, lane, SOW_GEMM_NT(M));
  }
}

// out[rows, 64] = [in[rows, r] | 0]: the k-contiguous extension operand of the backward product (A is
// [d_in, r] with 2r-byte rows, which no 16-byte DMA piece can address row by row)
__global__ __launch_bounds__(256) void pad64_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows,
                                                    int r) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // one 8-column group per thread
  const int row = idx >> 3, c0 = (idx & 7) * 8;
  if (row >= rows) return;
  u32x4 v;
  bf16_t* e = (bf16_t*)&v;
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = (c0 + j < r) ? in[(int64_t)row * r + c0 + j] : (bf16_t)0.f;
  *(u32x4*)(out + (int64_t)row * 64 + c0) = v;
}

int launch_pad64(const void* in, void* out, int rows, int r, hipStream_t stream) {
  if (rows <= 0) return SOW_OK;
  hipLaunchKernelGGL(pad64_kernel, dim3((rows * 8 + 255) / 256), dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out,
                     rows, r);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

// ------------------------------------------------------------------------------------------------
static bool g2_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool gemm2_supported(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                     const void* B2, int64_t ldb2, const void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                     int dtype) {
  if (dtype != SOW_BF16 || !A || !B || !C) return false;
  // one 256x256 tile per CU: below ~160 tiles the 128x128 tiles of gemm3s.hip fill the chip better (launch_gemm2
  // picks the kernel); below ~96 of those, or with a short K, the generic kernel is as good
  if (N < 64 || K < 32) return false;
  if ((int64_t)ceil_div(M, G2_BM) * ceil_div(N, G2_BN) < 160 &&
      ((int64_t)ceil_div(M, 128) * ceil_div(N, 128) < 96 || K < 512 || sw_on(SW_NO_GEMM3S)))
    return false;
  if (sw_on(SW_FORCE_GEMM_V1)) return false;      // A/B switch, as SOW_AMD_FORCE_CHAIN_V1
  if (K % 8 || N % 8 || lda % 8 || ldb % 8 || ldc % 8) return false;
  if (!g2_al16(A) || !g2_al16(B) || !g2_al16(C) || (bias && !g2_al16(bias))) return false;
  if (A2) {
    if (!B2 || lda2 % 8 || ldb2 % 8 || !g2_al16(A2) || !g2_al16(B2)) return false;
  }
  (void)nt;
  return true;
}

int launch_gemm2(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                 const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                 float alpha, float beta, hipStream_t stream, void* ws, size_t ws_bytes) {
  Gemm2Params p;
  p.A = (const bf16_t*)A, p.B = (const bf16_t*)B, p.A2 = (const bf16_t*)A2, p.B2 = (const bf16_t*)B2;
  p.C = (bf16_t*)C, p.bias = (const bf16_t*)bias;
  p.M = M, p.lda = lda, p.ldb = ldb, p.lda2 = lda2, p.ldb2 = ldb2, p.ldc = ldc;
  p.N = N, p.K = K, p.k2 = k2 < 64 ? k2 : 64;
  p.alpha = alpha, p.beta = beta;
  const int64_t tiles = (int64_t)ceil_div(M, G2_BM) * ceil_div(N, G2_BN);
  if (tiles <= 0) return SOW_OK;
  if (tiles > 0x7fffffff) return SOW_ERR_SHAPE;
  // gemm4 (anti-phase wave groups, 64-wide K-tiles): 15-30 % faster than the kernels below wherever its 256 x 256 tiles
  // fill at least half of the chip (profiles/r03_gemm4_vs_round2_vs_hipblaslt.txt: 4096^3 1.38 vs 0.97-1.03 PF, 32768 x 512
  // -> 1376 58 vs 72 us); with fewer tiles (M = 1024 x N = 4096: 64) the 128 x 128 tiles of gemm3s fill more CUs.
  // GEMM4 = 1 forces it on every supported shape, 0 forbids it.
  // With scratch from the caller, short products (<= 128 tiles, long K) run gemm4 split over K: M = 1024 x N = 4096 as 64 tiles
  // x 4 splits instead of gemm3s's 256 small tiles.
  const int g4 = sw(SW_GEMM4);
  const int64_t work = tiles * gemm4_splits(M, N, K, A2 != nullptr, ws, ws_bytes);
  if ((g4 > 0 || (g4 < 0 && work >= 120 && K >= 64)) &&
      gemm4_supported(A, lda, B, ldb, nt, A2, lda2, B2, ldb2, C, ldc, bias, M, N, K, SOW_BF16))
    return launch_gemm4(A, lda, B, ldb, nt, A2, lda2, B2, ldb2, k2, C, ldc, bias, M, N, K, alpha, beta, stream, ws, ws_bytes);
  const int gs = sw(SW_GEMM3S);
  if (gs >= 0 ? gs != 0 : ((int64_t)ceil_div(M, G2_BM) * ceil_div(N, G2_BN) < 160))
    return launch_gemm3s(A, lda, B, ldb, nt, A2, lda2, B2, ldb2, k2, C, ldc, bias, M, N, K, alpha, beta, stream);
  // long K: the one-wave-per-SIMD kernel (gemm3.hip) is 7-19 % faster from K ~ 2048 on (4096^3: 1.05 vs 0.90-0.96 PF);
  // at the llama_60m widths (K <= 1376) the two are level or this one is ahead.  SOW_AMD_GEMM3=1 / =0 force either.
  const int g3 = sw(SW_GEMM3);
  if (g3 >= 0 ? g3 != 0 : (K >= 2048))
    return launch_gemm3(A, lda, B, ldb, nt, A2, lda2, B2, ldb2, k2, C, ldc, bias, M, N, K, alpha, beta, stream);
  if (nt) {
    SOW_SET_MAX_LDS_ONCE(G2_LDS, gemm2_kernel<true>);
    hipLaunchKernelGGL(gemm2_kernel<true>, dim3((unsigned)tiles), dim3(G2_THREADS), G2_LDS, stream, p);
  } else {
    SOW_SET_MAX_LDS_ONCE(G2_LDS, gemm2_kernel<false>);
    hipLaunchKernelGGL(gemm2_kernel<false>, dim3((unsigned)tiles), dim3(G2_THREADS), G2_LDS, stream, p);
  }
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
