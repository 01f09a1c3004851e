// bf16 GEMM with an optional K-extension, two wave groups in anti-phase (gfx950):
//     C[M,N] = alpha * (A[M,K] . op(B) + A2[M,64] . op(B2)) + beta * C + bias[N]
// Same contract as gemm2.hip (the dense-accumulator form of the SoW layer, tn_gradient/layer/sow.py:109-121 as ONE
// fp32-accumulated product: y = [x, h] . [W_acc; B], dX = [dY, dh] . [W_acc^T; A^T]); a different main loop.
//
// Why another main loop.  gemm2 / gemm3 advance K in 32-wide stages with one barrier per stage and run at 36-50 % of the
// matrix pipe: every wave loads, waits and multiplies in the same rhythm, so the pipe idles while the fragments and the
// DMA issue go out.  Here (the structure of the CDNA guide's 256 x 256 "8-phase" template, rebuilt for this contract):
//   * 256 x 256 tile, K in 64-wide tiles, 8 waves = 2 (rows) x 4 (columns), 128 x 64 per wave, v_mfma_f32_16x16x32_bf16
//     (the chip holds a higher clock on it than on 32x32x16 at equal cycles per flop);
//   * the two wave ROWS run one barrier apart: between two barriers one group issues its 16 MFMAs of a 64 x 32 quadrant
//     while the other (its SIMD partners) reads the next quadrant's fragments and issues the DMA of one half-tile;
//   * a K-tile lives in LDS as four 16-KiB half-tiles cut along the quadrant boundaries -- A0 / A1 = the first / second 64
//     rows of BOTH wave rows, B0 / B1 = the first / second 32 columns of ALL FOUR wave columns -- so that a half-tile is
//     read in exactly one phase of its K-tile and its slot can be re-filled two phases later: quadrant order (0,0) (0,1)
//     (1,1) (1,0) reads A0 + B0, B1, A1, nothing; phase P issues the DMA of half-tile P + 6 (order A0 B0 B1 A1 per tile),
//     2 x 64 KiB ring = 8 slots, up to 5 half-tiles (80 KiB) in flight per CU;
//   * counted waits only: after its DMA issue every phase waits for vmcnt(8) = everything but the four youngest
//     half-tiles, which is exactly what phase P + 1 reads; that wait is followed by two barriers before the read (the
//     groups are one barrier apart);
//   * k-contiguous half-tiles are [128][64] images (128-byte rows), 16-byte chunk c of row r at c ^ ((r >> 1) & 7):
//     every ds_read_b128 lane group touches 16 distinct 16-byte slots; the k-major B of the forward product stays k-major
//     ([64 k][128 n], 256-byte rows, chunk c of k-row k at c ^ (((k & 3) | ((k >> 3) & 1) << 2) << 1)) and is read
//     with ds_read_b64_tr_b16.  The XOR sits on the per-lane SOURCE address (LDS-DMA writes lane-linearly);
//   * the products are computed TRANSPOSED (the W fragment is the MFMA's A operand): a lane then holds 4 consecutive
//     output columns of one token row, the epilogue parks 16 x 64 blocks in a wave-private LDS scratch with four
//     16-byte writes and stores whole 128-byte row segments;
//   * per-lane source pointers for the eight DMA instructions of a K-tile are carried in registers and advanced by one
//     64-bit add; rows / columns beyond the matrix are CLAMPED (they only feed outputs that are never stored); the K tail
//     and the extension tile take a checked path that reads the zero page where k is out of range;
//   * short M (config 5: T = 1024 -> 64 tiles for a 4096-wide output) runs split-K: S blocks per tile, each on a K range,
//     fp32 partial products through the caller's workspace, summed in split order by a second, chip-wide launch.
#include "kernels.hpp"
#include "lds_dma.hpp"
#include <type_traits>

namespace sow {

constexpr int G4_BM = 256, G4_BN = 256, G4_BK = 64;
constexpr int G4_THREADS = 512;
constexpr int G4_HALF = 128 * G4_BK * 2;   // 16 KiB
constexpr int G4_BUF = 4 * G4_HALF;        // 64 KiB: A0 | A1 | B0 | B1
constexpr int G4_LDS = 2 * G4_BUF;         // 128 KiB
constexpr int G4_OFF_A0 = 0, G4_OFF_A1 = G4_HALF, G4_OFF_B0 = 2 * G4_HALF, G4_OFF_B1 = 3 * G4_HALF;
constexpr int G4_SCR_LD = 68;              // floats per scratch row (64 + 4: 16-byte aligned rows, staggered banks)
constexpr int G4_SCR = 16 * G4_SCR_LD * 4; // bytes of one wave's epilogue scratch
constexpr int G4_HIMG = G4_LDS;            // gemm4h: the projected [256][64] tile, two half images of 16 KiB behind the ring
constexpr int G4_LDS_H = G4_LDS + 2 * G4_HALF;   // 160 KiB
constexpr int G4_PA_BUF = 2 * G4_HALF + 8192;    // gemm4h projection pass: A0 | A1 | F (8 KiB) per K-tile, three buffers in the ring

struct Gemm4Params {
  const bf16_t* A;
  const bf16_t* B;
  const bf16_t* A2;   // [M, 64] or nullptr
  const bf16_t* B2;   // NT: [N, 64]; NN: [k2, N]
  bf16_t* C;
  const bf16_t* bias;
  int64_t M, lda, ldb, lda2, ldb2, ldc;
  int N, K, k2;
  int k2e;            // NT: valid k columns of B2's rows (64: zero-padded rows; r: raw [N][r] rows, gemm4h)
  float alpha, beta;
  int nt_store;
  // HF = true (gemm4h): the rank-r projection H = hscale * A . op(F) is computed by the kernel itself and is the extension's
  // A operand (A2 is not read): F is [r, K] (NT, ld ldf) or a zero-padded [K, 64] (NN, ld ldf); Hout [M, 64] receives the
  // saved copy (column 63 <- 1.0 when r < 64) from the first column tile
  const bf16_t* F;
  int64_t ldf;
  bf16_t* Hout;
  int r;
  float hscale;
  // split-K (short M: fewer output tiles than CUs).  splits > 1: block b = split * tiles + tile runs K-tiles
  // [split * kt_per, ...) of its tile and leaves the fp32 sum of that range in partials[split][M][N]; a second launch
  // (gemm4_splitk_reduce_kernel, every CU) adds the splits in order and applies alpha / beta / bias.
  int splits, kt_per;
  float* partials;
};

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
}

// LDS reads with compile-time immediate offsets (inline asm: invisible to hipcc's vmcnt bookkeeping, see lds_dma.hpp)
template <int OFF> __device__ __forceinline__ void g4_rd128(u32x4& d, uint32_t a) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void g4_rdtr(u32x2& d, uint32_t a) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF) : "memory");
}
// x fragments of row half MH: 4 row tiles x 2 k-steps
template <int MH> __device__ __forceinline__ void g4_read_a(u32x4 (&af)[4][2], uint32_t a0, uint32_t a1) {
  g4_rd128<MH * G4_HALF + 0 * 2048>(af[0][0], a0);
  g4_rd128<MH * G4_HALF + 0 * 2048>(af[0][1], a1);
  g4_rd128<MH * G4_HALF + 1 * 2048>(af[1][0], a0);
  g4_rd128<MH * G4_HALF + 1 * 2048>(af[1][1], a1);
  g4_rd128<MH * G4_HALF + 2 * 2048>(af[2][0], a0);
  g4_rd128<MH * G4_HALF + 2 * 2048>(af[2][1], a1);
  g4_rd128<MH * G4_HALF + 3 * 2048>(af[3][0], a0);
  g4_rd128<MH * G4_HALF + 3 * 2048>(af[3][1], a1);
}
// W fragments of column half NH from a k-contiguous image: 2 column tiles x 2 k-steps
template <int NH> __device__ __forceinline__ void g4_read_b_nt(u32x4 (&bf)[2][2], uint32_t b0, uint32_t b1) {
  g4_rd128<NH * G4_HALF + 0 * 2048>(bf[0][0], b0);
  g4_rd128<NH * G4_HALF + 0 * 2048>(bf[0][1], b1);
  g4_rd128<NH * G4_HALF + 1 * 2048>(bf[1][0], b0);
  g4_rd128<NH * G4_HALF + 1 * 2048>(bf[1][1], b1);
}
// ... from a k-major image (transposed reads): [nt][ks] low / high k quads
template <int NH> __device__ __forceinline__ void g4_read_b_nn(u32x2 (&bl)[2][2], u32x2 (&bh)[2][2], uint32_t n0a, uint32_t n1a) {
  g4_rdtr<NH * G4_HALF + 0>(bl[0][0], n0a);
  g4_rdtr<NH * G4_HALF + 1024>(bh[0][0], n0a);
  g4_rdtr<NH * G4_HALF + 8192>(bl[0][1], n0a);
  g4_rdtr<NH * G4_HALF + 8192 + 1024>(bh[0][1], n0a);
  g4_rdtr<NH * G4_HALF + 0>(bl[1][0], n1a);
  g4_rdtr<NH * G4_HALF + 1024>(bh[1][0], n1a);
  g4_rdtr<NH * G4_HALF + 8192>(bl[1][1], n1a);
  g4_rdtr<NH * G4_HALF + 8192 + 1024>(bh[1][1], n1a);
}

// kinds of half-tile, in DMA issue order within a K-tile
enum : int { G4_A0 = 0, G4_B0 = 1, G4_B1 = 2, G4_A1 = 3 };
template <int KIND> __device__ __forceinline__ constexpr int g4_slot_off() {
  return KIND == G4_A0 ? G4_OFF_A0 : KIND == G4_A1 ? G4_OFF_A1 : KIND == G4_B0 ? G4_OFF_B0 : G4_OFF_B1;
}

// SK = the split-K form (its own instantiation: the plain kernel sits at 249-250 VGPRs and must not pay for the K-range
// bookkeeping or the partial-sum epilogue)
template <bool NT, bool HF, bool SK = false> __global__ __launch_bounds__(G4_THREADS, 2) void gemm4_kernel(const Gemm4Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int r16 = lane & 15, g = lane >> 4;
  const int tiles_n = (p.N + G4_BN - 1) / G4_BN;
  const int splits = SK ? p.splits : 1;
  const int ntiles = (int)gridDim.x / splits;
  const int split = splits > 1 ? (int)blockIdx.x / ntiles : 0;
  const int lid = splits > 1 ? (int)blockIdx.x % ntiles : xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(lid / tiles_n) * G4_BM;
  const int n0 = (lid % tiles_n) * G4_BN;
  const int K = p.K, N = p.N;
  const int64_t M = p.M;
  const int nfull = K / G4_BK;
  const bool has_ext = HF || p.A2 != nullptr;
  const int NTL_all = nfull + ((K % G4_BK) ? 1 : 0) + (has_ext ? 1 : 0);   // K-tiles of the product
  // this block's K-tiles [kt0, NTL): the whole product, or its split's range (never empty: the launcher sizes kt_per so)
  const int kt0 = splits > 1 ? split * p.kt_per : 0;
  const int NTL = splits > 1 ? (kt0 + p.kt_per < NTL_all ? kt0 + p.kt_per : NTL_all) : NTL_all;
  const int H = 4 * NTL;                                            // half-tiles (global numbering)
  const char* zp = zero_page_for(lane);

  // ------------------------------------------------------------------ DMA geometry (per lane)
  // k-contiguous image [128][64]: instruction ii of this wave covers image rows 16 w + 8 ii .. + 7
  const int c_pc = lane & 7;                                    // physical chunk this lane writes
  int c_lr[2], c_q[2];                                          // image row, logical chunk
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    c_lr[ii] = 16 * w + 8 * ii + (lane >> 3);
    c_q[ii] = c_pc ^ ((c_lr[ii] >> 1) & 7);
  }
  auto a_row = [&](int mh, int lr) -> int64_t {                 // global row of image row lr of A half mh (clamped)
    const int64_t gr = m0 + (lr >> 6) * 128 + mh * 64 + (lr & 63);
    return gr < M ? gr : M - 1;
  };
  auto b_col_nt = [&](int nh, int lr) -> int {                  // global column of image row lr of B half nh (clamped)
    const int gn = n0 + (lr >> 5) * 64 + nh * 32 + (lr & 31);
    return gn < N ? gn : N - 1;
  };
  // k-major image [64 k][128 n]: instruction ii covers k rows 8 w + 4 ii .. + 3
  const int m_kr0 = 8 * w + (lane >> 4);                        // k row of ii = 0 (ii = 1: + 4)
  const int m_f = (((lane >> 4) & 3) | ((w & 1) << 2)) << 1;    // swizzle of that k row (the same for ii = 1)
  const int m_q = (lane & 15) ^ m_f;                            // logical chunk
  auto b_col_nn = [&](int nh) -> int {                          // first global column of this lane's chunk (clamped)
    const int lc = 8 * m_q;
    const int gn = n0 + (lc >> 5) * 64 + nh * 32 + (lc & 31);
    return gn + 8 <= N ? gn : N - 8;
  };

  // per-lane source pointers of the full K-tiles, advanced per tile
  const bf16_t* pA[2][2];
  const bf16_t* pB[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      pA[hh][ii] = p.A + a_row(hh, c_lr[ii]) * p.lda + 8 * c_q[ii];
      if constexpr (NT) pB[hh][ii] = p.B + (int64_t)b_col_nt(hh, c_lr[ii]) * p.ldb + 8 * c_q[ii];
      else pB[hh][ii] = p.B + (int64_t)(m_kr0 + 4 * ii) * p.ldb + b_col_nn(hh);
    }
  const int64_t stepB = NT ? (int64_t)G4_BK : (int64_t)G4_BK * p.ldb;
  if (kt0 > 0) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) pA[hh][ii] += (int64_t)kt0 * G4_BK, pB[hh][ii] += (int64_t)kt0 * stepB;
  }

  // DMA of half-tile KIND of K-tile `tile`
  auto issue = [&](auto kind_c, int tile) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool IS_A = KIND == G4_A0 || KIND == G4_A1;
    constexpr int HH = (KIND == G4_A1 || KIND == G4_B1) ? 1 : 0;
    char* dst = smem + (tile & 1) * G4_BUF + g4_slot_off<KIND>() + (2 * w) * 1024;
    if (tile < nfull) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        if constexpr (IS_A) {
          dma16(pA[HH][ii], dst + ii * 1024);
          pA[HH][ii] += G4_BK;
        } else {
          dma16(pB[HH][ii], dst + ii * 1024);
          pB[HH][ii] += stepB;
        }
      }
      return;
    }
    // checked path: the K tail of the main operands, or the extension tile
    const bool ext = has_ext && tile == NTL_all - 1;
    const int k0 = ext ? 0 : nfull * G4_BK;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const void* src;
      if constexpr (IS_A) {
        const bf16_t* base = ext ? p.A2 : p.A;
        const int64_t ld = ext ? p.lda2 : p.lda;
        const int klim = ext ? 64 : K;
        const int kk = k0 + 8 * c_q[ii];
        // gemm4h: the extension's A operand is the projected tile in LDS; the DMA slot is filled with zeros only to keep
        // the counted waits uniform
        src = (kk < klim && !(HF && ext)) ? (const void*)(base + a_row(HH, c_lr[ii]) * ld + kk) : (const void*)zp;
      } else if constexpr (NT) {
        const bf16_t* base = ext ? p.B2 : p.B;
        const int64_t ld = ext ? p.ldb2 : p.ldb;
        const int klim = ext ? p.k2e : K;
        const int kk = k0 + 8 * c_q[ii];
        const int64_t eoff = (int64_t)b_col_nt(HH, c_lr[ii]) * ld + kk;
        // gemm4h: B2 = A as stored ([N][r], 2r-byte rows): pieces straddling r carry the next row's head (they meet the
        // exact zeros of H's ranks >= r); the piece that would cross the end of the buffer reads zeros, patched before use
        const bool ok = kk < klim && !(HF && ext && eoff + 8 > (int64_t)N * ld);
        src = ok ? (const void*)(base + eoff) : (const void*)zp;
      } else {
        const bf16_t* base = ext ? p.B2 : p.B;
        const int64_t ld = ext ? p.ldb2 : p.ldb;
        const int krows = ext ? p.k2 : K;
        const int gk = k0 + m_kr0 + 4 * ii;
        src = gk < krows ? (const void*)(base + (int64_t)gk * ld + b_col_nn(HH)) : (const void*)zp;
      }
      dma16(src, dst + ii * 1024);
    }
  };
  using KA0 = std::integral_constant<int, G4_A0>;
  using KB0 = std::integral_constant<int, G4_B0>;
  using KB1 = std::integral_constant<int, G4_B1>;
  using KA1 = std::integral_constant<int, G4_A1>;

  // ------------------------------------------------------------------ gemm4h: projection pass H^T = F' . A^T over all of K
  // A lighter pipeline of its own (HBM-bound: the row panel streams in once, the main loop below re-reads it from L2 /
  // the Infinity Cache): three 40-KiB buffers in the ring, two K-tiles in flight, every wave the same schedule.  Wave
  // (wr, wc) owns rows 128 wr .. + 127 x ranks 16 wc .. + 15 (32 accumulator registers, dead before the main loop starts).
  if constexpr (HF) {
    const int NTA = (K + G4_BK - 1) / G4_BK;
    const int f_row = 8 * w + (lane >> 3);     // NT: rank row of F; NN: k row of the padded [K][64] factor
    const int f_pc = lane & 7;
    const int f_q = NT ? (f_pc ^ ((f_row >> 1) & 7)) : (f_pc ^ ((((lane >> 4) & 1) | ((w & 1) << 1)) << 1));
    auto issue_a = [&](int tl) {
      char* buf = smem + (tl % 3) * G4_PA_BUF;
      const int k0 = tl * G4_BK;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const void* src = (k0 + 8 * c_q[ii] < K) ? (const void*)pA[hh][ii] : (const void*)zp;
          dma16(src, buf + hh * G4_HALF + (2 * w + ii) * 1024);
          pA[hh][ii] += G4_BK;
        }
      const void* fs;
      if constexpr (NT) {
        const int kk = k0 + 8 * f_q;
        fs = (f_row < p.r && kk < K) ? (const void*)(p.F + (int64_t)f_row * p.ldf + kk) : (const void*)zp;
      } else {
        // F = A as stored ([K][r], 2r-byte rows, r even): a 16-byte piece may carry the head of the next row -- those are
        // ranks >= r, masked when H is written -- and the ONE piece that would cross the end of the buffer (last row,
        // straddling piece) reads zeros and is patched below
        const int gk = k0 + f_row;
        const int64_t eoff = (int64_t)gk * p.ldf + 8 * f_q;
        const bool ok = gk < K && 8 * f_q < p.r && eoff + 8 <= (int64_t)K * p.ldf;
        fs = ok ? (const void*)(p.F + eoff) : (const void*)zp;
      }
      dma16(fs, buf + 2 * G4_HALF + w * 1024);
    };
    auto patch_f = [&](int tl) {   // after this wave's pieces of tile tl have landed, before the barrier that publishes them
      if constexpr (!NT) {
        const int gk = tl * G4_BK + f_row;
        const int64_t eoff = (int64_t)gk * p.ldf + 8 * f_q;
        if (gk < K && 8 * f_q < p.r && eoff + 8 > (int64_t)K * p.ldf) {
          u32x4 v;
          bf16_t* e = (bf16_t*)&v;
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = (8 * f_q + j < p.r) ? p.F[eoff + j] : (bf16_t)0.f;
          *(u32x4*)(smem + (tl % 3) * G4_PA_BUF + 2 * G4_HALF + w * 1024 + lane * 16) = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    };
    const uint32_t lbase = lds_addr(smem);
    const int fsw_a = (r16 >> 1) & 7;
    const int cha = (fsw_a & 4) | (g ^ (fsw_a & 3));
    const uint32_t ar0 = lbase + (uint32_t)((wr * 64 + r16) * 128 + cha * 16);
    const uint32_t ar1 = lbase + (uint32_t)((wr * 64 + r16) * 128 + (cha ^ 4) * 16);
    uint32_t fr0, fr1;
    if constexpr (NT) {
      fr0 = lbase + (uint32_t)(2 * G4_HALF + (wc * 16 + r16) * 128 + cha * 16);
      fr1 = lbase + (uint32_t)(2 * G4_HALF + (wc * 16 + r16) * 128 + (cha ^ 4) * 16);
    } else {
      const int qq = r16 >> 2, pp = r16 & 3;
      const int ff_sw = (((qq >> 1) & 1) | ((g & 1) << 1)) << 1;
      const int chunk = wc * 2 + (pp >> 1);
      fr0 = lbase + (uint32_t)(2 * G4_HALF + (8 * g + qq) * 128 + ((chunk ^ ff_sw) * 16) + 8 * (pp & 1));
      fr1 = fr0;
    }
    f32x4 hacc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) hacc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    issue_a(0);
    if (NTA > 1) issue_a(1);
#pragma unroll 1
    for (int tl = 0; tl < NTA; ++tl) {
      wait_groups<5>(NTA - 1 - tl < 1 ? NTA - 1 - tl : 1);
      if (tl == NTA - 1) patch_f(tl);   // the only tile that holds the last row of F
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (tl + 2 < NTA) issue_a(tl + 2);
      const uint32_t bo = (uint32_t)((tl % 3) * G4_PA_BUF);
      u32x4 xa[4][2], ffr[2];
      u32x2 fl[2], fh[2];
      g4_read_a<0>(xa, ar0 + bo, ar1 + bo);
      if constexpr (NT) {
        g4_rd128<0>(ffr[0], fr0 + bo);
        g4_rd128<0>(ffr[1], fr1 + bo);
      } else {
        g4_rdtr<0>(fl[0], fr0 + bo);
        g4_rdtr<512>(fh[0], fr0 + bo);
        g4_rdtr<4096>(fl[1], fr0 + bo);
        g4_rdtr<4096 + 512>(fh[1], fr0 + bo);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!NT) ffr[0] = join2(fl[0], fh[0]), ffr[1] = join2(fl[1], fh[1]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hacc[0][mt] = mfma16(ffr[ks], xa[mt][ks], hacc[0][mt]);
      __builtin_amdgcn_sched_barrier(0);
      g4_read_a<1>(xa, ar0 + bo, ar1 + bo);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hacc[1][mt] = mfma16(ffr[ks], xa[mt][ks], hacc[1][mt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // H^T tile (rows = rank 4 g + j of rank tile wc, column = token r16) -> bf16 -> the k-contiguous image the extension reads
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int lr = wr * 64 + mt * 16 + r16;
        const int q = wc * 2 + (g >> 1);
        char* dst = smem + G4_HIMG + mh * G4_HALF + lr * 128 + ((q ^ ((lr >> 1) & 7)) * 16) + (g & 1) * 8;
        f32x4 v = hacc[mh][mt];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (wc * 16 + 4 * g + j < p.r) ? v[j] * p.hscale : 0.f;   // ranks >= r: exact zeros
        *(u32x2*)dst = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // every read of the projection buffers is done: the ring belongs to the main loop
    __builtin_amdgcn_sched_barrier(0);
    // the main loop starts over at k = 0
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) pA[hh][ii] = p.A + a_row(hh, c_lr[ii]) * p.lda + 8 * c_q[ii];
  }

  // ------------------------------------------------------------------ fragment addresses (per lane)
  const uint32_t base = lds_addr(smem);
  const int fsw = (r16 >> 1) & 7;
  const int ch0 = (fsw & 4) | (g ^ (fsw & 3));                  // physical chunk of k-step 0 (k-step 1: ^ 4)
  uint32_t a_off[2], b_off[2];                                  // [ks] (NT) / [nt] (NN); buffer bit toggled per K-tile
  a_off[0] = base + (uint32_t)((wr * 64 + r16) * 128 + ch0 * 16);
  a_off[1] = base + (uint32_t)((wr * 64 + r16) * 128 + (ch0 ^ 4) * 16);
  if constexpr (NT) {
    b_off[0] = base + (uint32_t)(G4_OFF_B0 + (wc * 32 + r16) * 128 + ch0 * 16);
    b_off[1] = base + (uint32_t)(G4_OFF_B0 + (wc * 32 + r16) * 128 + (ch0 ^ 4) * 16);
  } else {
    const int qq = r16 >> 2, pp = r16 & 3;
    const int f = (qq | ((g & 1) << 2)) << 1;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int chunk = wc * 4 + nt * 2 + (pp >> 1);
      b_off[nt] = base + (uint32_t)(G4_OFF_B0 + (8 * g + qq) * 256 + ((chunk ^ f) * 16) + 8 * (pp & 1));
    }
  }

  f32x4 acc[2][4][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 af[4][2];           // [mt][ks]      x fragments of the current row half
  u32x4 bf[2][2][2];        // [nh][nt][ks]  W fragments of both column halves (k-contiguous image)
  u32x2 bl[2][2][2], bh[2][2][2];   // the same from the k-major image: low / high four k of every fragment

  auto read_a = [&](auto mh_c) { g4_read_a<decltype(mh_c)::value>(af, a_off[0], a_off[1]); };
  auto read_b = [&](auto nh_c) {
    constexpr int NH = decltype(nh_c)::value;
    if constexpr (NT) g4_read_b_nt<NH>(bf[NH], b_off[0], b_off[1]);
    else g4_read_b_nn<NH>(bl[NH], bh[NH], b_off[0], b_off[1]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // one phase of K-tile `tile`: fragment reads of the quadrant, DMA of half-tile P + 6, counted wait, barrier, 16 MFMAs, barrier
  auto phase = [&](auto ph_c, auto tail_c, int tile) {
    constexpr int PH = decltype(ph_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr int MH = PH >= 2 ? 1 : 0;
    constexpr int NH = (PH == 1 || PH == 2) ? 1 : 0;
    if constexpr (PH == 0) {
      read_b(I0{});
      __builtin_amdgcn_sched_barrier(0);
      read_a(I0{});
    } else if constexpr (PH == 1) {
      read_b(I1{});
    } else if constexpr (PH == 2) {
      read_a(I1{});
    }
    __builtin_amdgcn_sched_barrier(0);
    const int P = 4 * tile + PH;
    const int h = P + 6;
    if (!TAIL || h < H) {
      if constexpr (PH == 0) issue(KB1{}, h >> 2);
      else if constexpr (PH == 1) issue(KA1{}, h >> 2);
      else if constexpr (PH == 2) issue(KA0{}, h >> 2);
      else issue(KB0{}, h >> 2);
    }
    if constexpr (!TAIL) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    } else {
      int newer = H - 3 - P;
      newer = newer < 0 ? 0 : (newer > 4 ? 4 : newer);
      wait_groups<2>(newer);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          u32x4 wf;
          if constexpr (NT) wf = bf[NH][nt][ks];
          else wf = join2(bl[NH][nt][ks], bh[NH][nt][ks]);
          acc[MH][mt][NH][nt] = mfma16(wf, af[mt][ks], acc[MH][mt][NH][nt]);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PH == 3) {   // next K-tile: the other buffer
      a_off[0] ^= G4_BUF, a_off[1] ^= G4_BUF, b_off[0] ^= G4_BUF, b_off[1] ^= G4_BUF;
    }
  };
  using T0 = std::integral_constant<bool, false>;
  using T1 = std::integral_constant<bool, true>;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;

  // ------------------------------------------------------------------ prologue: half-tiles 0 .. 5 (of this block's K range)
  if (kt0 & 1) a_off[0] ^= G4_BUF, a_off[1] ^= G4_BUF, b_off[0] ^= G4_BUF, b_off[1] ^= G4_BUF;
  issue(KA0{}, kt0), issue(KB0{}, kt0), issue(KB1{}, kt0), issue(KA1{}, kt0);
  if (NTL - kt0 > 1) issue(KA0{}, kt0 + 1), issue(KB0{}, kt0 + 1);
  wait_groups<2>(NTL - kt0 > 1 ? 4 : 2);     // A0, B0 of the first K-tile have landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();        // ... everyone's
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave row runs one barrier behind the first
  __builtin_amdgcn_sched_barrier(0);

  int tile = kt0;
#pragma unroll 1
  for (; tile < NTL - 2; ++tile) {
    phase(P0{}, T0{}, tile);
    phase(P1{}, T0{}, tile);
    phase(P2{}, T0{}, tile);
    phase(P3{}, T0{}, tile);
  }
#pragma unroll 1
  for (; tile < NTL - 1; ++tile) {
    phase(P0{}, T1{}, tile);
    phase(P1{}, T1{}, tile);
    phase(P2{}, T1{}, tile);
    phase(P3{}, T1{}, tile);
  }
  // the last K-tile, outside the loop (gemm4h: the extension tile, with its operand switch and the end-of-buffer patch)
  if (tile < NTL) {
    if constexpr (HF) {
      if (tile == NTL - 1) {   // the extension tile: A fragments come from the projected tile
        a_off[0] = base + (uint32_t)(G4_HIMG + (wr * 64 + r16) * 128 + ch0 * 16);
        a_off[1] = base + (uint32_t)(G4_HIMG + (wr * 64 + r16) * 128 + (ch0 ^ 4) * 16);
        if constexpr (NT) {
          if (p.ldb2 != 64) {
            // B2 = raw A: rewrite the pieces that crossed the end of the buffer (every DMA has been issued by now).  Two
            // barriers: the wave rows are one barrier apart, and the other row's patch must be visible before the reads
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) {
                const int kk = 8 * c_q[ii];
                const int64_t eoff = (int64_t)b_col_nt(hh, c_lr[ii]) * p.ldb2 + kk;
                if (kk < p.k2e && eoff + 8 > (int64_t)N * p.ldb2) {
                  u32x4 v;
                  bf16_t* e = (bf16_t*)&v;
#pragma unroll
                  for (int j = 0; j < 8; ++j) e[j] = (kk + j < p.k2e) ? p.B2[eoff + j] : (bf16_t)0.f;
                  *(u32x4*)(smem + (tile & 1) * G4_BUF + (hh ? G4_OFF_B1 : G4_OFF_B0) + (2 * w + ii) * 1024 + lane * 16) = v;
                }
              }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    phase(P0{}, T1{}, tile);
    phase(P1{}, T1{}, tile);
    phase(P2{}, T1{}, tile);
    phase(P3{}, T1{}, tile);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // catch up: every wave has passed its last fragment read
  __builtin_amdgcn_sched_barrier(0);

  // ------------------------------------------------------------------ epilogue
  // acc[mh][mt][nh][nt][j] = C[row = 128 wr + 64 mh + 16 mt + r16][col = 64 wc + 32 nh + 16 nt + 4 g + j]
  float* sc = (float*)(smem + w * G4_SCR);
  const bool nts = p.nt_store != 0;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) *(f32x4*)(sc + r16 * G4_SCR_LD + nh * 32 + nt * 16 + 4 * g) = acc[mh][mt][nh][nt];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int r = pass * 8 + (lane >> 3), c = (lane & 7) * 8;
        const int64_t grow = m0 + wr * 128 + mh * 64 + mt * 16 + r;
        const int gcol = n0 + wc * 64 + c;
        if (grow < M && gcol < N) {
          float v[8];
          const f32x4 t0 = *(const f32x4*)(sc + r * G4_SCR_LD + c), t1 = *(const f32x4*)(sc + r * G4_SCR_LD + c + 4);
          if constexpr (SK) {
            // split-K: the fp32 sum of this block's K range, row-major [split][M][N] (256-byte row segments); alpha, beta, bias and
            // the rounding to bf16 happen once, in gemm4_splitk_reduce_kernel
            float* pd = p.partials + ((size_t)split * (size_t)M + (size_t)grow) * (size_t)N + gcol;
            *(f32x4*)pd = t0, *(f32x4*)(pd + 4) = t1;
            continue;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = t0[j] * p.alpha, v[4 + j] = t1[j] * p.alpha;
          bf16_t* dst = p.C + grow * p.ldc + gcol;
          if (p.beta != 0.f) {
            const u32x4 old = *(const u32x4*)dst;
            const bf16_t* o = (const bf16_t*)&old;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += p.beta * (float)o[j];
          }
          if (p.bias) {
            const u32x4 bv = *(const u32x4*)(p.bias + gcol);
            const bf16_t* b = (const bf16_t*)&bv;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)b[j];
          }
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) pk[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
          if (nts) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(pk) : "memory");
          else *(u32x4*)dst = pk;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  // gemm4h: the saved copy of the projection (h_save / dh: [M, 64], column 63 <- 1.0 when free -- the dbias column of the
  // weight-gradient kernels), written once per row panel, after the C stores
  if constexpr (HF) {
    if (n0 == 0 && p.Hout) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * G4_THREADS + t;
        const int half = idx >> 10, lr = (idx >> 3) & 127, pc = idx & 7;
        const int q = pc ^ ((lr >> 1) & 7);
        const int64_t grow = m0 + (lr >> 6) * 128 + half * 64 + (lr & 63);
        if (grow < M) {
          u32x4 v = *(const u32x4*)(smem + G4_HIMG + half * G4_HALF + lr * 128 + pc * 16);
          if (q == 7 && p.r < 64) v[3] = (v[3] & 0xffffu) | 0x3F800000u;
          *(u32x4*)(p.Hout + grow * 64 + q * 8) = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
static bool g4_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool gemm4_supported(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                     const void* B2, int64_t ldb2, const void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                     int dtype) {
  if (dtype != SOW_BF16 || !A || !B || !C) return false;
  if (M < 1 || N < 64 || K < 64) return false;
  if (K % 8 || N % 8 || lda % 8 || ldb % 8 || ldc % 8) return false;
  if (!g4_al16(A) || !g4_al16(B) || !g4_al16(C) || (bias && !g4_al16(bias))) return false;
  if (A2 && (!B2 || lda2 % 8 || ldb2 % 8 || !g4_al16(A2) || !g4_al16(B2))) return false;
  (void)nt;
  return true;
}

// C = alpha * sum_s partial[s] + beta * C + bias, 8 columns per thread (N % 8 == 0), splits added in order (deterministic)
__global__ __launch_bounds__(256) void gemm4_splitk_reduce_kernel(const float* __restrict__ part, int splits, int64_t M, int N,
                                                                  bf16_t* C, int64_t ldc, const bf16_t* bias, float alpha, float beta) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n8 = N / 8;
  if (idx >= M * n8) return;
  const int64_t row = idx / n8;
  const int col = (int)(idx % n8) * 8;
  const float* src = part + row * N + col;
  f32x4 a0 = __builtin_nontemporal_load((const f32x4*)src), a1 = __builtin_nontemporal_load((const f32x4*)(src + 4));
  for (int s = 1; s < splits; ++s) {
    const float* q = src + (size_t)s * (size_t)M * (size_t)N;
    a0 += __builtin_nontemporal_load((const f32x4*)q), a1 += __builtin_nontemporal_load((const f32x4*)(q + 4));
  }
  float v[8] = {a0[0] * alpha, a0[1] * alpha, a0[2] * alpha, a0[3] * alpha, a1[0] * alpha, a1[1] * alpha, a1[2] * alpha, a1[3] * alpha};
  bf16_t* dst = C + row * ldc + col;
  if (beta != 0.f) {
    const u32x4 old = *(const u32x4*)dst;
    const bf16_t* o = (const bf16_t*)&old;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += beta * (float)o[j];
  }
  if (bias) {
    const u32x4 bv = *(const u32x4*)(bias + col);
    const bf16_t* b = (const bf16_t*)&bv;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += (float)b[j];
  }
  u32x4 pk;
#pragma unroll
  for (int j = 0; j < 4; ++j) pk[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
  *(u32x4*)dst = pk;
}

// split-K plan for an M x N x K product with an optional K-extension tile: splits (1 = none) and K-tiles per split.  Taken
// when the output has at most 128 tiles (half of the CUs idle otherwise) and K >= 6144: the partial products cost a write and
// a read of S x M x N fp32 whatever K is -- ~33 us at 1024 x 4096, S = 4 -- so at K = 4096 the split (56-58 us) only ties with
// gemm3s's 256 small tiles (53-55 us) and is not worth its 64 MB of scratch; at K = 11008 it runs 90-93 us against 121-125
// (profiles/r03_gemm_splitk.txt).  Up to four splits, at least 8 K-tiles each.
static int gemm4_split_plan(int64_t M, int N, int K, bool has_ext, int* kt_per) {
  const int64_t tiles = (int64_t)ceil_div(M, G4_BM) * ceil_div(N, G4_BN);
  const int ntl = ceil_div(K, G4_BK) + (has_ext ? 1 : 0);
  *kt_per = ntl;
  if (tiles <= 0 || tiles > 128 || ntl < 96 || sw_on(SW_NO_SPLITK)) return 1;
  int s = (int)(256 / tiles);
  if (s > 4) s = 4;
  while (s > 1 && ntl / s < 8) --s;
  if (s <= 1) return 1;
  *kt_per = ceil_div(ntl, s);
  return ceil_div(ntl, *kt_per);
}
size_t gemm4_splitk_bytes(int64_t M, int N, int K, bool has_ext) {
  int kt_per;
  const int s = gemm4_split_plan(M, N, K, has_ext, &kt_per);
  if (s <= 1) return 0;
  return (size_t)s * (size_t)M * (size_t)N * sizeof(float) + 256;
}
int gemm4_splits(int64_t M, int N, int K, bool has_ext, const void* ws, size_t ws_bytes) {
  int kt_per;
  const int s = gemm4_split_plan(M, N, K, has_ext, &kt_per);
  return (s > 1 && ws && ws_bytes >= gemm4_splitk_bytes(M, N, K, has_ext)) ? s : 1;
}

int launch_gemm4(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                 const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                 float alpha, float beta, hipStream_t stream, void* ws, size_t ws_bytes) {
  Gemm4Params p;
  p.A = (const bf16_t*)A, p.B = (const bf16_t*)B, p.A2 = (const bf16_t*)A2, p.B2 = (const bf16_t*)B2;
  p.C = (bf16_t*)C, p.bias = (const bf16_t*)bias;
  p.M = M, p.lda = lda, p.ldb = ldb, p.lda2 = lda2, p.ldb2 = ldb2, p.ldc = ldc;
  p.N = N, p.K = K, p.k2 = k2 < 64 ? k2 : 64;
  p.k2e = 64;
  p.alpha = alpha, p.beta = beta;
  p.nt_store = SOW_GEMM_NT(M) ? 1 : 0;
  p.F = nullptr, p.ldf = 0, p.Hout = nullptr, p.r = 0, p.hscale = 0.f;
  p.splits = 1, p.kt_per = 0, p.partials = nullptr;
  const int64_t tiles = (int64_t)ceil_div(M, G4_BM) * ceil_div(N, G4_BN);
  if (tiles <= 0) return SOW_OK;
  if (tiles > 0x7fffffff) return SOW_ERR_SHAPE;
  int64_t grid = tiles;
  if (gemm4_splits(M, N, K, A2 != nullptr, ws, ws_bytes) > 1) {
    p.splits = gemm4_split_plan(M, N, K, A2 != nullptr, &p.kt_per);
    p.partials = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    grid = tiles * p.splits;
  }
  if (p.splits > 1 && nt) {
    SOW_SET_MAX_LDS_ONCE(G4_LDS, gemm4_kernel<true, false, true>);
    hipLaunchKernelGGL((gemm4_kernel<true, false, true>), dim3((unsigned)grid), dim3(G4_THREADS), G4_LDS, stream, p);
  } else if (p.splits > 1) {
    SOW_SET_MAX_LDS_ONCE(G4_LDS, gemm4_kernel<false, false, true>);
    hipLaunchKernelGGL((gemm4_kernel<false, false, true>), dim3((unsigned)grid), dim3(G4_THREADS), G4_LDS, stream, p);
  } else if (nt) {
    SOW_SET_MAX_LDS_ONCE(G4_LDS, gemm4_kernel<true, false>);
    hipLaunchKernelGGL((gemm4_kernel<true, false>), dim3((unsigned)grid), dim3(G4_THREADS), G4_LDS, stream, p);
  } else {
    SOW_SET_MAX_LDS_ONCE(G4_LDS, gemm4_kernel<false, false>);
    hipLaunchKernelGGL((gemm4_kernel<false, false>), dim3((unsigned)grid), dim3(G4_THREADS), G4_LDS, stream, p);
  }
  SOW_CHECK_LAUNCH();
  if (p.splits > 1) {
    const int64_t work = M * (N / 8);
    hipLaunchKernelGGL(gemm4_splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, p.partials, p.splits, M,
                       N, p.C, p.ldc, p.bias, p.alpha, p.beta);
    SOW_CHECK_LAUNCH();
  }
  return SOW_OK;
}

// ---- gemm4h: C = X . op(W) + H . op(G) + bias with H = hscale * X . op(F) computed in the kernel and saved -------------
// NN (forward: y = x W_acc + h B): W [K, N], F = A zero-padded to [K, 64] (ldf = 64), G = B [r, N].
// NT (backward: dX = dY W_acc^T + dh A^T): W [N, K], F = B [r, K], G = A zero-padded to [N, 64] (ldg = 64).
bool gemm4h_supported(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                      const void* G, int64_t ldg, const void* C, int64_t ldc, const void* bias, const void* H, int64_t M,
                      int N, int K, int r, int dtype) {
  if (dtype != SOW_BF16 || !X || !W || !F || !G || !C || !H) return false;
  if (sw_on(SW_NO_FUSED_H) || sw_on(SW_FORCE_GEMM_V1) || sw(SW_GEMM4) == 0 || sw_on(SW_NO_GEMM4H)) return false;
  if (r < 2 || r > 64 || (r & 1) || N < 64 || K < 64) return false;   // r even: A's 2r-byte rows stay 4-byte aligned
  const int tiles_n = ceil_div(N, G4_BN);
  // every column tile repeats the projection pass (the row panel streams in once more per tile): one or two tiles
  if (tiles_n > 2 || (int64_t)ceil_div(M, G4_BM) * tiles_n < 120) return false;
  if (K % 8 || N % 8 || ldx % 8 || ldw % 8 || ldc % 8) return false;
  if (!g4_al16(X) || !g4_al16(W) || !g4_al16(C) || !g4_al16(H) || (bias && !g4_al16(bias))) return false;
  auto al4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; };
  if (nt) {   // F = B [r, K] (16-byte pieces along rows), G = A [N, r] as stored (or zero-padded to 64 columns)
    if (ldf % 8 || !g4_al16(F) || !al4(G) || (ldg != r && ldg != 64)) return false;
  } else {    // F = A [K, r] as stored (or [K, 64]), G = B [r, N]
    if (ldg % 8 || !g4_al16(G) || !al4(F) || (ldf != r && ldf != 64)) return false;
  }
  return true;
}

int launch_gemm4h(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                  const void* G, int64_t ldg, void* C, int64_t ldc, const void* bias, void* H, int64_t M, int N, int K,
                  int r, float hscale, hipStream_t stream) {
  Gemm4Params p;
  p.A = (const bf16_t*)X, p.B = (const bf16_t*)W, p.A2 = nullptr, p.B2 = (const bf16_t*)G;
  p.C = (bf16_t*)C, p.bias = (const bf16_t*)bias;
  p.M = M, p.lda = ldx, p.ldb = ldw, p.lda2 = 64, p.ldb2 = ldg, p.ldc = ldc;
  p.N = N, p.K = K, p.k2 = r < 64 ? r : 64;
  p.k2e = nt ? (ldg == 64 ? 64 : r) : 64;
  p.alpha = 1.f, p.beta = 0.f;
  p.nt_store = SOW_GEMM_NT(M) ? 1 : 0;
  p.F = (const bf16_t*)F, p.ldf = ldf, p.Hout = (bf16_t*)H, p.r = r, p.hscale = hscale;
  p.splits = 1, p.kt_per = 0, p.partials = nullptr;
  const int64_t tiles = (int64_t)ceil_div(M, G4_BM) * ceil_div(N, G4_BN);
  if (tiles <= 0) return SOW_OK;
  if (tiles > 0x7fffffff) return SOW_ERR_SHAPE;
  if (nt) {
    SOW_SET_MAX_LDS_ONCE(G4_LDS_H, gemm4_kernel<true, true>);
    hipLaunchKernelGGL((gemm4_kernel<true, true>), dim3((unsigned)tiles), dim3(G4_THREADS), G4_LDS_H, stream, p);
  } else {
    SOW_SET_MAX_LDS_ONCE(G4_LDS_H, gemm4_kernel<false, true>);
    hipLaunchKernelGGL((gemm4_kernel<false, true>), dim3((unsigned)tiles), dim3(G4_THREADS), G4_LDS_H, stream, p);
  }
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
