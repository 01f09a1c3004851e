// General dense MFMA GEMM  C[M,N] = alpha * op(A) . op(B) + beta * C + bias[N]
//
// Used for the parts of the SoW path that are plain matrix products:
//   * dense accumulator term  y = x . W_acc            (sow.py:111-112)   and dX += dY . W_acc^T;
//   * accumulate():  W_acc += scale * A . B            (sow.py:131-140),  Q . R materialisation;
//   * truncated QR:  R[:vr, :] = Q[:, :vr]^T . W       (utils.py:19-22);
//   * TT reconstruct / TensorTrainLinear chain contractions (tt.py:213-237).
// Row-major operands; op = identity or transpose selected per operand, so any of NN / NT / TN / TT
// is one instantiation of the operand loader (k-contiguous or k-major storage).
// 128x128 tile, 256 threads (2x2 waves, 64x64 per wave = 4 MFMA 32x32 tiles), BK = 64 (bf16) / 32 (f32),
// global -> register -> LDS staging with the next tile's loads in flight during the MFMAs, XCD-aware
// tile order (N fastest so that the tiles sharing an A row-panel share an L2).
#include "kernels.hpp"
#include "epilogue.hpp"

namespace sow {

struct GemmParams {
  const void *A, *B;
  void* C;
  const void* bias;
  int64_t M, lda, ldb, ldc;
  int N, K;
  float alpha, beta;
  int vecA, vecB, vecC;
};

constexpr int G_BM = 128, G_BN = 128;
template <typename T> struct GemmCfg;
template <> struct GemmCfg<bf16_t> {
  static constexpr int BK = 64;
};
template <> struct GemmCfg<float> {
  static constexpr int BK = 32;
};

// Stages a [128 rows x BK] operand tile.  Source element (row, k) = KC ? P[row*ld + k] : P[k*ld + row].
template <typename T, bool KC, bool FAST> struct OperandTile {
  static constexpr bool F32 = sizeof(T) == 4;
  static constexpr int BK = GemmCfg<T>::BK;
  static constexpr int VE = DT<T>::VE;
  static constexpr int BYTES = F32 ? (KC ? 128 * (BK + 1) * 4 : BK * 128 * 4) : 128 * BK * 2;
  static constexpr int NV = 128 * BK / VE / 256;  // 4
  static constexpr int NE = 128 * BK / 256;       // 32 bf16 / 16 f32
  u32x4 v[FAST ? NV : 1];
  uint32_t d[(FAST && !KC && !F32) ? 2 : 1][8];
  T e[FAST ? 1 : NE];

  __device__ __forceinline__ void load(const T* P, int64_t ld, int64_t row0, int64_t nrows, int k0, int K, int t) {
    if constexpr (FAST) {  // 16-byte vectors along the storage-contiguous axis
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = t + 256 * i;
        if constexpr (KC) {
          const int row = idx / (BK / VE), c = idx % (BK / VE);
          const int64_t gr = row0 + row;
          const int gk = k0 + c * VE;
          v[i] = (gr < nrows && gk < K) ? *(const u32x4*)(P + gr * ld + gk) : u32x4{0, 0, 0, 0};
        } else if constexpr (F32) {
          const int k = idx >> 5, c = idx & 31;
          const int64_t gr = row0 + c * 4;
          v[i] = (gr < nrows && k0 + k < K) ? *(const u32x4*)(P + (int64_t)(k0 + k) * ld + gr) : u32x4{0, 0, 0, 0};
        }
      }
      if constexpr (!KC && !F32) {
        // bf16 k-major: dword pairs along rows, 8 k per item, two items per thread
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int item = t + 256 * it, rp = item & 63, ko = item >> 6;
          const int64_t gr = row0 + 2 * rp;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int gk = k0 + ko * 8 + j;
            d[it][j] = (gr < nrows && gk < K) ? *(const uint32_t*)(P + (int64_t)gk * ld + gr) : 0u;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int idx = t + 256 * i;
        const int row = KC ? idx / BK : idx & 127;
        const int k = KC ? idx % BK : idx >> 7;
        const int64_t gr = row0 + row;
        const int gk = k0 + k;
        e[i] = (gr < nrows && gk < K) ? (KC ? P[gr * ld + gk] : P[(int64_t)gk * ld + gr]) : from_f32<T>(0.f);
      }
    }
  }

  __device__ __forceinline__ void store(char* img, int t) const {
    if constexpr (FAST) {
      if constexpr (KC) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int idx = t + 256 * i;
          const int row = idx / (BK / VE), c = idx % (BK / VE);
          if constexpr (F32) {
            float* dst = (float*)img + row * (BK + 1) + c * 4;
            const float* s = (const float*)&v[i];
            dst[0] = s[0], dst[1] = s[1], dst[2] = s[2], dst[3] = s[3];
          } else {
            *(u32x4*)(img + bf16_img_off<BK>(row, c)) = v[i];
          }
        }
      } else if constexpr (F32) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int idx = t + 256 * i;
          const int k = idx >> 5, c = idx & 31;
          *(u32x4*)(img + (k * 128 + c * 4) * 4) = v[i];
        }
      } else {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int item = t + 256 * it, rp = item & 63, ko = item >> 6;
          u32x4 c0, c1;
          transpose_8x2(d[it], c0, c1);
          *(u32x4*)(img + bf16_img_off<BK>(2 * rp, ko)) = c0;
          *(u32x4*)(img + bf16_img_off<BK>(2 * rp + 1, ko)) = c1;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int idx = t + 256 * i;
        const int row = KC ? idx / BK : idx & 127;
        const int k = KC ? idx % BK : idx >> 7;
        if constexpr (F32) {
          if constexpr (KC)
            ((float*)img)[row * (BK + 1) + k] = e[i];
          else
            ((float*)img)[k * 128 + row] = e[i];
        } else {
          *(bf16_t*)(img + bf16_img_off<BK>(row, k >> 3) + (k & 7) * 2) = e[i];
        }
      }
    }
  }
};

template <typename T, bool KC> __device__ __forceinline__ auto frag(const char* img, int row, int ks, int lh) {
  constexpr int BK = GemmCfg<T>::BK;
  if constexpr (sizeof(T) == 4) {
    const float* f = (const float*)img;
    return KC ? f[row * (BK + 1) + 2 * ks + lh] : f[(2 * ks + lh) * 128 + row];
  } else {
    return *(const bf16x8*)(img + bf16_img_off<BK>(row, 2 * ks + lh));
  }
}

// TA: A given transposed (stored [K, M]); TB: B given as [N, K].
template <typename T, bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int BK = GemmCfg<T>::BK;
  using TileA = OperandTile<T, !TA, FAST>;  // A k-contiguous unless transposed
  using TileB = OperandTile<T, TB, FAST>;   // B k-contiguous only when given as [N, K]
  constexpr int SCR = EpiScratch<2>::FLOATS * 4;
  constexpr int AB = ((TileA::BYTES + 15) & ~15) + TileB::BYTES;
  constexpr int LDS = AB > 4 * SCR ? AB : 4 * SCR;
  __shared__ __attribute__((aligned(16))) char smem[LDS];
  char* As = smem;
  char* Bs = smem + ((TileA::BYTES + 15) & ~15);

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = (p.N + G_BN - 1) / G_BN;
  const int nwg = gridDim.x;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int64_t m0 = (int64_t)(lid / tiles_n) * G_BM;
  const int n0 = (lid % tiles_n) * G_BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  TileA ta;
  TileB tb;
  const T* A = (const T*)p.A;
  const T* B = (const T*)p.B;
  const int nk = (p.K + BK - 1) / BK;
  ta.load(A, p.lda, m0, p.M, 0, p.K, t);
  tb.load(B, p.ldb, n0, p.N, 0, p.K, t);
  for (int kt = 0; kt < nk; ++kt) {
    ta.store(As, t);
    tb.store(Bs, t);
    __syncthreads();
    if (kt + 1 < nk) {
      ta.load(A, p.lda, m0, p.M, (kt + 1) * BK, p.K, t);
      tb.load(B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, t);
    }
#pragma unroll
    for (int ks = 0; ks < (F32 ? BK / 2 : BK / 16); ++ks) {
      const auto a0 = frag<T, !TA>(As, wm * 64 + li, ks, lh);
      const auto a1 = frag<T, !TA>(As, wm * 64 + 32 + li, ks, lh);
      const auto b0 = frag<T, TB>(Bs, wn * 64 + li, ks, lh);
      const auto b1 = frag<T, TB>(Bs, wn * 64 + 32 + li, ks, lh);
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  float* scratch = (float*)(smem + w * SCR);
  T* Cp = (T*)p.C;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
    wave_store_tiles<T, 2, FAST>(acc[mh], scratch, Cp, p.ldc, m0 + wm * 64 + mh * 32, n0 + wn * 64, p.M, p.N, p.alpha,
                                 p.beta, (const T*)p.bias, lane);
}

template <typename T, bool TA, bool TB> static int launch_gemm_tt(const GemmParams& p, hipStream_t stream) {
  const int64_t tiles = (int64_t)ceil_div(p.M, G_BM) * ceil_div(p.N, G_BN);
  if (tiles <= 0) return SOW_OK;
  if (tiles > 0x7fffffff) return SOW_ERR_SHAPE;
  if (p.vecA && p.vecB && p.vecC)
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, true>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, false>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
static bool al4(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; }

// C[M,N] = alpha * op(A) op(B) + beta * C + bias.   transA: A stored [K,M]; transB: B stored [N,K].
int launch_gemm(const void* A, int64_t lda, bool transA, const void* B, int64_t ldb, bool transB, void* C, int64_t ldc,
                const void* bias, int64_t M, int N, int K, float alpha, float beta, int dtype, hipStream_t stream) {
  if (!A || !B || !C) return SOW_ERR_NULL;
  if (M < 0 || N < 0 || K < 0) return SOW_ERR_SHAPE;
  if (M == 0 || N == 0) return SOW_OK;
  GemmParams p;
  p.A = A, p.B = B, p.C = C, p.bias = bias;
  p.M = M, p.N = N, p.K = K, p.lda = lda, p.ldb = ldb, p.ldc = ldc;
  p.alpha = alpha, p.beta = beta;
  const int ve = dtype == SOW_F32 ? 4 : 8;
  // mode 2 = vector/dword staging, 0 = scalar.  k-contiguous operands need 16-byte aligned rows along k;
  // k-major operands need aligned rows along m/n (f32: 16 bytes; bf16: dword pairs -> even extents).
  auto mode = [&](const void* ptr, int64_t ld, bool kcontig, int64_t rows) -> int {
    if (kcontig) return (ld % ve == 0 && K % ve == 0 && al16(ptr)) ? 2 : 0;
    if (dtype == SOW_F32) return (ld % 4 == 0 && rows % 4 == 0 && al16(ptr)) ? 2 : 0;
    return (ld % 2 == 0 && rows % 2 == 0 && al4(ptr)) ? 2 : 0;
  };
  p.vecA = mode(A, lda, !transA, M);
  p.vecB = mode(B, ldb, transB, N);
  p.vecC = (ldc % ve == 0 && N % ve == 0 && al16(C) && (!bias || al16(bias))) ? 1 : 0;
#define SOW_GEMM_DISPATCH(T)                                                    \
  if (transA)                                                                   \
    return transB ? launch_gemm_tt<T, true, true>(p, stream) : launch_gemm_tt<T, true, false>(p, stream); \
  else                                                                          \
    return transB ? launch_gemm_tt<T, false, true>(p, stream) : launch_gemm_tt<T, false, false>(p, stream);
  if (dtype == SOW_BF16) {
    SOW_GEMM_DISPATCH(bf16_t)
  } else if (dtype == SOW_F32) {
    // vector-aligned fp32 products run on the bf16 matrix pipe as 3 x bf16 splits (gemm_x3.hip; F32_EXACT switch: fp32 MFMA)
    if (p.vecA && p.vecB && p.vecC && K >= 16 && !sw_on(SW_F32_EXACT))
      return launch_gemm_x3(A, lda, transA, B, ldb, transB, C, ldc, bias, M, N, K, alpha, beta, stream);
    SOW_GEMM_DISPATCH(float)
  }
#undef SOW_GEMM_DISPATCH
  return SOW_ERR_DTYPE;
}

}  // namespace sow
