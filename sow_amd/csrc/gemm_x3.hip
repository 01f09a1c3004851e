// fp32 GEMM on the bf16 matrix pipe:  C[M,N] = alpha * op(A) . op(B) + beta * C + bias[N]  for float tensors.
//
// Same contract and tile shape as gemm_kernel<float, ...> (gemm.hip: 128x128 tile, 2x2 waves, global -> register -> LDS
// staging with the next tile's loads in flight during the MFMAs), but every operand element is split ONCE, on its way
// into LDS, into three bf16 planes that hold its 24 mantissa bits exactly (lds_dma.hpp: split3), and the product is
//     a b ~= ah bh + ah bm + am bh + am bm + ah bl + al bh      (fp32 accumulation inside the MFMA, error <= 2^-23 |a b|)
// i.e. six v_mfma_f32_32x32x16_bf16 (32 cycles, 16 k) instead of eight v_mfma_f32_32x32x2_f32 (64 cycles, 2 k): 2.7 x less
// matrix-pipe time.  Unlike the streaming kernels (chain2f, skinny-TN), where every fragment is used once and the split
// is paid per use, a GEMM tile re-uses each staged element 128 times, so the split is noise.
// This is the product behind the DENSE accumulator in fp32 -- BASELINE config 4 (roberta-base fine-tuning runs fp32 with
// decompose='keep': y = x . W_acc is 2 d_in d_out flops per token, sow.py:111-112) -- and behind accumulate()'s Q . R /
// R = Q^T W products (utils.py:19-22).  LDS: 2 operands x 3 planes x [128 rows][32 k] bf16 = 48 KiB, two workgroups per CU.
#include "kernels.hpp"
#include "epilogue.hpp"
#include "lds_dma.hpp"

namespace sow {

constexpr int GX_BM = 128, GX_BN = 128, GX_BK = 32;
constexpr int GX_PLANE = 128 * GX_BK * 2;   // 8 KiB: one bf16 plane of an operand tile

struct GemmX3Params {
  const float *A, *B;
  float* C;
  const float* bias;
  int64_t M, lda, ldb, ldc;
  int N, K;
  float alpha, beta;
};

// Stages a [128 rows x 32 k] fp32 operand tile through registers and stores it as three bf16 planes.
// Source element (row, k) = KC ? P[row * ld + k] : P[k * ld + row]; 16-byte vectors along the storage-contiguous axis.
template <bool KC> struct X3Tile {
  f32x4 v[4];
  __device__ __forceinline__ void load(const float* P, int64_t ld, int64_t row0, int64_t nrows, int k0, int K, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i;
      if constexpr (KC) {
        const int row = idx >> 3, c = idx & 7;   // 8 vectors of 4 k per row
        const int64_t gr = row0 + row;
        const int gk = k0 + 4 * c;
        v[i] = (gr < nrows && gk < K) ? *(const f32x4*)(P + gr * ld + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const int k = idx >> 5, c = idx & 31;    // 32 vectors of 4 rows per k
        const int64_t gr = row0 + 4 * c;
        v[i] = (gr < nrows && k0 + k < K) ? *(const f32x4*)(P + (int64_t)(k0 + k) * ld + gr) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __device__ __forceinline__ void store(char* img, int t) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i;
      if constexpr (KC) {
        // four consecutive k of one row: half of a 16-byte chunk (8 k) per plane
        const int row = idx >> 3, c = idx & 7;
        uint32_t h0, m0, l0, h1, m1, l1;
        split3(v[i][0], v[i][1], h0, m0, l0);
        split3(v[i][2], v[i][3], h1, m1, l1);
        char* dst = img + bf16_img_off<GX_BK>(row, c >> 1) + (c & 1) * 8;
        *(u32x2*)(dst) = u32x2{h0, h1};
        *(u32x2*)(dst + GX_PLANE) = u32x2{m0, m1};
        *(u32x2*)(dst + 2 * GX_PLANE) = u32x2{l0, l1};
      } else {
        // four consecutive rows (m or n) at one k: the plane images of a k-major operand stay k-major, [32 k][128] bf16 with
        // 256-byte rows (16-byte chunk c at c ^ ((k & 3) << 2)), and are read transposed (ds_read_b64_tr_b16) as in gemm2.hip
        const int k = idx >> 5, c = idx & 31;
        uint32_t h0, m0, l0, h1, m1, l1;
        split3(v[i][0], v[i][1], h0, m0, l0);
        split3(v[i][2], v[i][3], h1, m1, l1);
        char* dst = img + k * 256 + (((c >> 1) ^ ((k & 3) << 2)) * 16) + (c & 1) * 8;
        *(u32x2*)(dst) = u32x2{h0, h1};
        *(u32x2*)(dst + GX_PLANE) = u32x2{m0, m1};
        *(u32x2*)(dst + 2 * GX_PLANE) = u32x2{l0, l1};
      }
    }
  }
};

// fragments of the two 32-row tiles at row0, row0 + 32 of an operand, k-step ks, all three planes
template <bool KC> __device__ __forceinline__ void load_frags(const char* img, int row0, int ks, int lane, u32x4 (&f)[2][3]) {
  if constexpr (KC) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        f[tl][pl] = *(const u32x4*)(img + pl * GX_PLANE + bf16_img_off<GX_BK>(row0 + tl * 32 + li, 2 * ks + lh));
  } else {
    const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3;
    const int r1 = 8 * (g >> 1) + q;
    const uint32_t base = lds_addr(img) + (uint32_t)(ks * 4096 + r1 * 256);
    u32x2 lo[2][3], hi[2][3];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const int col = row0 + tl * 32 + 16 * (g & 1) + 4 * pp;
      const uint32_t ad = base + (uint32_t)((((col >> 3) ^ ((r1 & 3) << 2)) * 16) + (col & 7) * 2);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        DS_READ_TR(lo[tl][pl], ad + (uint32_t)(pl * GX_PLANE), 0);
        DS_READ_TR(hi[tl][pl], ad + (uint32_t)(pl * GX_PLANE), 1024);
      }
    }
    // the transposed reads are inline asm: ONE wait for all twelve, with the results as read-write operands so that the
    // compiler cannot move a use (or a register copy) of a pending result above it
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lo[0][0]), "+v"(hi[0][0]), "+v"(lo[0][1]), "+v"(hi[0][1]), "+v"(lo[0][2]), "+v"(hi[0][2]), "+v"(lo[1][0]),
                   "+v"(hi[1][0]), "+v"(lo[1][1]), "+v"(hi[1][1]), "+v"(lo[1][2]), "+v"(hi[1][2])
                 :
                 : "memory");
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) f[tl][pl] = join2(lo[tl][pl], hi[tl][pl]);
  }
}

// TA: A given transposed (stored [K, M]); TB: B given as [N, K].
template <bool TA, bool TB> __global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const GemmX3Params p) {
  using TileA = X3Tile<!TA>;
  using TileB = X3Tile<TB>;
  constexpr int SCR = EpiScratch<2>::FLOATS * 4;
  constexpr int AB = 6 * GX_PLANE;   // 48 KiB
  constexpr int LDS = AB > 4 * SCR ? AB : 4 * SCR;
  __shared__ __attribute__((aligned(16))) char smem[LDS];
  char* As = smem;
  char* Bs = smem + 3 * GX_PLANE;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = (p.N + GX_BN - 1) / GX_BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(lid / tiles_n) * GX_BM;
  const int n0 = (lid % tiles_n) * GX_BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  TileA ta;
  TileB tb;
  const int nk = (p.K + GX_BK - 1) / GX_BK;
  ta.load(p.A, p.lda, m0, p.M, 0, p.K, t);
  tb.load(p.B, p.ldb, n0, p.N, 0, p.K, t);
  for (int kt = 0; kt < nk; ++kt) {
    ta.store(As, t);
    tb.store(Bs, t);
    __syncthreads();
    if (kt + 1 < nk) {
      ta.load(p.A, p.lda, m0, p.M, (kt + 1) * GX_BK, p.K, t);
      tb.load(p.B, p.ldb, n0, p.N, (kt + 1) * GX_BK, p.K, t);
    }
#pragma unroll
    for (int ks = 0; ks < GX_BK / 16; ++ks) {
      u32x4 af[2][3], bfr[2][3];
      load_frags<!TA>(As, wm * 64, ks, lane, af);
      load_frags<TB>(Bs, wn * 64, ks, lane, bfr);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma_x3(af[a], bfr[b], acc[a][b]);
    }
    __syncthreads();
  }
  float* scratch = (float*)(smem + w * SCR);
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
    wave_store_tiles<float, 2, true>(acc[mh], scratch, p.C, p.ldc, m0 + wm * 64 + mh * 32, n0 + wn * 64, p.M, p.N, p.alpha,
                                     p.beta, p.bias, lane);
}

// the vector-aligned case only (16-byte rows along the storage-contiguous axis of both operands and of C)
int launch_gemm_x3(const void* A, int64_t lda, bool transA, const void* B, int64_t ldb, bool transB, void* C, int64_t ldc,
                   const void* bias, int64_t M, int N, int K, float alpha, float beta, hipStream_t stream) {
  GemmX3Params p;
  p.A = (const float*)A, p.B = (const float*)B, p.C = (float*)C, p.bias = (const float*)bias;
  p.M = M, p.N = N, p.K = K, p.lda = lda, p.ldb = ldb, p.ldc = ldc;
  p.alpha = alpha, p.beta = beta;
  const int64_t tiles = (int64_t)ceil_div(M, GX_BM) * ceil_div(N, GX_BN);
  if (tiles <= 0) return SOW_OK;
  if (tiles > 0x7fffffff) return SOW_ERR_SHAPE;
  if (transA) {
    if (transB) hipLaunchKernelGGL((gemm_x3_kernel<true, true>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_x3_kernel<true, false>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
  } else {
    if (transB) hipLaunchKernelGGL((gemm_x3_kernel<false, true>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_x3_kernel<false, false>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
  }
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
