// Fused low-rank chain  Y = beta*Y + ((X . F1) * colscale) . F2 + bias,  H = X . F1 saved.
//
// One kernel serves both directions of SoWLinear (reference: tn_gradient/layer/sow.py:107-126 and
// the autograd backward of it):
//   forward  (BWD=false): X = x [T,d_in],  F1 = A stored [d_in, r] (k-major),
//                         F2 = B stored [r, d_out];  H = s*x.A saved (for dB).
//   backward (BWD=true):  X = dY [T,d_out], F1 = B^T given as stored [r, d_out],
//                         F2 = A^T given as stored [d_in, r];  H = s*dY.B^T saved (for dA).
// The live factors are scaled by `scale` (sow.py:117-121); a frozen low-rank accumulator (never scaled,
// sow.py:110) is a second call of the same kernel with scale = 1 and beta chaining the two.  r <= 64.
//
// Data flow per workgroup (64 token rows, 256 threads = 4 waves):
//   phase 1: K-loop over D1; X tile and F1 tile staged global -> registers -> LDS (next tile's
//            loads are issued before the current tile's MFMAs), H[64,64] accumulated by MFMA;
//   hand-off: H scaled, rounded to T, written to LDS (A operand of phase 2) and to Hsave;
//   phase 2: N-loop over D2; F2 chunk staged to LDS, Y chunk = H . F2chunk by MFMA, written through
//            the wave-private scratch epilogue as 16-byte row segments.
// X is read once, Y written once (plus one read when beta != 0); H never round-trips HBM inside
// the kernel.  Bound: HBM.
#include <stdlib.h>

#include "kernels.hpp"
#include "epilogue.hpp"

namespace sow {

template <typename T> struct ChainCfg;
template <> struct ChainCfg<bf16_t> {
  static constexpr int BK = 64, BN = 128, NT = 2;
};
template <> struct ChainCfg<float> {
  static constexpr int BK = 32, BN = 64, NT = 1;
};

constexpr int CH_BM = 64;
constexpr int CH_RP = 64;

// ---- LDS layout (bytes) -------------------------------------------------------------------
// phase 1: Xs | F1s            phase 2: Hs | F2s | scratch[4 waves]
template <typename T, bool BWD> struct ChainLds {
  using C = ChainCfg<T>;
  static constexpr bool F32 = sizeof(T) == 4;
  static constexpr int XS = F32 ? CH_BM * (C::BK + 1) * 4 : CH_BM * C::BK * 2;
  static constexpr int F1S = F32 ? (BWD ? CH_RP * (C::BK + 1) * 4 : C::BK * CH_RP * 4) : CH_RP * C::BK * 2;
  static constexpr int P1 = XS + F1S;
  static constexpr int HS = F32 ? CH_BM * (CH_RP + 1) * 4 : CH_BM * CH_RP * 2;
  static constexpr int F2S = F32 ? (BWD ? C::BN * (CH_RP + 1) * 4 : CH_RP * C::BN * 4) : C::BN * CH_RP * 2;
  static constexpr int SCR = EpiScratch<C::NT>::FLOATS * 4;
  static constexpr int align16(int v) { return (v + 15) & ~15; }
  static constexpr int OFF_F1S = align16(XS);
  static constexpr int OFF_HS = 0;
  static constexpr int OFF_F2S = align16(HS);
  static constexpr int OFF_SCR = OFF_F2S + align16(F2S);
  static constexpr int P2 = OFF_SCR + 4 * SCR;
  static constexpr int BYTES = (P1 > P2 ? P1 : P2) + 16;
};

// ---- factor element fetch ---------------------------------------------------------------------
// F1: fwd stored [D1, r] (k-major), bwd stored [r, D1];  F2: fwd stored [r, D2], bwd stored [D2, r].
template <typename T, bool BWD> __device__ __forceinline__ T f1_fetch(const ChainParams& p, int kg, int r) {
  if (kg >= p.D1 || r >= p.rb) return from_f32<T>(0.f);
  const T* base = (const T*)p.F1b;
  return BWD ? base[(int64_t)r * p.ldf1b + kg] : base[(int64_t)kg * p.ldf1b + r];
}
template <typename T, bool BWD> __device__ __forceinline__ T f2_fetch(const ChainParams& p, int r, int ng) {
  if (ng >= p.D2 || r >= p.rb) return from_f32<T>(0.f);
  const T* base = (const T*)p.F2b;
  return BWD ? base[(int64_t)ng * p.ldf2b + r] : base[(int64_t)r * p.ldf2b + ng];
}
// dword (2 x bf16) fetches for the fast paths (rb, leading dims, D1, D2 even; 4-byte aligned bases)
template <bool BWD> __device__ __forceinline__ uint32_t f1_fetch2(const ChainParams& p, int kg, int r) {
  // !BWD: elements (kg, r), (kg, r+1) ; BWD: elements (r, kg), (r, kg+1)
  if (kg >= p.D1 || r >= p.rb) return 0u;
  const bf16_t* base = (const bf16_t*)p.F1b;
  const bf16_t* q = BWD ? base + (int64_t)r * p.ldf1b + kg : base + (int64_t)kg * p.ldf1b + r;
  return *(const uint32_t*)q;
}
template <bool BWD> __device__ __forceinline__ uint32_t f2_fetch2(const ChainParams& p, int r, int ng) {
  // !BWD: elements (r, ng), (r, ng+1) ; BWD: elements (ng, r), (ng, r+1)
  if (ng >= p.D2 || r >= p.rb) return 0u;
  const bf16_t* base = (const bf16_t*)p.F2b;
  const bf16_t* q = BWD ? base + (int64_t)ng * p.ldf2b + r : base + (int64_t)r * p.ldf2b + ng;
  return *(const uint32_t*)q;
}

// =================================================================================================
template <typename T, bool BWD, bool VEC, bool FASTF>
__global__ __launch_bounds__(256, 2) void chain_kernel(const ChainParams p) {
  using C = ChainCfg<T>;
  using L = ChainLds<T, BWD>;
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int BK = C::BK, BN = C::BN, NT = C::NT, VE = DT<T>::VE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;
  char* F1s = smem + L::OFF_F1S;
  char* Hs = smem + L::OFF_HS;
  char* F2s = smem + L::OFF_F2S;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * CH_BM;
  const T* X = (const T*)p.X;
  const int rtot = p.rb;
  static_assert(!(F32 && FASTF), "dword factor loaders are bf16 only");

  // ------------------------------------------------------------------ phase 1 staging registers
  constexpr int XV = CH_BM * BK / VE / 256;       // 16-byte vectors of X per thread (2)
  constexpr int XE = CH_BM * BK / 256;            // scalar elements of X per thread
  constexpr int FE = CH_RP * BK / 256;            // factor elements per thread (16 bf16 / 8 f32)
  u32x4 xv[VEC ? XV : 1];
  T xe[VEC ? 1 : XE];
  T fe[FASTF ? 1 : FE];
  uint32_t fd[FASTF ? 8 : 1];

  auto load_x = [&](int k0) {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int v = t + 256 * i, row = v >> 3, c = v & 7;
        const int gk = k0 + c * VE;
        const int64_t gm = m0 + row;
        if (gm < p.M && gk < p.D1)
          xv[i] = *(const u32x4*)(X + gm * p.ldx + gk);
        else
          xv[i] = u32x4{0, 0, 0, 0};
      }
    } else {
#pragma unroll
      for (int i = 0; i < XE; ++i) {
        const int e = t + 256 * i, row = e / BK, k = e % BK;
        const int64_t gm = m0 + row;
        xe[i] = (gm < p.M && k0 + k < p.D1) ? X[gm * p.ldx + k0 + k] : from_f32<T>(0.f);
      }
    }
  };
  auto store_x = [&]() {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int v = t + 256 * i, row = v >> 3, c = v & 7;
        if constexpr (F32) {
          float* dst = (float*)Xs + row * (BK + 1) + c * 4;
          const float* s = (const float*)&xv[i];
          dst[0] = s[0], dst[1] = s[1], dst[2] = s[2], dst[3] = s[3];
        } else {
          *(u32x4*)(Xs + bf16_img_off<BK>(row, c)) = xv[i];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < XE; ++i) {
        const int e = t + 256 * i, row = e / BK, k = e % BK;
        if constexpr (F32)
          ((float*)Xs)[row * (BK + 1) + k] = xe[i];
        else
          *(bf16_t*)(Xs + bf16_img_off<BK>(row, k >> 3) + (k & 7) * 2) = xe[i];
      }
    }
  };
  // F1 tile: logical [k in 0..BK) x r in 0..64)
  auto load_f1 = [&](int k0) {
    if constexpr (FASTF) {
      if constexpr (!BWD) {
        // stored [k][r]: thread = (r pair, k octet); 8 dwords down the k axis
        const int rp = t & 31, ko = t >> 5;
#pragma unroll
        for (int j = 0; j < 8; ++j) fd[j] = f1_fetch2<false>(p, k0 + ko * 8 + j, 2 * rp);
      } else {
        // stored [r][k]: thread = (r, k quarter); 8 dwords along k
        const int r = t >> 2, kq = t & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) fd[j] = f1_fetch2<true>(p, k0 + kq * 16 + 2 * j, r);
      }
    } else {
#pragma unroll
      for (int i = 0; i < FE; ++i) {
        const int e = t + 256 * i;
        const int r = BWD ? e / BK : e & 63;
        const int k = BWD ? e % BK : e >> 6;
        fe[i] = f1_fetch<T, BWD>(p, k0 + k, r);
      }
    }
  };
  auto store_f1 = [&]() {
    if constexpr (FASTF) {
      if constexpr (!BWD) {
        const int rp = t & 31, ko = t >> 5;
        u32x4 c0, c1;
        transpose_8x2(fd, c0, c1);
        *(u32x4*)(F1s + bf16_img_off<BK>(2 * rp, ko)) = c0;
        *(u32x4*)(F1s + bf16_img_off<BK>(2 * rp + 1, ko)) = c1;
      } else {
        const int r = t >> 2, kq = t & 3;
        *(u32x4*)(F1s + bf16_img_off<BK>(r, 2 * kq)) = u32x4{fd[0], fd[1], fd[2], fd[3]};
        *(u32x4*)(F1s + bf16_img_off<BK>(r, 2 * kq + 1)) = u32x4{fd[4], fd[5], fd[6], fd[7]};
      }
    } else {
#pragma unroll
      for (int i = 0; i < FE; ++i) {
        const int e = t + 256 * i;
        const int r = BWD ? e / BK : e & 63;
        const int k = BWD ? e % BK : e >> 6;
        if constexpr (F32) {
          if constexpr (BWD)
            ((float*)F1s)[r * (BK + 1) + k] = fe[i];
          else
            ((float*)F1s)[k * CH_RP + r] = fe[i];
        } else {
          *(bf16_t*)(F1s + bf16_img_off<BK>(r, k >> 3) + (k & 7) * 2) = fe[i];
        }
      }
    }
  };

  // ------------------------------------------------------------------ phase 1: H = X . F1
  f32x16 hacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) hacc[i] = 0.f;

  const int nk = (p.D1 + BK - 1) / BK;
  load_x(0);
  load_f1(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_x();
    store_f1();
    __syncthreads();
    if (kt + 1 < nk) {
      load_x((kt + 1) * BK);
      load_f1((kt + 1) * BK);
    }
    if constexpr (F32) {
      const float* xs = (const float*)Xs + (wm * 32 + li) * (BK + 1) + lh;
      const float* fs = BWD ? (const float*)F1s + (wn * 32 + li) * (BK + 1) + lh
                            : (const float*)F1s + lh * CH_RP + wn * 32 + li;
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        const float a = xs[2 * ks];
        const float b = BWD ? fs[2 * ks] : fs[2 * ks * CH_RP];
        hacc = mfma32(a, b, hacc);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const bf16x8 a = *(const bf16x8*)(Xs + bf16_img_off<BK>(wm * 32 + li, 2 * ks + lh));
        const bf16x8 b = *(const bf16x8*)(F1s + bf16_img_off<BK>(wn * 32 + li, 2 * ks + lh));
        hacc = mfma32(a, b, hacc);
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ hand-off: scale, save, Hs
  {
    const int c = wn * 32 + li;  // H column (rank index) held by this lane
    const bool live = c < rtot;
    const float cs = p.scale;
    T* Hsave = (T*)p.Hsave;
    const int hc = c;  // Hsave is [M, 64]: live columns then zeros
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = wm * 32 + acc_row(reg, lane);
      const float hu = hacc[reg];
      const float hsc = hu * cs;
      if constexpr (F32)
        ((float*)Hs)[row * (CH_RP + 1) + c] = hsc;
      else
        *(bf16_t*)(Hs + bf16_img_off<CH_RP>(row, c >> 3) + (c & 7) * 2) = (bf16_t)hsc;
      if (Hsave && m0 + row < p.M) {
        // live columns, zero padding, and 1.0 in column 63 (when free): the skinny-TN kernel turns that
        // column into the column sums of its other operand (dbias) at no cost
        float sv = live ? hsc : 0.f;  // saved SCALED (fwd: s*x.A, so dB = hsave^T dY needs no further scale)
        if (c == CH_RP - 1 && rtot < CH_RP) sv = 1.f;
        Hsave[(m0 + row) * CH_RP + hc] = from_f32<T>(sv);
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase 2: Y = H . F2
  constexpr int F2E = CH_RP * BN / 256;  // elements per thread (32 bf16 / 16 f32)
  T f2e[FASTF ? 1 : F2E];
  uint32_t f2d[FASTF ? 2 : 1][8];
  auto load_f2 = [&](int n0) {
    if constexpr (FASTF) {
      if constexpr (!BWD) {
        // stored [r][n]: items (n pair 0..63, k octet 0..7), two per thread; 8 dwords down r
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int item = t + 256 * it, np = item & 63, ko = item >> 6;
#pragma unroll
          for (int j = 0; j < 8; ++j) f2d[it][j] = f2_fetch2<false>(p, ko * 8 + j, n0 + 2 * np);
        }
      } else {
        // stored [n][r]: thread = (n 0..127, half); 16 dwords along r
        const int n = t >> 1, half = t & 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) f2d[j >> 3][j & 7] = f2_fetch2<true>(p, half * 32 + 2 * j, n0 + n);
      }
    } else {
#pragma unroll
      for (int i = 0; i < F2E; ++i) {
        const int e = t + 256 * i;
        const int n = BWD ? e >> 6 : e % BN;
        const int r = BWD ? e & 63 : e / BN;
        f2e[i] = f2_fetch<T, BWD>(p, r, n0 + n);
      }
    }
  };
  auto store_f2 = [&]() {
    if constexpr (FASTF) {
      if constexpr (!BWD) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int item = t + 256 * it, np = item & 63, ko = item >> 6;
          u32x4 c0, c1;
          transpose_8x2(f2d[it], c0, c1);
          *(u32x4*)(F2s + bf16_img_off<CH_RP>(2 * np, ko)) = c0;
          *(u32x4*)(F2s + bf16_img_off<CH_RP>(2 * np + 1, ko)) = c1;
        }
      } else {
        const int n = t >> 1, half = t & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(u32x4*)(F2s + bf16_img_off<CH_RP>(n, half * 4 + q)) =
              u32x4{f2d[q >> 1][4 * (q & 1)], f2d[q >> 1][4 * (q & 1) + 1], f2d[q >> 1][4 * (q & 1) + 2],
                    f2d[q >> 1][4 * (q & 1) + 3]};
      }
    } else {
#pragma unroll
      for (int i = 0; i < F2E; ++i) {
        const int e = t + 256 * i;
        const int n = BWD ? e >> 6 : e % BN;
        const int r = BWD ? e & 63 : e / BN;
        if constexpr (F32) {
          if constexpr (BWD)
            ((float*)F2s)[n * (CH_RP + 1) + r] = f2e[i];
          else
            ((float*)F2s)[r * BN + n] = f2e[i];
        } else {
          *(bf16_t*)(F2s + bf16_img_off<CH_RP>(n, r >> 3) + (r & 7) * 2) = f2e[i];
        }
      }
    }
  };

  float* scratch = (float*)(smem + L::OFF_SCR + w * L::SCR);
  T* Y = (T*)p.Y;
  const T* bias = (const T*)p.bias;
  const int nn = (p.D2 + BN - 1) / BN;
  const int ksteps = F32 ? (rtot + 1) / 2 : (rtot + 15) / 16;
  load_f2(0);
  for (int nc = 0; nc < nn; ++nc) {
    store_f2();
    __syncthreads();
    if (nc + 1 < nn) load_f2((nc + 1) * BN);
    f32x16 yacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) yacc[nt][i] = 0.f;
    const int ncol = wn * (BN / 2);  // this wave's first column inside the chunk
    if constexpr (F32) {
      const float* hs = (const float*)Hs + (wm * 32 + li) * (CH_RP + 1) + lh;
      const float* fs = BWD ? (const float*)F2s + (ncol + li) * (CH_RP + 1) + lh
                            : (const float*)F2s + lh * BN + ncol + li;
      for (int ks = 0; ks < ksteps; ++ks) {
        const float a = hs[2 * ks];
        const float b = BWD ? fs[2 * ks] : fs[2 * ks * BN];
        yacc[0] = mfma32(a, b, yacc[0]);
      }
    } else {
      for (int ks = 0; ks < ksteps; ++ks) {
        const bf16x8 a = *(const bf16x8*)(Hs + bf16_img_off<CH_RP>(wm * 32 + li, 2 * ks + lh));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 b = *(const bf16x8*)(F2s + bf16_img_off<CH_RP>(ncol + nt * 32 + li, 2 * ks + lh));
          yacc[nt] = mfma32(a, b, yacc[nt]);
        }
      }
    }
    wave_store_tiles<T, NT, VEC>(yacc, scratch, Y, p.ldy, m0 + wm * 32, nc * BN + ncol, p.M, p.D2, 1.f, p.beta,
                                 bias, lane);
    __syncthreads();
  }
}

// =================================================================================================
template <typename T, bool BWD, bool VEC, bool FASTF> static int launch_chain_k(const ChainParams& p, hipStream_t stream) {
  using L = ChainLds<T, BWD>;
  const int grid = ceil_div(p.M, CH_BM);
  if (grid <= 0) return SOW_OK;
  auto k = chain_kernel<T, BWD, VEC, FASTF>;
  SOW_SET_MAX_LDS_ONCE(L::BYTES, chain_kernel<T, BWD, VEC, FASTF>);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), L::BYTES, stream, p);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}
template <typename T, bool BWD> static int launch_chain_t(const ChainParams& p, bool vec, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    if (p.fast_factors)
      return vec ? launch_chain_k<T, BWD, true, true>(p, stream) : launch_chain_k<T, BWD, false, true>(p, stream);
  }
  return vec ? launch_chain_k<T, BWD, true, false>(p, stream) : launch_chain_k<T, BWD, false, false>(p, stream);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

// Host entry used by api.hip.  dtype: SOW_F32 / SOW_BF16.
int launch_chain(ChainParams p, int dtype, bool bwd, hipStream_t stream) {
  if (!p.X || !p.Y) return SOW_ERR_NULL;
  if (p.ra != 0 || p.rb <= 0 || p.rb > CH_RP) return SOW_ERR_SHAPE;
  if (!p.F1b || !p.F2b) return SOW_ERR_NULL;
  if (chain2_supported(p, dtype) && !sw_on(SW_FORCE_CHAIN_V1)) {
    const int rc = launch_chain2(p, bwd, stream);
    if (rc != SOW_ERR_ALIGN) return rc;  // factor alignment not met: fall through to the generic kernel
  }
  if (chain3f_supported(p, dtype)) {
    const int rc = launch_chain3f(p, bwd, stream);
    if (rc != SOW_ERR_ALIGN) return rc;
  }
  if (chain2f_supported(p, dtype) && !sw_on(SW_FORCE_CHAIN_V1)) {
    const int rc = launch_chain2f(p, bwd, stream);
    if (rc != SOW_ERR_ALIGN) return rc;
  }
  const int ve = dtype == SOW_F32 ? 4 : 8;
  const bool vec = p.D1 % ve == 0 && p.D2 % ve == 0 && p.ldx % ve == 0 && p.ldy % ve == 0 && aligned16(p.X) &&
                   aligned16(p.Y) && (!p.bias || aligned16(p.bias));
  if (dtype == SOW_BF16) {
    // dword factor loaders: pairs of bf16 must not straddle a segment or a row
    bool f = p.rb % 2 == 0 && p.ldf1b % 2 == 0 && p.ldf2b % 2 == 0 && aligned4(p.F1b) && aligned4(p.F2b);
    // the dword reads pair elements along the STORAGE-contiguous axis: fwd F1 pairs ranks, F2 pairs n;
    // bwd F1 pairs k (= D1), F2 pairs ranks.  Even extents keep the pairs inside the matrix.
    f = f && p.D1 % 2 == 0 && p.D2 % 2 == 0;
    p.fast_factors = f ? 1 : 0;
    return bwd ? launch_chain_t<bf16_t, true>(p, vec, stream) : launch_chain_t<bf16_t, false>(p, vec, stream);
  } else if (dtype == SOW_F32) {
    p.fast_factors = 0;
    return bwd ? launch_chain_t<float, true>(p, vec, stream) : launch_chain_t<float, false>(p, vec, stream);
  }
  return SOW_ERR_DTYPE;
}

}  // namespace sow
