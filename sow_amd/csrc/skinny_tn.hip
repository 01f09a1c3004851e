// Tall-skinny transposed products  G[D, 64] = sum_t M[t, D]^T . S[t, 64]   (K = tokens)
//
// These are the weight gradients of SoWLinear (autograd of tn_gradient/layer/sow.py:117):
//   dA   = x^T  . dh        M = x  [T, d_in],  S = dh = s*dY.B^T  [T, 64] (saved by the chain kernel)
//   dB^T = dY^T . h         M = dY [T, d_out], S = h  = x.A       [T, 64]
//   dbias = colsum(dY)      obtained for free as an extra all-ones column of S (column 63).
// The reduction runs over T (tens of thousands) into a tiny [D, r] output, so the grid is
// (column groups of 64) x (token slabs); every workgroup keeps its [64, 64] fp32 partial in MFMA
// accumulators over its whole slab and writes it once to the workspace; tn_reduce sums the slabs in a
// fixed order (deterministic, no float atomics), applies alpha/beta, crops 64 -> r and writes
// [D, r] or its transpose [r, D].
// Column groups of one slab are NS blocks apart (NS % 8 == 0) so they share an XCD and S is served
// from that XCD's L2 after the first read.
#include "kernels.hpp"

namespace sow {


constexpr int TN_BD = 64;

template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> {
  static constexpr int BT = 64;  // tokens per staged chunk
};
template <> struct TnCfg<float> {
  static constexpr int BT = 32;
};

template <typename T> __global__ __launch_bounds__(256, 2) void tn_partial_kernel(const TnParams p) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int BT = TnCfg<T>::BT;
  // images: bf16 -> Mi[d][t], Si[r][t] (k = t contiguous, swizzled); f32 -> Mi[t][d], Si[t][r]
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 64 * (F32 ? 2 : 2)];
  char* Mi = smem;
  char* Si = smem + 8192;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int dm = w >> 1, rn = w & 1, li = lane & 31, lh = lane >> 5;

  // block -> (job, column group, slab); same-slab blocks are ns apart
  int b = blockIdx.x;
  int jid = 0;
  if (p.njobs > 1 && b >= p.job[0].ncg * p.ns) {
    b -= p.job[0].ncg * p.ns;
    jid = 1;
  }
  const TnJob& J = p.job[jid];
  const int cg = b / p.ns, slab = b % p.ns;
  const int d0 = cg * TN_BD;
  const int64_t t_begin = (int64_t)slab * p.slab_len;
  int64_t t_end = t_begin + p.slab_len;
  if (t_end > p.T) t_end = p.T;
  const T* Mg = (const T*)J.M;
  const T* Sg = (const T*)J.S;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // staging registers
  uint32_t md[8], sd[8];       // bf16 dword path
  u32x4 mv[2], sv[2];          // f32 vector path
  T me[16];                    // generic M path (64*BT/256 elements: 16 bf16 / 8 f32)

  auto load = [&](int64_t tt0) {
    if constexpr (!F32) {
      const int dp = t & 31, to = t >> 5;  // (column pair, token octet)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t tt = tt0 + to * 8 + j;
        const bool tv = tt < t_end;
        sd[j] = tv ? *(const uint32_t*)(Sg + tt * 64 + 2 * dp) : 0u;
        if (J.ones_col >= 0 && (J.ones_col >> 1) == dp && tv) {
          // replace element ones_col of the pair by bf16(1.0) = 0x3F80
          sd[j] = (J.ones_col & 1) ? ((sd[j] & 0xffffu) | 0x3F800000u) : ((sd[j] & 0xffff0000u) | 0x3F80u);
        }
      }
      if (J.vec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t tt = tt0 + to * 8 + j;
          const int d = d0 + 2 * dp;
          md[j] = (tt < t_end && d < J.D) ? *(const uint32_t*)(Mg + tt * J.ldm + d) : 0u;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          const int64_t tt = tt0 + tl;
          me[i] = (tt < t_end && d0 + d < J.D) ? Mg[tt * J.ldm + d0 + d] : (bf16_t)0.f;
        }
      }
    } else {
      // f32: chunk [32 t][64 cols]; 512 vectors of 16 B, two per thread
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = t + 256 * i, tl = v >> 4, c = v & 15;
        const int64_t tt = tt0 + tl;
        sv[i] = tt < t_end ? *(const u32x4*)(Sg + tt * 64 + c * 4) : u32x4{0, 0, 0, 0};
        if (J.ones_col >= 0 && (J.ones_col >> 2) == c && tt < t_end)
          sv[i][J.ones_col & 3] = __builtin_bit_cast(uint32_t, 1.0f);
      }
      if (J.vec) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int v = t + 256 * i, tl = v >> 4, c = v & 15;
          const int64_t tt = tt0 + tl;
          const int d = d0 + c * 4;
          mv[i] = (tt < t_end && d < J.D) ? *(const u32x4*)(Mg + tt * J.ldm + d) : u32x4{0, 0, 0, 0};
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          const int64_t tt = tt0 + tl;
          me[i] = (tt < t_end && d0 + d < J.D) ? Mg[tt * J.ldm + d0 + d] : 0.f;
        }
      }
    }
  };
  auto store = [&]() {
    if constexpr (!F32) {
      const int dp = t & 31, to = t >> 5;
      u32x4 c0, c1;
      transpose_8x2(sd, c0, c1);
      *(u32x4*)(Si + bf16_img_off<BT>(2 * dp, to)) = c0;
      *(u32x4*)(Si + bf16_img_off<BT>(2 * dp + 1, to)) = c1;
      if (J.vec) {
        transpose_8x2(md, c0, c1);
        *(u32x4*)(Mi + bf16_img_off<BT>(2 * dp, to)) = c0;
        *(u32x4*)(Mi + bf16_img_off<BT>(2 * dp + 1, to)) = c1;
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          *(bf16_t*)(Mi + bf16_img_off<BT>(d, tl >> 3) + (tl & 7) * 2) = me[i];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = t + 256 * i, tl = v >> 4, c = v & 15;
        *(u32x4*)(Si + (tl * 64 + c * 4) * 4) = sv[i];
      }
      if (J.vec) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int v = t + 256 * i, tl = v >> 4, c = v & 15;
          *(u32x4*)(Mi + (tl * 64 + c * 4) * 4) = mv[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          ((float*)Mi)[tl * 64 + d] = me[i];
        }
      }
    }
  };

  if (t_begin < t_end) load(t_begin);
  for (int64_t tt0 = t_begin; tt0 < t_end; tt0 += BT) {
    store();
    __syncthreads();
    if (tt0 + BT < t_end) load(tt0 + BT);
    if constexpr (F32) {
      const float* ms = (const float*)Mi + lh * 64 + dm * 32 + li;
      const float* ss = (const float*)Si + lh * 64 + rn * 32 + li;
#pragma unroll
      for (int ks = 0; ks < BT / 2; ++ks) acc = mfma32(ms[2 * ks * 64], ss[2 * ks * 64], acc);
    } else {
#pragma unroll
      for (int ks = 0; ks < BT / 16; ++ks) {
        const bf16x8 a = *(const bf16x8*)(Mi + bf16_img_off<BT>(dm * 32 + li, 2 * ks + lh));
        const bf16x8 bb = *(const bf16x8*)(Si + bf16_img_off<BT>(rn * 32 + li, 2 * ks + lh));
        acc = mfma32(a, bb, acc);
      }
    }
    __syncthreads();
  }

  // partial[slab][d0 + row][col]  (Dpad = ncg * 64 rows)
  float* P = J.partial + ((int64_t)slab * J.ncg * TN_BD + d0) * 64;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) P[(dm * 32 + acc_row(reg, lane)) * 64 + rn * 32 + li] = acc[reg];
}

// ---------------------------------------------------------------------------------------------
// Deterministic slab reduction + crop + optional transpose + cast.
//   out[d][r]   (transpose = 0, ld = out_ld)   or   out[r][d]   (transpose = 1)
//   colsum[d] = sum_s partial[s][d][ones_col]
// ---------------------------------------------------------------------------------------------

template <typename T> __global__ __launch_bounds__(256) void tn_reduce_kernel(const ReduceParams p) {
  int b = blockIdx.x, jid = 0;
  if (p.njobs > 1 && b >= p.blocks0) {
    b -= p.blocks0;
    jid = 1;
  }
  const ReduceJob& J = p.job[jid];
  // one thread per (d, 4 consecutive columns): 16 threads per row, 16 rows per block
  const int d = b * 16 + (threadIdx.x >> 4);
  const int c4 = (threadIdx.x & 15) * 4;
  if (d >= J.D) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const int64_t stride = (int64_t)J.Dpad * 64;
  const float* src = J.partial + (int64_t)d * 64 + c4;
  for (int i = 0; i < p.ns; ++i) {
    const f32x4 v = *(const f32x4*)(src + i * stride);
    s[0] += v[0], s[1] += v[1], s[2] += v[2], s[3] += v[3];
  }
  T* out = (T*)J.out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c4 + j;
    if (c < J.r) {
      T* dst = J.transpose ? out + (int64_t)c * J.out_ld + d : out + (int64_t)d * J.out_ld + c;
      float v = J.alpha * s[j];
      if (J.beta != 0.f) v += J.beta * to_f32(*dst);
      *dst = from_f32<T>(v);
    }
    if (J.colsum && c == J.ones_col) {
      T* cs = (T*)J.colsum + d;
      float v = s[j];
      if (J.beta != 0.f) v += J.beta * to_f32(*cs);
      *cs = from_f32<T>(v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
int tn_pick_slabs(int64_t T, int total_colgroups, int dtype, int* slab_len) {
  const int bt = dtype == SOW_F32 ? TnCfg<float>::BT : TnCfg<bf16_t>::BT;
  // aim for ~2 workgroups per CU (512) but keep slabs >= 512 tokens so the partial traffic
  // (ns * D * 64 * 4 bytes) stays a small fraction of the streamed operand
  int ns = (512 + total_colgroups - 1) / total_colgroups;
  int64_t max_ns = (T + 511) / 512;
  if (max_ns < 1) max_ns = 1;
  if (ns > max_ns) ns = (int)max_ns;
  if (ns < 1) ns = 1;
  if (ns > 8) ns = (ns + 7) & ~7;  // multiple of 8: same-slab blocks share an XCD
  int64_t len = (T + ns - 1) / ns;
  len = (len + bt - 1) / bt * bt;
  if (len < bt) len = bt;
  ns = (int)((T + len - 1) / len);
  if (ns < 1) ns = 1;
  *slab_len = (int)len;
  return ns;
}

size_t tn_partial_bytes(int ns, int D) { return (size_t)ns * ((D + 63) / 64 * 64) * 64 * sizeof(float); }

int launch_tn(const TnParams& p, int dtype, hipStream_t stream) {
  int blocks = 0;
  for (int j = 0; j < p.njobs; ++j) blocks += p.job[j].ncg * p.ns;
  if (blocks == 0) return SOW_OK;
  if (dtype == SOW_BF16)
    hipLaunchKernelGGL(tn_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, p);
  else if (dtype == SOW_F32)
    hipLaunchKernelGGL(tn_partial_kernel<float>, dim3(blocks), dim3(256), 0, stream, p);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

int launch_tn_reduce(ReduceParams p, int dtype, hipStream_t stream) {
  p.blocks0 = (p.job[0].D + 15) / 16;
  int blocks = p.blocks0;
  if (p.njobs > 1) blocks += (p.job[1].D + 15) / 16;
  if (blocks == 0) return SOW_OK;
  if (dtype == SOW_BF16)
    hipLaunchKernelGGL(tn_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, p);
  else if (dtype == SOW_F32)
    hipLaunchKernelGGL(tn_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, p);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
