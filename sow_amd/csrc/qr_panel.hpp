// Householder panel + org2r for ONE matrix, executed by one 1024-thread workgroup (shared by qr.hip and accumulate.hip).
#pragma once
#include "common.hpp"

namespace sow {

constexpr int QR_THREADS = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// W[:, :kc] (row-major, Tin) -> Pt column-major fp32 [kc][m]
template <typename Tin>
__global__ void qr_copy_in_kernel(const Tin* W, int64_t ldw, float* Pt, int m, int kc) {
  const int64_t n = (int64_t)m * kc;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / kc), c = (int)(idx % kc);  // consecutive threads read consecutive columns
    Pt[(int64_t)c * m + i] = to_f32(W[(int64_t)i * ldw + c]);
  }
}

// qsm: dynamic LDS of qr_panel_lds_bytes(m, kc) bytes
__device__ __forceinline__ void qr_panel_body(float* Pt, float* Qt, int m, int kc, int r, float* qsm) {
  float* vs = qsm;            // [m] active reflector
  float* taus = qsm + m;      // [kc]
  float* red = taus + kc;     // [16] cross-wave reduction
  __shared__ float sh_tau;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int j = 0; j < kc; ++j) {
    float* col = Pt + (int64_t)j * m;
    float s = 0.f;
    for (int i = j + 1 + tid; i < m; i += QR_THREADS) {
      const float a = col[i];
      s += a * a;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
      float x2 = 0.f;
      for (int q = 0; q < QR_THREADS / 64; ++q) x2 += red[q];
      const float alpha = col[j];
      float tau = 0.f, scale = 0.f;
      if (x2 != 0.f) {
        const float nrm = sqrtf(alpha * alpha + x2);
        const float beta = alpha >= 0.f ? -nrm : nrm;  // -sign(alpha) * nrm, sign(0) = +
        tau = (beta - alpha) / beta;
        scale = 1.f / (alpha - beta);
        col[j] = beta;
      }
      taus[j] = tau;
      sh_tau = tau;
      red[0] = scale;
    }
    __syncthreads();
    const float tau = sh_tau, scale = red[0];
    for (int i = j + tid; i < m; i += QR_THREADS) {
      if (i == j) {
        vs[i] = 1.f;
      } else {
        const float v = col[i] * scale;
        col[i] = v;
        vs[i] = v;
      }
    }
    __syncthreads();
    if (tau != 0.f) {
      for (int c = j + 1 + wave; c < kc; c += QR_THREADS / 64) {
        float* cc = Pt + (int64_t)c * m;
        float d = 0.f;
        for (int i = j + lane; i < m; i += 64) d += vs[i] * cc[i];
        d = wave_sum(d) * tau;
        for (int i = j + lane; i < m; i += 64) cc[i] -= d * vs[i];
      }
    }
    __syncthreads();
  }

  // ---- form Q[:, :r] = H_0 H_1 ... H_{kc-1} I[:, :r]  (column-major Qt[r][m])
  for (int64_t idx = tid; idx < (int64_t)r * m; idx += QR_THREADS) {
    const int c = (int)(idx / m), i = (int)(idx % m);
    Qt[idx] = (i == c) ? 1.f : 0.f;
  }
  __syncthreads();
  for (int j = kc - 1; j >= 0; --j) {
    const float* col = Pt + (int64_t)j * m;
    for (int i = j + tid; i < m; i += QR_THREADS) vs[i] = (i == j) ? 1.f : col[i];
    __syncthreads();
    const float tau = taus[j];
    if (tau != 0.f) {
      for (int c = j + wave; c < r; c += QR_THREADS / 64) {
        float* qc = Qt + (int64_t)c * m;
        float d = 0.f;
        for (int i = j + lane; i < m; i += 64) d += vs[i] * qc[i];
        d = wave_sum(d) * tau;
        for (int i = j + lane; i < m; i += 64) qc[i] -= d * vs[i];
      }
    }
    __syncthreads();
  }
}

}  // namespace sow
