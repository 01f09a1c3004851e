// Small HBM-bound helper kernels around the SoW hot path.
//   * multi-tensor zero fill       -> reset_optimizer (scripts/utils/training_utils.py:257-277) and B <- 0
//                                     in SoWLinear.accumulate (sow.py:159)
//   * flat AdamW step              -> the factor parameter group of simple_train.py:502-506 as ONE launch
//   * dense Adam moment update     -> TTAdam.step's dense section (ttadam.py:89-111)
//   * TT Hadamard core product     -> TensorTrain.__mul__ (tt.py:469-475)
//   * axpby / scale                -> TTSGD p += -lr * d_p (ttsgd.py:78)
#include "kernels.hpp"

namespace sow {

constexpr int MT_MAX = 48;  // tensors per multi-tensor launch (kernel-argument resident table)

struct MultiTensor {
  void* ptr[MT_MAX];
  int64_t bytes[MT_MAX];
  int n;
};

// each workgroup walks its tensor with 16-byte stores; grid.y = tensor, grid.x = chunks
__global__ __launch_bounds__(256) void multi_zero_kernel(const MultiTensor mt) {
  const int ti = blockIdx.y;
  if (ti >= mt.n) return;
  char* base = (char*)mt.ptr[ti];
  const int64_t nb = mt.bytes[ti];
  const uintptr_t addr = reinterpret_cast<uintptr_t>(base);
  const int64_t head = ((16 - (addr & 15)) & 15) < nb ? ((16 - (addr & 15)) & 15) : nb;
  const int64_t nvec = (nb - head) / 16;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  u32x4* v = (u32x4*)(base + head);
  for (int64_t i = tid; i < nvec; i += nth) v[i] = u32x4{0, 0, 0, 0};
  const int64_t tail0 = head + nvec * 16;
  for (int64_t i = tid; i < head; i += nth) base[i] = 0;
  for (int64_t i = tail0 + tid; i < nb; i += nth) base[i] = 0;
}

int launch_multi_zero(void* const* ptrs, const int64_t* bytes, int n, hipStream_t stream) {
  for (int off = 0; off < n; off += MT_MAX) {
    MultiTensor mt;
    mt.n = n - off < MT_MAX ? n - off : MT_MAX;
    int64_t maxb = 0;
    for (int i = 0; i < mt.n; ++i) {
      mt.ptr[i] = ptrs[off + i];
      mt.bytes[i] = bytes[off + i];
      if (bytes[off + i] > maxb) maxb = bytes[off + i];
      if (bytes[off + i] < 0 || (bytes[off + i] > 0 && !ptrs[off + i])) return SOW_ERR_NULL;
    }
    if (maxb == 0) continue;
    int gx = (int)((maxb / 16 + 255) / 256);
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(multi_zero_kernel, dim3(gx, mt.n), dim3(256), 0, stream, mt);
    SOW_CHECK_LAUNCH();
  }
  return SOW_OK;
}

// ---------------------------------------------------------------------------------------------
// AdamW over one flat buffer (torch.optim.AdamW semantics, no amsgrad, maximize=False):
//   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
//   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// State m, v are fp32 or the parameter dtype (TS); math in fp32.
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
__global__ __launch_bounds__(256) void adamw_flat_kernel(T* p, const T* g, TS* m, TS* v, int64_t n, float lr, float b1,
                                                         float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                         float grad_scale) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < n; i += nth) {
    float pv = to_f32(p[i]);
    const float gv = to_f32(g[i]) * grad_scale;
    float mv = to_f32(m[i]), vv = to_f32(v[i]);
    pv *= 1.f - lr * wd;
    mv = b1 * mv + (1.f - b1) * gv;
    vv = b2 * vv + (1.f - b2) * gv * gv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[i] = from_f32<T>(pv);
    m[i] = from_f32<TS>(mv);
    v[i] = from_f32<TS>(vv);
  }
}

int launch_adamw_flat(void* p, const void* g, void* m, void* v, int64_t n, float lr, float b1, float b2, float eps,
                      float wd, int step, float grad_scale, int dtype, int state_dtype, hipStream_t stream) {
  if (n <= 0) return SOW_OK;
  if (!p || !g || !m || !v) return SOW_ERR_NULL;
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2s = sqrtf(1.f - powf(b2, (float)step));
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
#define SOW_ADAMW(T, TS) \
  hipLaunchKernelGGL((adamw_flat_kernel<T, TS>), dim3(grid), dim3(256), 0, stream, (T*)p, (const T*)g, (TS*)m, (TS*)v, n, lr, b1, b2, eps, wd, bc1, bc2s, grad_scale)
  if (dtype == SOW_F32 && state_dtype == SOW_F32) SOW_ADAMW(float, float);
  else if (dtype == SOW_BF16 && state_dtype == SOW_BF16) SOW_ADAMW(bf16_t, bf16_t);
  else if (dtype == SOW_BF16 && state_dtype == SOW_F32) SOW_ADAMW(bf16_t, float);
  else return SOW_ERR_DTYPE;
#undef SOW_ADAMW
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

// ---------------------------------------------------------------------------------------------
// TTAdam dense section (ttadam.py:84-111), fp32:  v<0 -> 0 clamp (only when clamp_v), then
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p += -step_size * m/(sqrt(v)+eps) ; p += -lr*wd*p
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ttadam_dense_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                           float b1, float b2, float eps, float step_size,
                                                           float lr_wd, int clamp_v) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < n; i += nth) {
    const float gv = g[i];
    float vv = v[i];
    if (clamp_v && vv < 0.f) vv = 0.f;
    const float mv = m[i] * b1 + gv * (1.f - b1);
    vv = vv * b2 + gv * gv * (1.f - b2);
    float pv = p[i] + (mv / (sqrtf(vv) + eps)) * (-step_size);
    if (lr_wd > 0.f) pv = pv + pv * (-lr_wd);
    p[i] = pv;
    m[i] = mv;
    v[i] = vv;
  }
}

int launch_ttadam_dense(float* p, const float* g, float* m, float* v, int64_t n, float b1, float b2, float eps,
                        float step_size, float lr_wd, int clamp_v, hipStream_t stream) {
  if (n <= 0) return SOW_OK;
  if (!p || !g || !m || !v) return SOW_ERR_NULL;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(ttadam_dense_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v, n, b1, b2, eps, step_size, lr_wd, clamp_v);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

// ---------------------------------------------------------------------------------------------
// TT Hadamard core product: out[(a,c), i, j, (b,d)] = A[a,i,j,b] * B[c,i,j,d]   (tt.py:469-475)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tt_kron_core_kernel(const float* A, const float* B, float* out, int ra0, int rb0,
                                                           int ij, int ra1, int rb1) {
  const int64_t n = (int64_t)ra0 * rb0 * ij * ra1 * rb1;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = tid; idx < n; idx += nth) {
    int64_t q = idx;
    const int d = (int)(q % rb1);
    q /= rb1;
    const int b = (int)(q % ra1);
    q /= ra1;
    const int e = (int)(q % ij);
    q /= ij;
    const int c = (int)(q % rb0);
    const int a = (int)(q / rb0);
    out[idx] = A[((int64_t)a * ij + e) * ra1 + b] * B[((int64_t)c * ij + e) * rb1 + d];
  }
}

int launch_tt_kron_core(const float* A, const float* B, float* out, int ra0, int rb0, int ij, int ra1, int rb1,
                        hipStream_t stream) {
  const int64_t n = (int64_t)ra0 * rb0 * ij * ra1 * rb1;
  if (n <= 0) return SOW_OK;
  if (!A || !B || !out) return SOW_ERR_NULL;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(tt_kron_core_kernel, dim3(grid), dim3(256), 0, stream, A, B, out, ra0, rb0, ij, ra1, rb1);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

// ---------------------------------------------------------------------------------------------
// max |x| over a contiguous fp32 buffer (TensorTrain.sqrt / sqrtinv scaling, tt.py:288, 322): one
// workgroup, result written to out[0]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void absmax_kernel(const float* x, int64_t n, float* out) {
  __shared__ float red[16];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 16; ++q) m = fmaxf(m, red[q]);
    out[0] = m;
  }
}

int launch_absmax(const float* x, int64_t n, float* out, hipStream_t stream) {
  if (!x || !out) return SOW_ERR_NULL;
  if (n < 0) return SOW_ERR_SHAPE;
  hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(1024), 0, stream, x, n, out);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

// ---------------------------------------------------------------------------------------------
// Batched inverse of small [r, r] matrices by Gauss-Jordan with partial pivoting, one thread per matrix
// (TensorTrain.reciprocal, tt.py:480-494: the middle cores' [:, i, j, :] slices).  In [batch, r, r] fp32.
// ---------------------------------------------------------------------------------------------
constexpr int INV_MAX_R = 16;
__global__ __launch_bounds__(64) void small_inverse_kernel(const float* A, float* out, int batch, int r) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  float a[INV_MAX_R][INV_MAX_R], inv[INV_MAX_R][INV_MAX_R];
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < r; ++j) {
      a[i][j] = A[((int64_t)b * r + i) * r + j];
      inv[i][j] = i == j ? 1.f : 0.f;
    }
  for (int c = 0; c < r; ++c) {
    int piv = c;
    float best = fabsf(a[c][c]);
    for (int i = c + 1; i < r; ++i)
      if (fabsf(a[i][c]) > best) best = fabsf(a[i][c]), piv = i;
    if (piv != c)
      for (int j = 0; j < r; ++j) {
        float t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t;
        t = inv[c][j]; inv[c][j] = inv[piv][j]; inv[piv][j] = t;
      }
    const float d = 1.f / a[c][c];
    for (int j = 0; j < r; ++j) a[c][j] *= d, inv[c][j] *= d;
    for (int i = 0; i < r; ++i) {
      if (i == c) continue;
      const float f = a[i][c];
      for (int j = 0; j < r; ++j) a[i][j] -= f * a[c][j], inv[i][j] -= f * inv[c][j];
    }
  }
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < r; ++j) out[((int64_t)b * r + i) * r + j] = inv[i][j];
}

int launch_small_inverse(const float* A, float* out, int batch, int r, hipStream_t stream) {
  if (!A || !out) return SOW_ERR_NULL;
  if (batch < 0 || r < 1 || r > INV_MAX_R) return SOW_ERR_UNSUPPORTED;
  if (batch == 0) return SOW_OK;
  hipLaunchKernelGGL(small_inverse_kernel, dim3((batch + 63) / 64), dim3(64), 0, stream, A, out, batch, r);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

// y = a*x + b*y   (fp32 or bf16)
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* x, T* y, int64_t n, float a, float b) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < n; i += nth) {
    const float yv = b != 0.f ? b * to_f32(y[i]) : 0.f;
    y[i] = from_f32<T>(a * to_f32(x[i]) + yv);
  }
}

int launch_axpby(const void* x, void* y, int64_t n, float a, float b, int dtype, hipStream_t stream) {
  if (n <= 0) return SOW_OK;
  if (!x || !y) return SOW_ERR_NULL;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (dtype == SOW_F32)
    hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (float*)y, n, a, b);
  else if (dtype == SOW_BF16)
    hipLaunchKernelGGL(axpby_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n, a, b);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
