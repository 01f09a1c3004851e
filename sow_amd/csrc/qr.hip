// Householder QR of a column panel, LAPACK conventions (geqr2 + org2r), fp32 internals.
//
// Reference call sites: tn_gradient/utils.py:8-30 (qr_weight: reduced QR, keep Q[:, :k], R[:k, :]),
// tn_gradient/layer/sow.py:96-99 (init), :144-150 (truncated QR of the accumulator), :168-172 (A re-init
// keeps only Q[:, :r] of an [in, out] Gaussian), tn_gradient/tt.py:128-136 (complete-mode QR, truncated).
// Only the first kc = min(k, m, n) columns have to be factored to obtain Q[:, :k]; R[:k, j] for j >= kc
// is Q[:, :k]^T W[:, j], computed by the GEMM kernel from the host wrapper in api.hip.
//
// Sign convention (LAPACK slarfg): beta = -sign(alpha) * ||x||, tau = (beta - alpha) / beta,
// v = x / (alpha - beta), v[0] = 1;  a zero sub-column gives tau = 0 (H = I).  diag(R) therefore has
// mixed signs exactly as torch.linalg.qr on CPU.
//
// One workgroup (1024 threads = 16 waves) per matrix: the panel lives column-major in a global fp32
// workspace (L2 resident), the active reflector is cached in LDS, every wave owns trailing columns
// (dot product + rank-1 update with wave-64 shuffles), two workgroup barriers per column.
#include "kernels.hpp"
#include "qr_panel.hpp"

namespace sow {

__global__ __launch_bounds__(QR_THREADS) void qr_panel_kernel(float* Pt, float* Qt, int m, int kc, int r) {
  extern __shared__ __attribute__((aligned(16))) float qsm[];
  qr_panel_body(Pt, Qt, m, kc, r, qsm);
}

// Qt [r][m] fp32 -> Q [m, r] row-major Tout ; upper triangle of the factored panel -> R[:k, :kc]
template <typename Tout>
__global__ void qr_copy_out_kernel(const float* Qt, const float* Pt, Tout* Q, int64_t ldq, Tout* R, int64_t ldr, int m,
                                   int kc, int r, int k_rows) {
  const int64_t nq = (int64_t)m * r;
  const int64_t nr = R ? (int64_t)k_rows * kc : 0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nq + nr;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < nq) {
      const int i = (int)(idx / r), c = (int)(idx % r);
      Q[(int64_t)i * ldq + c] = from_f32<Tout>(Qt[(int64_t)c * m + i]);
    } else {
      const int64_t e = idx - nq;
      const int i = (int)(e / kc), c = (int)(e % kc);
      const float v = (i <= c && i < m) ? Pt[(int64_t)c * m + i] : 0.f;
      R[(int64_t)i * ldr + c] = from_f32<Tout>(v);
    }
  }
}

// generic 2-D cast copy (strided rows), used for the fp32 staging of bf16 inputs / outputs
template <typename Tin, typename Tout>
__global__ void cast_copy_kernel(const Tin* src, int64_t lds, Tout* dst, int64_t ldd, int64_t rows, int cols) {
  const int64_t n = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / cols;
    const int c = (int)(idx % cols);
    dst[i * ldd + c] = from_f32<Tout>(to_f32(src[i * lds + c]));
  }
}

static int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

int launch_cast_copy(const void* src, int64_t lds, int src_dtype, void* dst, int64_t ldd, int dst_dtype, int64_t rows,
                     int cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return SOW_OK;
  const int g = grid_for(rows * cols);
  if (src_dtype == SOW_F32 && dst_dtype == SOW_F32)
    hipLaunchKernelGGL((cast_copy_kernel<float, float>), dim3(g), dim3(256), 0, stream, (const float*)src, lds, (float*)dst, ldd, rows, cols);
  else if (src_dtype == SOW_F32 && dst_dtype == SOW_BF16)
    hipLaunchKernelGGL((cast_copy_kernel<float, bf16_t>), dim3(g), dim3(256), 0, stream, (const float*)src, lds, (bf16_t*)dst, ldd, rows, cols);
  else if (src_dtype == SOW_BF16 && dst_dtype == SOW_F32)
    hipLaunchKernelGGL((cast_copy_kernel<bf16_t, float>), dim3(g), dim3(256), 0, stream, (const bf16_t*)src, lds, (float*)dst, ldd, rows, cols);
  else if (src_dtype == SOW_BF16 && dst_dtype == SOW_BF16)
    hipLaunchKernelGGL((cast_copy_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, stream, (const bf16_t*)src, lds, (bf16_t*)dst, ldd, rows, cols);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

size_t qr_panel_lds_bytes(int m, int kc) { return ((size_t)m + kc + 16) * sizeof(float); }

// Factor W[:, :kc] and form Q[:, :r].  Pt: [kc*m] floats, Qt: [r*m] floats (workspace).
int launch_qr_panel(const void* W, int64_t ldw, int in_dtype, int m, int kc, int r, float* Pt, float* Qt,
                    hipStream_t stream) {
  if (m <= 0 || kc <= 0 || r <= 0 || kc > m || r > m) return SOW_ERR_SHAPE;
  const size_t lds = qr_panel_lds_bytes(m, kc);
  if (lds > 150 * 1024) return SOW_ERR_UNSUPPORTED;
  const int g = grid_for((int64_t)m * kc);
  if (in_dtype == SOW_F32)
    hipLaunchKernelGGL(qr_copy_in_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)W, ldw, Pt, m, kc);
  else if (in_dtype == SOW_BF16)
    hipLaunchKernelGGL(qr_copy_in_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, (const bf16_t*)W, ldw, Pt, m, kc);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  SOW_SET_MAX_LDS_ONCE(150 * 1024, qr_panel_kernel);
  hipLaunchKernelGGL(qr_panel_kernel, dim3(1), dim3(QR_THREADS), lds, stream, Pt, Qt, m, kc, r);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

int launch_qr_copy_out(const float* Qt, const float* Pt, void* Q, int64_t ldq, void* R, int64_t ldr, int out_dtype, int m,
                       int kc, int r, int k_rows, hipStream_t stream) {
  const int g = grid_for((int64_t)m * r + (R ? (int64_t)k_rows * kc : 0));
  if (out_dtype == SOW_F32)
    hipLaunchKernelGGL(qr_copy_out_kernel<float>, dim3(g), dim3(256), 0, stream, Qt, Pt, (float*)Q, ldq, (float*)R, ldr, m, kc, r, k_rows);
  else if (out_dtype == SOW_BF16)
    hipLaunchKernelGGL(qr_copy_out_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, Qt, Pt, (bf16_t*)Q, ldq, (bf16_t*)R, ldr, m, kc, r, k_rows);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
