// The periodic step of a whole model in a handful of launches (sow_accumulate_batch, include/sow_amd.h).
//
// Reference: tn_gradient/prepare.py:219-222 loops SoWLinear.accumulate() (tn_gradient/layer/sow.py:128-178) over every
// layer -- 56 (llama_60m) to 160 (llama-7b) times: a rank-r update of the dense accumulator, a full QR of a fresh
// [in, out] Gaussian of which only Q[:, :r] is kept, B <- 0.  The layers are independent, every piece is small and
// latency-bound, so the batch entry point runs each PHASE once for all layers:
//   1. acc_l = beta_l * acc_l + scale_l * A_l . B_l          one grid, (64 x 64 tile, layer) per workgroup
//   2. panel copy-in, Householder panel + org2r, copy-out     one grid each, one workgroup per layer for the panel
//      (the panel kernel is latency-bound on one CU: 0.1 - 1.3 ms; 56 - 112 of them run side by side in ONE launch)
//   3. B_l <- 0                                               one multi-tensor memset
// A is overwritten IN PLACE by the new orthonormal factor after the update has consumed it (stream order), so the
// caller's parameter storage -- and any flat bucket that views it -- stays where it is.
#include "kernels.hpp"
#include "qr_panel.hpp"
#include <vector>

namespace sow {

size_t qr_panel_lds_bytes(int m, int kc);   // qr.hip

// ---------------------------------------------------------------------------------------------
// 1. batched rank-r update (r <= 64), fp32 accumulation, one rounding.  HBM-bound on acc (1 read + 1 write).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rank_update_batch_kernel(const AccBatch b) {
  const AccItem& it = b.it[blockIdx.y];
  const int tiles_n = (it.d_out + 63) / 64, tiles = ((it.d_in + 63) / 64) * tiles_n;
  __shared__ float As[64][65];   // [row][k]
  __shared__ float Bs[64][65];   // [k][col]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const T* A = (const T*)it.A;
  const T* B = (const T*)it.B;
  T* acc = (T*)it.acc;
  const int r = it.r;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      const int i = idx >> 6, k = idx & 63;
      As[i][k] = (m0 + i < it.d_in && k < r) ? to_f32(A[(int64_t)(m0 + i) * r + k]) : 0.f;
      const int kk = idx >> 6, j = idx & 63;
      Bs[kk][j] = (kk < r && n0 + j < it.d_out) ? to_f32(B[(int64_t)kk * it.d_out + n0 + j]) : 0.f;
    }
    __syncthreads();
    float c[4][4] = {};
    for (int k = 0; k < r; ++k) {
      float a[4], bb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = As[ty * 4 + u][k], bb[u] = Bs[k][tx * 4 + u];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) c[u][v] = fmaf(a[u], bb[v], c[u][v]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = m0 + ty * 4 + u;
      if (i >= it.d_in) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int j = n0 + tx * 4 + v;
        if (j >= it.d_out) continue;
        T* dst = acc + (int64_t)i * it.d_out + j;
        float val = it.scale * c[u][v];
        if (it.beta != 0.f) val += it.beta * to_f32(*dst);
        *dst = from_f32<T>(val);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// 2. batched Householder panels (the single-matrix kernels of qr.hip with a per-layer parameter block)
// ---------------------------------------------------------------------------------------------
template <typename Tin> __global__ void qr_copy_in_batch_kernel(const AccBatch b) {
  const AccItem& it = b.it[blockIdx.y];
  if (!it.draw) return;
  const Tin* W = (const Tin*)it.draw;
  const int m = it.d_in, kc = it.kc;
  const int64_t n = (int64_t)m * kc;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / kc), c = (int)(idx % kc);
    it.Pt[(int64_t)c * m + i] = to_f32(W[(int64_t)i * it.ld_draw + c]);
  }
}

// the panel kernel is latency-bound (0.5 - 1.3 ms per matrix on one CU) and needs four words per layer: its own compact
// descriptor block (kernels.hpp: QrItem) lets ONE launch cover up to 112 matrices (the slowest matrix sets the time, not
// the sum over launches)
constexpr int QR_MAXB = 112;   // 112 x 32 bytes of kernel arguments
struct QrBatch {
  QrItem it[QR_MAXB];
  int n;
};
__global__ __launch_bounds__(1024) void qr_panel_batch_kernel(const QrBatch b) {
  extern __shared__ __attribute__((aligned(16))) float qsm_b[];
  const QrItem& it = b.it[blockIdx.x];
  qr_panel_body(it.Pt, it.Qt, it.m, it.kc, it.r_new, qsm_b);
}

// Householder panel + org2r of n independent matrices, one workgroup each, QR_MAXB per launch (also used by tt_batch.hip)
int launch_qr_panel_batch(const QrItem* items, int n, hipStream_t stream) {
  for (int i = 0; i < n; ++i) {
    if (items[i].m <= 0 || items[i].kc <= 0 || items[i].kc > items[i].m || items[i].r_new <= 0 || items[i].r_new > items[i].m)
      return SOW_ERR_SHAPE;
    if (qr_panel_lds_bytes(items[i].m, items[i].kc) > 150 * 1024) return SOW_ERR_UNSUPPORTED;
  }
  for (int base = 0; base < n; base += QR_MAXB) {
    QrBatch q{};
    q.n = n - base < QR_MAXB ? n - base : QR_MAXB;
    size_t max_lds = 0;
    for (int i = 0; i < q.n; ++i) {
      q.it[i] = items[base + i];
      const size_t l = qr_panel_lds_bytes(q.it[i].m, q.it[i].kc);
      if (l > max_lds) max_lds = l;
    }
    SOW_SET_MAX_LDS_ONCE(150 * 1024, qr_panel_batch_kernel);
    hipLaunchKernelGGL(qr_panel_batch_kernel, dim3(q.n), dim3(1024), max_lds, stream, q);
    SOW_CHECK_LAUNCH();
  }
  return SOW_OK;
}

template <typename Tout> __global__ void qr_copy_out_batch_kernel(const AccBatch b) {
  const AccItem& it = b.it[blockIdx.y];
  if (!it.draw) return;
  Tout* Q = (Tout*)it.A_new;
  const int m = it.d_in, r = it.r_new;
  const int64_t nq = (int64_t)m * r;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nq; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / r), c = (int)(idx % r);
    Q[(int64_t)i * r + c] = from_f32<Tout>(it.Qt[(int64_t)c * m + i]);
  }
}

int launch_accumulate_batch(const AccItem* items, int n, int dtype, hipStream_t stream) {
  // phase-major over the whole list (stream order = dependency order): 1. rank-r updates and panel copy-in, 48 layers per
  // launch; 2. the Householder panels, up to 112 layers per launch; 3. copy-out (new A, in place: every update has read
  // the old A by then)
  auto batch_of = [&](int base, AccBatch& b, int& max_tiles, int& max_m, bool& any_qr) {
    b = AccBatch{};
    b.n = n - base < ACC_MAXB ? n - base : ACC_MAXB;
    max_tiles = 0, max_m = 0, any_qr = false;
    for (int i = 0; i < b.n; ++i) {
      b.it[i] = items[base + i];
      const AccItem& it = b.it[i];
      const int tiles = ((it.d_in + 63) / 64) * ((it.d_out + 63) / 64);
      if (tiles > max_tiles) max_tiles = tiles;
      if (it.draw) {
        any_qr = true;
        if (it.d_in > max_m) max_m = it.d_in;
      }
    }
  };
  for (int i = 0; i < n; ++i)
    if (items[i].draw && qr_panel_lds_bytes(items[i].d_in, items[i].kc) > 150 * 1024) return SOW_ERR_UNSUPPORTED;
  for (int base = 0; base < n; base += ACC_MAXB) {
    AccBatch b;
    int max_tiles, max_m;
    bool any_qr;
    batch_of(base, b, max_tiles, max_m, any_qr);
    const int gx = max_tiles < 64 ? max_tiles : 64;   // tile loop inside: keeps the grid at <= 64 x n workgroups
    if (dtype == SOW_F32)
      hipLaunchKernelGGL(rank_update_batch_kernel<float>, dim3(gx, b.n), dim3(256), 0, stream, b);
    else
      hipLaunchKernelGGL(rank_update_batch_kernel<bf16_t>, dim3(gx, b.n), dim3(256), 0, stream, b);
    SOW_CHECK_LAUNCH();
    if (!any_qr) continue;
    const int gc = (max_m * 64 + 255) / 256 < 64 ? (max_m * 64 + 255) / 256 : 64;
    if (dtype == SOW_F32)
      hipLaunchKernelGGL(qr_copy_in_batch_kernel<float>, dim3(gc, b.n), dim3(256), 0, stream, b);
    else
      hipLaunchKernelGGL(qr_copy_in_batch_kernel<bf16_t>, dim3(gc, b.n), dim3(256), 0, stream, b);
    SOW_CHECK_LAUNCH();
  }
  {
    std::vector<QrItem> qs;
    for (int i = 0; i < n; ++i)
      if (items[i].draw) qs.push_back(QrItem{items[i].Pt, items[i].Qt, items[i].d_in, items[i].kc, items[i].r_new, 0});
    const int rc = launch_qr_panel_batch(qs.data(), (int)qs.size(), stream);
    if (rc) return rc;
  }
  for (int base = 0; base < n; base += ACC_MAXB) {
    AccBatch b;
    int max_tiles, max_m;
    bool any_qr;
    batch_of(base, b, max_tiles, max_m, any_qr);
    if (!any_qr) continue;
    const int gc = (max_m * 64 + 255) / 256 < 64 ? (max_m * 64 + 255) / 256 : 64;
    if (dtype == SOW_F32)
      hipLaunchKernelGGL(qr_copy_out_batch_kernel<float>, dim3(gc, b.n), dim3(256), 0, stream, b);
    else
      hipLaunchKernelGGL(qr_copy_out_batch_kernel<bf16_t>, dim3(gc, b.n), dim3(256), 0, stream, b);
    SOW_CHECK_LAUNCH();
  }
  return SOW_OK;
}

}  // namespace sow
