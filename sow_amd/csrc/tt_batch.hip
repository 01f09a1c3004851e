// Batched tensor-train kernels for TT-compressed optimizer state.
//
// Reference: tn_gradient/optimizer/ttadam.py:68-115 keeps Adam's moments as tensor trains and, for EVERY parameter and
// EVERY step, (1) reconstructs m and v (tn_gradient/tt.py:213-247: a chain contraction over the bonds, `to_matrix`
// un-pads), (2) runs the dense Adam update, (3) re-compresses both with TensorTrain.from_matrix (tt.py:48-67: zero-pad
// to mm^d x nn^d, reshape, interleave to (i1,o1,i2,o2,...)) + decompose (tt.py:111-140: d-1 truncated complete-mode QRs).
// Per parameter that is ~40 small launches.  Here the three steps run for ALL parameters of a group at once:
//   tt_adam_eval_kernel   one thread per element of the padded, interleaved tensor: evaluates TT_m and TT_v at its index
//                         (vector-times-core products over the bonds), clamps v < 0 (ttadam.py:84), applies the update of
//                         ttadam.py:92-111 to the parameter and writes the new m / v straight into the layout the first
//                         unfolding L_0 = reshape(r_0 i_0 o_0, -1) of the decomposition reads -- padding elements are zero;
//   per bond k            panel copy-in, ONE Householder-panel launch for every train (qr_panel_batch, the kernel of the
//                         periodic accumulate step: LAPACK sign convention), copy-out of core k = Q[:, :r_{k+1}] and of the
//                         triangular part of R, and R[:, kc:] = Q^T L[:, kc:] -- R, viewed as [r_{k+1} i_{k+1} o_{k+1}, -1],
//                         IS the next unfolding (no copy), the last R is written into the last core.
// Launches per step: 1 + 4 (d - 1) per 16 trains, whatever the number of parameters.
// sow_tt_reconstruct_batch / sow_tt_decompose_batch expose the two halves on their own (TensorTrain.to_matrix /
// from_matrix for many trains).  fp32 throughout; ranks <= 32, order <= 6; anything else stays on the per-train path.
#include "kernels.hpp"
#include <vector>

namespace sow {

constexpr int TT_MAXO = SOW_TT_MAX_ORDER;
constexpr int TT_MAXR = 32;
constexpr int TT_PER = 8;    // items (parameter: m + v train) per launch: the descriptors travel as kernel arguments

struct TtDev {
  float* cores[TT_MAXO];
  int order;
  int ranks[TT_MAXO + 1];
  int in[TT_MAXO];
  int out[TT_MAXO];
  int rows, cols;
};

static TtDev to_dev(const sow_tt_desc& d) {
  TtDev t{};
  t.order = d.order;
  for (int k = 0; k < TT_MAXO; ++k) {
    t.cores[k] = k < d.order ? (float*)d.cores[k] : nullptr;
    t.in[k] = k < d.order ? d.in_dims[k] : 1;
    t.out[k] = k < d.order ? d.out_dims[k] : 1;
  }
  for (int k = 0; k <= TT_MAXO; ++k) t.ranks[k] = k <= d.order ? d.ranks[k] : 1;
  t.rows = d.rows, t.cols = d.cols;
  return t;
}

static bool tt_ok(const sow_tt_desc& d) {
  if (d.order < 1 || d.order > TT_MAXO || d.rows < 1 || d.cols < 1) return false;
  if (d.ranks[0] != 1 || d.ranks[d.order] != 1) return false;
  int64_t pin = 1, pout = 1;
  for (int k = 0; k < d.order; ++k) {
    if (d.in_dims[k] < 1 || d.out_dims[k] < 1 || d.ranks[k] < 1 || d.ranks[k] > TT_MAXR || !d.cores[k]) return false;
    pin *= d.in_dims[k], pout *= d.out_dims[k];
    if (pin * pout > (int64_t)1 << 30) return false;
  }
  return d.rows <= pin && d.cols <= pout;
}

// value of the train at interleaved multi-index (i_k, o_k): vector-times-core over the bonds.  The loops over the order are
// unrolled against TT_MAXO with a predicate, so that the index arrays stay in registers (a runtime-indexed local array
// lives in scratch memory).
template <int MAXR> __device__ __forceinline__ float tt_eval(const TtDev& t, const int (&ik)[TT_MAXO], const int (&ok)[TT_MAXO]) {
  float vec[MAXR];
#pragma unroll
  for (int a = 0; a < MAXR; ++a) vec[a] = 0.f;
  {
    const int r1 = t.ranks[1];
    const float* c = t.cores[0] + (int64_t)(ik[0] * t.out[0] + ok[0]) * r1;
#pragma unroll
    for (int b = 0; b < MAXR; ++b)
      if (b < r1) vec[b] = c[b];
  }
#pragma unroll
  for (int k = 1; k < TT_MAXO; ++k) {
    if (k < t.order) {
      const int rk = t.ranks[k], rn = t.ranks[k + 1];
      const int64_t astride = (int64_t)t.in[k] * t.out[k] * rn;
      const float* c = t.cores[k] + (int64_t)(ik[k] * t.out[k] + ok[k]) * rn;
      float nxt[MAXR];
#pragma unroll
      for (int b = 0; b < MAXR; ++b) nxt[b] = 0.f;
      if constexpr (MAXR <= 8) {
#pragma unroll
        for (int a = 0; a < MAXR; ++a) {
          if (a < rk) {
            const float va = vec[a];
            const float* ca = c + a * astride;
#pragma unroll
            for (int b = 0; b < MAXR; ++b)
              if (b < rn) nxt[b] = fmaf(va, ca[b], nxt[b]);
          }
        }
      } else {
        for (int a = 0; a < rk; ++a) {      // wide bonds: a runtime loop (the unrolled form would be 1024 predicated FMAs per bond)
          float va = 0.f;
#pragma unroll
          for (int q = 0; q < MAXR; ++q) va = (q == a) ? vec[q] : va;
          const float* ca = c + a * astride;
#pragma unroll
          for (int b = 0; b < MAXR; ++b)
            if (b < rn) nxt[b] = fmaf(va, ca[b], nxt[b]);
        }
      }
#pragma unroll
      for (int b = 0; b < MAXR; ++b) vec[b] = nxt[b];
    }
  }
  return vec[0];
}

// interleaved index e of the padded tensor -> (i_k, o_k), matrix row / column
__device__ __forceinline__ void tt_decode(const TtDev& t, int64_t e, int (&ik)[TT_MAXO], int (&ok)[TT_MAXO], int64_t& row,
                                          int64_t& col) {
#pragma unroll
  for (int k = TT_MAXO - 1; k >= 0; --k) {
    ik[k] = 0, ok[k] = 0;
    if (k < t.order) {
      ok[k] = (int)(e % t.out[k]);
      e /= t.out[k];
      ik[k] = (int)(e % t.in[k]);
      e /= t.in[k];
    }
  }
  row = 0, col = 0;
#pragma unroll
  for (int k = 0; k < TT_MAXO; ++k)
    if (k < t.order) row = row * t.in[k] + ik[k], col = col * t.out[k] + ok[k];
}

// ---------------------------------------------------------------------------------------------
struct AdamBatch {
  TtDev m[TT_PER], v[TT_PER];
  float* p[TT_PER];
  const float* g[TT_PER];
  int64_t ldp[TT_PER], ldg[TT_PER];
  float *Lm[TT_PER], *Lv[TT_PER];       // first unfoldings (padded, interleaved) of the new moments
  float step_size[TT_PER], lr_wd[TT_PER];
  int has_state[TT_PER];
  float b1, b2, eps;
  int n;
};

template <int MAXR> __global__ __launch_bounds__(256) void tt_adam_eval_kernel(const AdamBatch b) {
  const int it = blockIdx.y;
  const TtDev& tm = b.m[it];
  int64_t P = 1;
  for (int k = 0; k < tm.order; ++k) P *= (int64_t)tm.in[k] * tm.out[k];
  const bool st = b.has_state[it] != 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < P; e += (int64_t)gridDim.x * 256) {
    int ik[TT_MAXO], ok[TT_MAXO];
    int64_t row, col;
    tt_decode(tm, e, ik, ok, row, col);
    float mv = 0.f, vv = 0.f;
    if (row < tm.rows && col < tm.cols) {
      if (st) {
        mv = tt_eval<MAXR>(tm, ik, ok);
        vv = tt_eval<MAXR>(b.v[it], ik, ok);
        if (vv < 0.f) vv = 0.f;                          // ttadam.py:84
      }
      const float gv = b.g[it][row * b.ldg[it] + col];
      mv = mv * b.b1 + gv * (1.f - b.b1);                // the arithmetic of ttadam_dense_kernel (misc.hip), same order
      vv = vv * b.b2 + gv * gv * (1.f - b.b2);
      float* pp = b.p[it] + row * b.ldp[it] + col;
      float pv = *pp + (mv / (sqrtf(vv) + b.eps)) * (-b.step_size[it]);
      if (b.lr_wd[it] > 0.f) pv = pv + pv * (-b.lr_wd[it]);
      *pp = pv;
    }
    b.Lm[it][e] = mv;                                    // padding: zero (utils.py:78-84)
    b.Lv[it][e] = vv;
  }
}

// out[rows, cols] = TT.to_matrix
struct ReconBatch {
  TtDev t[2 * TT_PER];
  float* out[2 * TT_PER];
  int64_t ld[2 * TT_PER];
  int n;
};
template <int MAXR> __global__ __launch_bounds__(256) void tt_reconstruct_kernel(const ReconBatch b) {
  const int it = blockIdx.y;
  const TtDev& t = b.t[it];
  const int64_t n = (int64_t)t.rows * t.cols;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    int64_t row = idx / t.cols, col = idx % t.cols;
    int ik[TT_MAXO], ok[TT_MAXO];
    int64_t r = row, c = col;
#pragma unroll
    for (int k = TT_MAXO - 1; k >= 0; --k) {
      ik[k] = 0, ok[k] = 0;
      if (k < t.order) {
        ik[k] = (int)(r % t.in[k]), r /= t.in[k];
        ok[k] = (int)(c % t.out[k]), c /= t.out[k];
      }
    }
    b.out[it][row * b.ld[it] + col] = tt_eval<MAXR>(t, ik, ok);
  }
}

// dense [rows, cols] -> padded interleaved L_0
struct PadBatch {
  TtDev t[2 * TT_PER];
  const float* src[2 * TT_PER];
  int64_t ld[2 * TT_PER];
  float* L[2 * TT_PER];
  int n;
};
__global__ __launch_bounds__(256) void tt_pad_interleave_kernel(const PadBatch b) {
  const int it = blockIdx.y;
  const TtDev& t = b.t[it];
  int64_t P = 1;
  for (int k = 0; k < t.order; ++k) P *= (int64_t)t.in[k] * t.out[k];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < P; e += (int64_t)gridDim.x * 256) {
    int ik[TT_MAXO], ok[TT_MAXO];
    int64_t row, col;
    tt_decode(t, e, ik, ok, row, col);
    b.L[it][e] = (row < t.rows && col < t.cols) ? b.src[it][row * b.ld[it] + col] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// one stage of the sequential truncated QR of many trains
struct StageItem {
  const float* L;   // [m, ncols] row-major
  float* Pt;        // [kc][m] column-major panel
  float* Qt;        // [r][m]
  float* core;      // [m, r]  = Q[:, :r]
  float* R;         // [r, ncols] (the next unfolding / the last core)
  int m, ncols, kc, r;
};
struct StageBatch {
  StageItem it[2 * TT_PER];
  int n;
};
__global__ __launch_bounds__(256) void tt_stage_copy_in_kernel(const StageBatch b) {
  const StageItem& s = b.it[blockIdx.y];
  const int64_t n = (int64_t)s.m * s.kc;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / s.kc), c = (int)(idx % s.kc);
    s.Pt[(int64_t)c * s.m + i] = s.L[(int64_t)i * s.ncols + c];
  }
}
__global__ __launch_bounds__(256) void tt_stage_copy_out_kernel(const StageBatch b) {
  const StageItem& s = b.it[blockIdx.y];
  const int64_t nq = (int64_t)s.m * s.r, nr = (int64_t)s.r * s.kc;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nq + nr; idx += (int64_t)gridDim.x * 256) {
    if (idx < nq) {
      const int i = (int)(idx / s.r), c = (int)(idx % s.r);
      s.core[idx] = s.Qt[(int64_t)c * s.m + i];
    } else {
      const int64_t e = idx - nq;
      const int i = (int)(e / s.kc), c = (int)(e % s.kc);
      s.R[(int64_t)i * s.ncols + c] = (i <= c && i < s.m) ? s.Pt[(int64_t)c * s.m + i] : 0.f;
    }
  }
}
// R[j, col] = sum_i Q[i, j] L[i, col] for col >= kc.  Wide unfoldings (the first bonds: thousands of columns, few rows): one
// thread per (j, col), consecutive threads on consecutive columns (coalesced L rows, broadcast Q).  Narrow unfoldings (the
// last bonds: tens of columns, hundreds of rows): one WAVE per (j, col), the lanes stride over the rows and sum by shuffles --
// a thread per output would walk 512 rows alone.
__global__ __launch_bounds__(256) void tt_stage_rrest_kernel(const StageBatch b) {
  const StageItem& s = b.it[blockIdx.y];
  const int rest = s.ncols - s.kc;
  if (rest <= 0) return;
  const int64_t nout = (int64_t)s.r * rest;
  if (rest >= 512) {
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < nout; o += (int64_t)gridDim.x * 256) {
      const int j = (int)(o / rest), col = s.kc + (int)(o % rest);
      const float* q = s.Qt + (int64_t)j * s.m;
      const float* l = s.L + col;
      float acc = 0.f;
      for (int i = 0; i < s.m; ++i) acc = fmaf(q[i], l[(int64_t)i * s.ncols], acc);
      s.R[(int64_t)j * s.ncols + col] = acc;
    }
  } else {
    const int lane = threadIdx.x & 63;
    for (int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); o < nout; o += (int64_t)gridDim.x * 4) {
      const int j = (int)(o / rest), col = s.kc + (int)(o % rest);
      const float* q = s.Qt + (int64_t)j * s.m;
      const float* l = s.L + col;
      float acc = 0.f;
      for (int i = lane; i < s.m; i += 64) acc = fmaf(q[i], l[(int64_t)i * s.ncols], acc);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) s.R[(int64_t)j * s.ncols + col] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
struct TtWs {
  size_t off_L0, off_Ra, off_Rb, off_Pt, off_Qt, total;
};
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static TtWs tt_ws_plan(const sow_tt_desc& d) {
  TtWs w{};
  int64_t P = 1;
  for (int k = 0; k < d.order; ++k) P *= (int64_t)d.in_dims[k] * d.out_dims[k];
  int64_t ncols = P, max_r = 0, max_pt = 0, max_qt = 0;
  for (int k = 0; k + 1 < d.order; ++k) {
    const int64_t m = (int64_t)d.ranks[k] * d.in_dims[k] * d.out_dims[k];
    ncols = ncols / ((int64_t)d.in_dims[k] * d.out_dims[k]);
    const int64_t r = d.ranks[k + 1];
    int64_t kc = r < m ? r : m;
    if (ncols < kc) kc = ncols;
    if (r * ncols > max_r) max_r = r * ncols;
    if (kc * m > max_pt) max_pt = kc * m;
    if (r * m > max_qt) max_qt = r * m;
  }
  size_t off = 0;
  w.off_L0 = off, off += al256((size_t)P * 4);
  w.off_Ra = off, off += al256((size_t)max_r * 4);
  w.off_Rb = off, off += al256((size_t)max_r * 4);
  w.off_Pt = off, off += al256((size_t)max_pt * 4);
  w.off_Qt = off, off += al256((size_t)max_qt * 4);
  w.total = off;
  return w;
}
static bool tt_decomposable(const sow_tt_desc& d) {
  if (!tt_ok(d)) return false;
  for (int k = 0; k + 1 < d.order; ++k) {
    const int64_t m = (int64_t)d.ranks[k] * d.in_dims[k] * d.out_dims[k];
    if (d.ranks[k + 1] > m) return false;                        // Q[:, :r] needs r <= m (as sow_qr_thin)
    if (((size_t)m + 64 + 16) * 4 > 150 * 1024) return false;    // panel LDS
  }
  return true;
}
static inline int grid_x(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

// the d - 1 QR stages of up to 2 * TT_PER trains whose L_0 sits at ws + off_L0
static int decompose_stages(const sow_tt_desc* tts, char* const* ws, int n, hipStream_t stream) {
  int max_order = 0;
  for (int i = 0; i < n; ++i) max_order = tts[i].order > max_order ? tts[i].order : max_order;
  std::vector<int64_t> ncols((size_t)n);
  std::vector<const float*> Lcur((size_t)n);
  for (int i = 0; i < n; ++i) {
    int64_t P = 1;
    for (int k = 0; k < tts[i].order; ++k) P *= (int64_t)tts[i].in_dims[k] * tts[i].out_dims[k];
    ncols[(size_t)i] = P;
    Lcur[(size_t)i] = (const float*)(ws[i] + tt_ws_plan(tts[i]).off_L0);
    if (tts[i].order == 1) {   // a single core IS the padded tensor
      hipError_t e = hipMemcpyAsync(tts[i].cores[0], Lcur[(size_t)i], (size_t)P * 4, hipMemcpyDeviceToDevice, stream);
      if (e != hipSuccess) return (int)e;
    }
  }
  for (int k = 0; k + 1 < max_order; ++k) {
    StageBatch sb{};
    QrItem qi[2 * TT_PER];
    int64_t max_in = 0, max_out = 0, max_rest = 0;
    for (int i = 0; i < n; ++i) {
      const sow_tt_desc& d = tts[i];
      if (k + 1 >= d.order) continue;
      const TtWs w = tt_ws_plan(d);
      const int m = d.ranks[k] * d.in_dims[k] * d.out_dims[k];
      const int64_t nc = ncols[(size_t)i] / ((int64_t)d.in_dims[k] * d.out_dims[k]);
      const int r = d.ranks[k + 1];
      int kc = r < m ? r : m;
      if (nc < kc) kc = (int)nc;
      StageItem& s = sb.it[sb.n];
      s.L = Lcur[(size_t)i], s.Pt = (float*)(ws[i] + w.off_Pt), s.Qt = (float*)(ws[i] + w.off_Qt);
      s.core = (float*)d.cores[k];
      s.R = (k + 2 == d.order) ? (float*)d.cores[k + 1] : (float*)(ws[i] + ((k & 1) ? w.off_Rb : w.off_Ra));
      s.m = m, s.ncols = (int)nc, s.kc = kc, s.r = r;
      qi[sb.n] = QrItem{s.Pt, s.Qt, m, kc, r, 0};
      ++sb.n;
      if ((int64_t)m * kc > max_in) max_in = (int64_t)m * kc;
      if ((int64_t)m * r + (int64_t)r * kc > max_out) max_out = (int64_t)m * r + (int64_t)r * kc;
      {
        const int64_t rest = nc - kc;
        const int64_t work = rest >= 512 ? (int64_t)r * rest : (int64_t)r * rest * 64;   // threads this item wants
        if (rest > 0 && work > max_rest) max_rest = work;
      }
      ncols[(size_t)i] = nc;
      Lcur[(size_t)i] = s.R;
    }
    if (sb.n == 0) continue;
    hipLaunchKernelGGL(tt_stage_copy_in_kernel, dim3(grid_x(max_in), sb.n), dim3(256), 0, stream, sb);
    SOW_CHECK_LAUNCH();
    int rc = launch_qr_panel_batch(qi, sb.n, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(tt_stage_copy_out_kernel, dim3(grid_x(max_out), sb.n), dim3(256), 0, stream, sb);
    SOW_CHECK_LAUNCH();
    if (max_rest > 0) {
      hipLaunchKernelGGL(tt_stage_rrest_kernel, dim3(grid_x(max_rest), sb.n), dim3(256), 0, stream, sb);
      SOW_CHECK_LAUNCH();
    }
  }
  return SOW_OK;
}

static int max_rank_of(const sow_tt_desc& d) {
  int r = 1;
  for (int k = 0; k <= d.order; ++k) r = d.ranks[k] > r ? d.ranks[k] : r;
  return r;
}

}  // namespace sow

using namespace sow;

extern "C" {

size_t sow_tt_decompose_workspace_bytes(const sow_tt_desc* tt) {
  if (!tt || !tt_ok(*tt)) return 0;
  return tt_ws_plan(*tt).total + 256;
}

int sow_tt_reconstruct_batch(const sow_tt_desc* tts, void* const* out, const int64_t* ld_out, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!tts || !out || !ld_out) return SOW_ERR_NULL;
  for (int i = 0; i < n; ++i) {
    if (!tt_ok(tts[i])) return SOW_ERR_UNSUPPORTED;
    if (!out[i]) return SOW_ERR_NULL;
    if (ld_out[i] < tts[i].cols) return SOW_ERR_SHAPE;
  }
  for (int base = 0; base < n; base += 2 * TT_PER) {
    ReconBatch b{};
    b.n = n - base < 2 * TT_PER ? n - base : 2 * TT_PER;
    int64_t max_n = 0;
    int maxr = 1;
    for (int i = 0; i < b.n; ++i) {
      b.t[i] = to_dev(tts[base + i]), b.out[i] = (float*)out[base + i], b.ld[i] = ld_out[base + i];
      const int64_t ne = (int64_t)tts[base + i].rows * tts[base + i].cols;
      max_n = ne > max_n ? ne : max_n;
      const int mr = max_rank_of(tts[base + i]);
      maxr = mr > maxr ? mr : maxr;
    }
    if (maxr <= 8) hipLaunchKernelGGL(tt_reconstruct_kernel<8>, dim3(grid_x(max_n), b.n), dim3(256), 0, stream, b);
    else hipLaunchKernelGGL(tt_reconstruct_kernel<TT_MAXR>, dim3(grid_x(max_n), b.n), dim3(256), 0, stream, b);
    SOW_CHECK_LAUNCH();
  }
  return SOW_OK;
}

int sow_tt_decompose_batch(const sow_tt_desc* tts, const void* const* mats, const int64_t* ld, int n, void* const* workspaces,
                           const size_t* workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!tts || !mats || !ld || !workspaces || !workspace_bytes) return SOW_ERR_NULL;
  for (int i = 0; i < n; ++i) {
    if (!tt_decomposable(tts[i])) return SOW_ERR_UNSUPPORTED;
    if (!mats[i] || !workspaces[i]) return SOW_ERR_NULL;
    if (ld[i] < tts[i].cols) return SOW_ERR_SHAPE;
    if (workspace_bytes[i] < tt_ws_plan(tts[i]).total + 255) return SOW_ERR_WORKSPACE;
  }
  for (int base = 0; base < n; base += 2 * TT_PER) {
    PadBatch b{};
    b.n = n - base < 2 * TT_PER ? n - base : 2 * TT_PER;
    char* ws[2 * TT_PER];
    int64_t max_p = 0;
    for (int i = 0; i < b.n; ++i) {
      const sow_tt_desc& d = tts[base + i];
      ws[i] = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspaces[base + i]) + 255) & ~(uintptr_t)255);
      b.t[i] = to_dev(d), b.src[i] = (const float*)mats[base + i], b.ld[i] = ld[base + i];
      b.L[i] = (float*)(ws[i] + tt_ws_plan(d).off_L0);
      int64_t P = 1;
      for (int k = 0; k < d.order; ++k) P *= (int64_t)d.in_dims[k] * d.out_dims[k];
      max_p = P > max_p ? P : max_p;
    }
    hipLaunchKernelGGL(tt_pad_interleave_kernel, dim3(grid_x(max_p), b.n), dim3(256), 0, stream, b);
    SOW_CHECK_LAUNCH();
    int rc = decompose_stages(tts + base, ws, b.n, stream);
    if (rc) return rc;
  }
  return SOW_OK;
}

size_t sow_ttadam_workspace_bytes(const sow_tt_desc* tt) {
  if (!tt || !tt_ok(*tt)) return 0;
  return 2 * (tt_ws_plan(*tt).total + 256);
}

int sow_ttadam_batch(const sow_ttadam_item* items, int n, float beta1, float beta2, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!items) return SOW_ERR_NULL;
  for (int i = 0; i < n; ++i) {
    const sow_ttadam_item& it = items[i];
    if (!tt_decomposable(it.m) || !tt_decomposable(it.v)) return SOW_ERR_UNSUPPORTED;
    if (it.m.order != it.v.order || it.m.rows != it.v.rows || it.m.cols != it.v.cols) return SOW_ERR_SHAPE;
    for (int k = 0; k < it.m.order; ++k)
      if (it.m.in_dims[k] != it.v.in_dims[k] || it.m.out_dims[k] != it.v.out_dims[k]) return SOW_ERR_SHAPE;
    if (!it.param || !it.grad || !it.workspace) return SOW_ERR_NULL;
    if (it.ld_param < it.m.cols || it.ld_grad < it.m.cols) return SOW_ERR_SHAPE;
    if (it.workspace_bytes < 2 * (tt_ws_plan(it.m).total + 256)) return SOW_ERR_WORKSPACE;
  }
  for (int base = 0; base < n; base += TT_PER) {
    AdamBatch b{};
    b.n = n - base < TT_PER ? n - base : TT_PER;
    b.b1 = beta1, b.b2 = beta2, b.eps = eps;
    sow_tt_desc tts[2 * TT_PER];
    char* ws[2 * TT_PER];
    int64_t max_p = 0;
    int maxr = 1;
    for (int i = 0; i < b.n; ++i) {
      const sow_ttadam_item& it = items[base + i];
      const TtWs w = tt_ws_plan(it.m);
      char* w0 = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(it.workspace) + 255) & ~(uintptr_t)255);
      ws[2 * i] = w0, ws[2 * i + 1] = w0 + w.total;
      tts[2 * i] = it.m, tts[2 * i + 1] = it.v;
      b.m[i] = to_dev(it.m), b.v[i] = to_dev(it.v);
      b.p[i] = it.param, b.g[i] = it.grad, b.ldp[i] = it.ld_param, b.ldg[i] = it.ld_grad;
      b.Lm[i] = (float*)(ws[2 * i] + w.off_L0), b.Lv[i] = (float*)(ws[2 * i + 1] + w.off_L0);
      b.step_size[i] = it.step_size, b.lr_wd[i] = it.lr_times_wd, b.has_state[i] = it.has_state;
      int64_t P = 1;
      for (int k = 0; k < it.m.order; ++k) P *= (int64_t)it.m.in_dims[k] * it.m.out_dims[k];
      max_p = P > max_p ? P : max_p;
      const int mr = max_rank_of(it.m) > max_rank_of(it.v) ? max_rank_of(it.m) : max_rank_of(it.v);
      maxr = mr > maxr ? mr : maxr;
    }
    if (maxr <= 8) hipLaunchKernelGGL(tt_adam_eval_kernel<8>, dim3(grid_x(max_p), b.n), dim3(256), 0, stream, b);
    else hipLaunchKernelGGL(tt_adam_eval_kernel<TT_MAXR>, dim3(grid_x(max_p), b.n), dim3(256), 0, stream, b);
    SOW_CHECK_LAUNCH();
    // the old cores have been read (stream order): the new ones are written in place
    int rc = decompose_stages(tts, ws, 2 * b.n, stream);
    if (rc) return rc;
  }
  return SOW_OK;
}

}  // extern "C"
