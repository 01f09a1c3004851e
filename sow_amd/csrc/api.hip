// extern "C" entry points of libsow_amd.so (declared in include/sow_amd.h).
// Host-side dispatch only: picks the fused low-rank chain / skinny-TN kernels for r <= 64 and composes
// the dense GEMM kernel for everything else.  No allocation, no synchronisation; the only process-wide state is the
// table of kernel-selection switches below (atomics, read from the environment once).
#include "kernels.hpp"
#include <cstring>
#include <vector>
#include <cstdio>
#include <cstdlib>

namespace sow {

// column sums of a [T, D] matrix (fallback path only: r_live > 64 with bias)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* M, int64_t ld, int64_t rows, int D, T* out, float beta) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float s = 0.f;
  if (c < D)
    for (int64_t i = w; i < rows; i += 4) s += to_f32(M[i * ld + c]);
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < D) {
    float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (beta != 0.f) v += beta * to_f32(out[c]);
    out[c] = from_f32<T>(v);
  }
}
}  // namespace sow

// ---- kernel-selection switches (common.hpp: enum Switch) ----------------------------------------------
namespace sow {
static const char* const kSwitchNames[SW_COUNT] = {"FORCE_CHAIN_V1", "NO_SHORT_SPLIT", "NO_FUSED_H", "FORCE_GEMM_V1", "TN_NARROW",
                                                   "NO_GEMM3S",      "GEMM3S",         "GEMM3",      "NO_GROUPED",     "NO_PERSIST",     "NO_NT_STORE",    "NT_LOAD",        "NO_PAIR_FLUSH",  "F32_EXACT",
                                                   "NO_PARK16",      "TN_NO_NT_LOAD",  "NO_TN_ROWS",     "GEMM4",          "NO_GEMM4H",      "NO_CHAIN3F",     "NO_TN_F32Q",     "NO_SPLITK"};
static std::atomic<int> g_switch[SW_COUNT];
static std::once_flag g_switch_once;
static void switches_from_env() {
  for (int i = 0; i < SW_COUNT; ++i) {
    char name[64];
    snprintf(name, sizeof name, "SOW_AMD_%s", kSwitchNames[i]);
    const char* v = getenv(name);
    // tri-state switches take "0" / "1"; for the boolean ones any value (even empty) means on, as before
    const bool tri = i == SW_GEMM3S || i == SW_GEMM3 || i == SW_GEMM4;
    g_switch[i].store(!v ? -1 : (tri ? (v[0] != '0') : 1), std::memory_order_relaxed);
  }
}
int sw(int which) {
  std::call_once(g_switch_once, switches_from_env);
  return g_switch[which].load(std::memory_order_relaxed);
}
}  // namespace sow

using namespace sow;

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline size_t esize(int dtype) { return dtype == SOW_F32 ? 4 : 2; }
static inline bool ok_dtype(int d) { return d == SOW_F32 || d == SOW_BF16; }
static inline bool al4p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }
static inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- short inputs ---------------------------------------------------------------------------------
// T / 64 workgroups cannot fill 256 CUs: below SHORT_NTB token blocks the streaming chain is launched as
// phase 1 split over K (fp32 partials + h_reduce) and phase 2 split over the output columns (kernels.hpp).
constexpr int SHORT_NTB = 128;
static inline int short_want(int ntb) { return ntb > 0 ? (256 + ntb - 1) / ntb : 1; }
static size_t short_hp_bytes(int64_t T, int d_in, int d_out, int r_live, int dtype) {
  const int ntb = ceil_div(T, 64);
  if (r_live > 64 || ntb <= 0 || ntb > SHORT_NTB) return 0;
  const int dmax = d_in > d_out ? d_in : d_out;
  int ks = short_want(ntb);
  const int nst = (dmax + 63) / 64;
  if (ks > nst) ks = nst;
  return al256((size_t)ks * (size_t)T * 64 * sizeof(float));
}
// the live-factor chain of one direction (Hsave required); returns SOW_ERR_UNSUPPORTED when the split does not apply
static int launch_chain_short(const ChainParams& p, int dtype, bool bwd, float* hpartial, hipStream_t stream) {
  const int ntb = ceil_div(p.M, 64);
  const bool f32 = dtype == SOW_F32;
  if (!hpartial || !p.Hsave || ntb > SHORT_NTB || !(f32 ? chain2f_supported(p, dtype) : chain2_supported(p, dtype)) ||
      sw_on(SW_FORCE_CHAIN_V1) || sw_on(SW_NO_SHORT_SPLIT))
    return SOW_ERR_UNSUPPORTED;
  const int want = short_want(ntb);
  const int nst = (p.D1 + 63) / 64, nsl = (p.D2 + 63) / 64;
  if (nst + nsl < 24) return SOW_ERR_UNSUPPORTED;   // tiny layers: three launches cost more than the idle CUs (26 vs 19 us at 64 x 256 x 256)
  int rc;
  // phase 1: H = scale * X . F1
  int ks = want < nst ? want : nst;
  if (ks > 1) {
    ChainParams a = p;
    a.ntb = ntb, a.st_per = ceil_div(nst, ks), a.sl_per = 0, a.Hpartial = hpartial, a.Hload = nullptr;
    ks = ceil_div(nst, a.st_per);
    rc = f32 ? launch_chain2f(a, bwd, stream) : launch_chain2(a, bwd, stream);
    if (rc) return rc == SOW_ERR_ALIGN ? SOW_ERR_UNSUPPORTED : rc;
    rc = launch_h_reduce(hpartial, ks, p.Hsave, p.M, p.rb, p.scale, dtype, stream);
    if (rc) return rc;
  } else {
    ChainParams a = p;
    a.Y = nullptr, a.D2 = 0, a.bias = nullptr;   // H-only mode
    rc = f32 ? launch_chain2f(a, bwd, stream) : launch_chain2(a, bwd, stream);
    if (rc) return rc == SOW_ERR_ALIGN ? SOW_ERR_UNSUPPORTED : rc;
  }
  if (nsl == 0) return SOW_OK;
  // phase 2: Y = beta * Y + H . F2 + bias
  ChainParams b = p;
  const int kn = want < nsl ? want : nsl;
  b.ntb = ntb, b.st_per = 0, b.sl_per = ceil_div(nsl, kn), b.Hpartial = nullptr, b.Hload = p.Hsave, b.Hsave = nullptr;
  return f32 ? launch_chain2f(b, bwd, stream) : launch_chain2(b, bwd, stream);
}

// row-major A, plain product: the streaming kernel when it fills the chip, else the 128x128 kernel
static int gemm_auto(const void* A, int64_t lda, const void* B, int64_t ldb, bool transB, void* C, int64_t ldc,
                     const void* bias, int64_t M, int N, int K, float alpha, float beta, int dtype, hipStream_t stream,
                     void* ws = nullptr, size_t ws_bytes = 0) {
  if (gemm2_supported(A, lda, B, ldb, transB, nullptr, 0, nullptr, 0, C, ldc, bias, M, N, K, dtype))
    return launch_gemm2(A, lda, B, ldb, transB, nullptr, 0, nullptr, 0, 0, C, ldc, bias, M, N, K, alpha, beta, stream, ws, ws_bytes);
  return launch_gemm(A, lda, false, B, ldb, transB, C, ldc, bias, M, N, K, alpha, beta, dtype, stream);
}

extern "C" {

int sow_version(void) { return 111; }

int sow_set_switch(const char* name, int value) {
  if (!name) return SOW_ERR_NULL;
  (void)sw(0);   // make sure the environment has been read first
  for (int i = 0; i < SW_COUNT; ++i)
    if (!strcmp(name, kSwitchNames[i])) {
      g_switch[i].store(value < 0 ? -1 : (value != 0), std::memory_order_relaxed);
      return SOW_OK;
    }
  return SOW_ERR_UNSUPPORTED;
}

#ifdef SOW_STAMPS
// debug builds only (not declared in include/sow_amd.h): device buffer for the in-kernel timeline of chain2_kernel
int sow_debug_set_stamps(void* buf) {
  g_chain2_stamps = buf;
  return SOW_OK;
}
#endif

int sow_get_switch(const char* name) {
  if (!name) return SOW_ERR_NULL;
  for (int i = 0; i < SW_COUNT; ++i)
    if (!strcmp(name, kSwitchNames[i])) return sw(i);
  return SOW_ERR_UNSUPPORTED;
}

const char* sow_error_string(int code) {
  switch (code) {
    case SOW_OK: return "ok";
    case SOW_ERR_NULL: return "null pointer argument";
    case SOW_ERR_SHAPE: return "invalid shape argument";
    case SOW_ERR_DTYPE: return "unsupported dtype";
    case SOW_ERR_ALIGN: return "misaligned pointer";
    case SOW_ERR_WORKSPACE: return "workspace too small";
    case SOW_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

size_t sow_h_save_elems(int64_t T, int r_live) { return (size_t)T * (size_t)(r_live <= 64 ? 64 : r_live); }

// workspace carve (identical in the query and in the calls)
struct WsPlan {
  size_t off_dh, off_t, off_apad, off_hp, off_p0, off_p1, off_planes, planes_bytes, off_sk, sk_bytes, total;
  int ns, slab_len;
  int ns_cap;   // slabs the partial regions can hold (group-planned slab counts may exceed the single-layer choice)
};
constexpr int64_t C3F_MIN_T = 8192;   // chain3f_supported's threshold
static WsPlan plan_ws(int64_t T, int d_in, int d_out, int r_live, int r_acc, int acc_kind, int dtype) {
  WsPlan w{};
  const size_t es = esize(dtype);
  size_t off = 0;
  w.off_dh = off;
  off += al256((size_t)T * (size_t)(r_live <= 64 ? 64 : r_live) * es);
  w.off_t = off;
  if (acc_kind == SOW_ACC_LOWRANK && r_acc > 64) off += al256((size_t)T * r_acc * es);
  w.off_apad = off;   // A zero-padded to [d_in, 64]: the k-contiguous K-extension operand of the dense backward
  if (acc_kind == SOW_ACC_DENSE && r_live <= 64) off += al256((size_t)d_in * 64 * es);
  w.off_hp = off;     // fp32 partial H of the short-T split: splits * T * 64 floats, splits <= 256 / ceil(T / 64)
  off += short_hp_bytes(T, d_in, d_out, r_live, dtype);
  if (r_live <= 64) {
    const int cg_in = (d_in + 63) / 64, cg_out = (d_out + 63) / 64;
    const int quads = (dtype == SOW_F32 && tn_f32q_shape_ok(T, d_in, d_out)) ? (cg_in + 3) / 4 + (cg_out + 3) / 4 : 0;
    w.ns = tn_pick_slabs(T, cg_in + cg_out, (cg_in + 1) / 2 + (cg_out + 1) / 2, quads, dtype, &w.slab_len);
    w.ns_cap = w.ns;
    if (dtype == SOW_BF16 && T >= 1024) {
      const int by_len = (int)(T / 512 < TNR_MAX_SLABS ? T / 512 : TNR_MAX_SLABS);
      if (by_len > w.ns_cap) w.ns_cap = by_len;
    }
    w.off_p0 = off;
    off += al256(tn_partial_bytes(w.ns_cap, d_in));
    w.off_p1 = off;
    off += al256(tn_partial_bytes(w.ns_cap, d_out));
  }
  // fp32 streaming chain (chain3f.hip): the factors of a launch pre-split into bf16 planes, 24 KiB per 64-wide chunk of
  // d_in and of d_out (either direction; a low-rank accumulator's chain reuses the region: same stream, launch after launch)
  w.off_planes = off;
  w.planes_bytes = 0;
  if (dtype == SOW_F32 && T >= C3F_MIN_T && (r_live <= 64 || (acc_kind == SOW_ACC_LOWRANK && r_acc <= 64))) {
    w.planes_bytes = chain3f_plane_bytes(d_in, d_out);
    off += al256(w.planes_bytes);
  }
  // split-K scratch of the dense-accumulator products of short inputs (gemm4.hip): forward [T, d_out] over K = d_in (+ the
  // rank extension), backward [T, d_in] over K = d_out
  w.off_sk = off;
  w.sk_bytes = 0;
  if (dtype == SOW_BF16 && acc_kind == SOW_ACC_DENSE) {
    const size_t f = gemm4_splitk_bytes(T, d_out, d_in, true), b = gemm4_splitk_bytes(T, d_in, d_out, true);
    w.sk_bytes = f > b ? f : b;
    off += al256(w.sk_bytes);
  }
  w.total = off;
  return w;
}

static char* ws_base(void* workspace) {
  uintptr_t a = reinterpret_cast<uintptr_t>(workspace);
  return reinterpret_cast<char*>((a + 255) & ~(uintptr_t)255);
}
// hands the factor-plane scratch of the workspace to a chain launch (fp32, long T: chain3f.hip); without it the launch
// takes chain2f
static void set_planes(ChainParams& p, char* ws, const WsPlan& w, size_t workspace_bytes) {
  if (ws && w.planes_bytes && workspace_bytes >= w.total) p.planes = ws + w.off_planes, p.planes_bytes = w.planes_bytes;
}

// the reduction of the slab partials of one layer (shared by sow_backward_ex and the deferred, batched form)
static ReduceParams make_reduce_params(const WsPlan& w, char* ws, void* dA, void* dB, void* dbias_ones, int d_in, int d_out,
                                       int r_live, float grad_beta) {
  ReduceParams rp{};
  rp.njobs = 2, rp.ns = w.ns;
  rp.job[0] = ReduceJob{(const float*)(ws + w.off_p0), dA, nullptr, (int64_t)r_live, d_in, (d_in + 63) / 64 * 64, r_live, 0, -1,
                        1.f, grad_beta};
  rp.job[1] = ReduceJob{(const float*)(ws + w.off_p1), dB, dbias_ones, (int64_t)d_out, d_out, (d_out + 63) / 64 * 64, r_live, 1,
                        dbias_ones ? 63 : -1, 1.f /* h_save is already scaled */, grad_beta};
  rp.blocks0 = (d_in + 3) / 4;
  return rp;
}

size_t sow_reduce_desc_bytes(void) { return sizeof(ReduceParams); }

int sow_backward_reduce_desc(void* dA, void* dB, void* dbias, int64_t T, int d_in, int d_out, int r_live, int r_acc,
                             int acc_kind, float grad_beta, int dtype, void* workspace, size_t workspace_bytes, void* desc_out,
                             int* blocks_out) {
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (T <= 0 || d_in <= 0 || d_out <= 0 || r_live <= 0) return SOW_ERR_SHAPE;
  if (!dA || !dB || !workspace || !desc_out || !blocks_out) return SOW_ERR_NULL;
  if (r_live > 64 || (dbias && r_live > 63)) return SOW_ERR_UNSUPPORTED;   // those paths have no separate reduction
  if (acc_kind != SOW_ACC_LOWRANK) r_acc = 0;
  const WsPlan w = plan_ws(T, d_in, d_out, r_live, r_acc, acc_kind, dtype);
  if (workspace_bytes < w.total + 255) return SOW_ERR_WORKSPACE;
  const ReduceParams rp = make_reduce_params(w, ws_base(workspace), dA, dB, dbias, d_in, d_out, r_live, grad_beta);
  memcpy(desc_out, &rp, sizeof(rp));
  *blocks_out = (d_in + 3) / 4 + (d_out + 3) / 4;
  return SOW_OK;
}

int sow_reduce_batch(const void* descs, const int* starts, int n, int total_blocks, int dtype, void* stream) {
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (n < 0 || total_blocks < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!descs || !starts) return SOW_ERR_NULL;
  return launch_tn_reduce_batch((const ReduceParams*)descs, starts, n, total_blocks, dtype, (hipStream_t)stream);
}

size_t sow_forward_workspace_bytes(int64_t T, int d_in, int d_out, int r_live, int r_acc, int acc_kind, int dtype) {
  if (T < 0 || d_in <= 0 || d_out <= 0 || r_live <= 0 || !ok_dtype(dtype)) return 0;
  const bool wide_acc = acc_kind == SOW_ACC_LOWRANK && r_acc > 64;
  const WsPlan w = plan_ws(T, d_in, d_out, r_live, r_acc, acc_kind, dtype);
  if (!wide_acc && short_hp_bytes(T, d_in, d_out, r_live, dtype) == 0 && w.planes_bytes == 0 && w.sk_bytes == 0) return 0;   // the forward does not touch it
  return w.total + 256;
}

size_t sow_workspace_bytes(int64_t T, int d_in, int d_out, int r_live, int r_acc, int acc_kind, int dtype) {
  if (T < 0 || d_in <= 0 || d_out <= 0 || r_live <= 0 || !ok_dtype(dtype)) return 0;
  return plan_ws(T, d_in, d_out, r_live, r_acc, acc_kind, dtype).total + 256;
}


int sow_forward(const void* x, const void* A, const void* B, const void* acc_down, const void* acc_up, const void* bias,
                void* y, void* h_save, int64_t T, int d_in, int d_out, int r_live, int r_acc, int acc_kind, float scale,
                int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (T < 0 || d_in <= 0 || d_out <= 0 || r_live <= 0) return SOW_ERR_SHAPE;
  if (T == 0) return SOW_OK;
  if (!x || !A || !B || !y) return SOW_ERR_NULL;
  if (acc_kind != SOW_ACC_NONE && !acc_down) return SOW_ERR_NULL;
  if (acc_kind == SOW_ACC_LOWRANK && (!acc_up || r_acc <= 0)) return SOW_ERR_SHAPE;
  if (acc_kind != SOW_ACC_LOWRANK) r_acc = 0;
  const WsPlan w = plan_ws(T, d_in, d_out, r_live, r_acc, acc_kind, dtype);
  char* ws = workspace ? ws_base(workspace) : nullptr;
  float beta = 0.f;
  int rc;
  if (acc_kind == SOW_ACC_DENSE) {
    if (r_live <= 64 && h_save) {
      // one product with the low-rank term as a K-extension:  y = [x, h] . [W_acc; B] + bias,  h = scale * x . A
      // -- in one launch (h projected inside the GEMM) when d_out spans at most two column tiles
      // gemm4h: projection pass + anti-phase main loop in one launch, A read in place
      if (gemm4h_supported(x, d_in, acc_down, d_out, false, A, r_live, B, d_out, y, d_out, bias, h_save, T, d_out, d_in, r_live,
                           dtype))
        return launch_gemm4h(x, d_in, acc_down, d_out, false, A, r_live, B, d_out, y, d_out, bias, h_save, T, d_out, d_in,
                             r_live, scale, stream);
      if (gemm2h_supported(x, d_in, acc_down, d_out, false, A, r_live, B, d_out, y, d_out, bias, h_save, T, d_out, d_in,
                           r_live, dtype))
        return launch_gemm2h(x, d_in, acc_down, d_out, false, A, r_live, B, d_out, y, d_out, bias, h_save, T, d_out, d_in,
                             r_live, scale, stream);
      ChainParams ph{};
      ph.X = x, ph.Y = nullptr, ph.Hsave = h_save, ph.bias = nullptr;
      ph.M = T, ph.ldx = d_in, ph.ldy = d_out, ph.D1 = d_in, ph.D2 = 0;
      ph.F1b = A, ph.ldf1b = r_live, ph.F2b = B, ph.ldf2b = d_out, ph.rb = r_live;
      ph.scale = scale, ph.beta = 0.f;
      set_planes(ph, ws, w, workspace_bytes);
      if (chain2_supported(ph, dtype) &&
          gemm2_supported(x, d_in, acc_down, d_out, false, h_save, 64, B, d_out, y, d_out, bias, T, d_out, d_in, dtype)) {
        // short T: the H-only pass split over K (T / 64 workgroups cannot fill the chip: 29 us on 16 workgroups at 1024 x 4096)
        rc = SOW_ERR_UNSUPPORTED;
        if (ws && workspace_bytes >= w.total) rc = launch_chain_short(ph, dtype, false, (float*)(ws + w.off_hp), stream);
        if (rc == SOW_ERR_UNSUPPORTED) rc = launch_chain2(ph, false, stream);
        if (rc) return rc;
        const bool sk = ws && w.sk_bytes && workspace_bytes >= w.total;
        return launch_gemm2(x, d_in, acc_down, d_out, false, h_save, 64, B, d_out, r_live, y, d_out, bias, T, d_out, d_in,
                            1.f, 0.f, stream, sk ? ws + w.off_sk : nullptr, sk ? w.sk_bytes : 0);
      }
    }
    rc = gemm_auto(x, d_in, acc_down, d_out, false, y, d_out, nullptr, T, d_out, d_in, 1.f, 0.f, dtype, stream);
    if (rc) return rc;
    beta = 1.f;
  } else if (acc_kind == SOW_ACC_LOWRANK) {
    if (r_acc <= 64) {
      ChainParams p{};
      p.X = x, p.Y = y, p.Hsave = nullptr, p.bias = nullptr;
      p.M = T, p.ldx = d_in, p.ldy = d_out, p.D1 = d_in, p.D2 = d_out;
      p.F1b = acc_down, p.ldf1b = r_acc, p.F2b = acc_up, p.ldf2b = d_out, p.rb = r_acc;
      p.scale = 1.f, p.beta = 0.f;
      set_planes(p, ws, w, workspace_bytes);
      rc = launch_chain(p, dtype, false, stream);
      if (rc) return rc;
      beta = 1.f;
    } else {
      if (!ws || workspace_bytes < w.total) return SOW_ERR_WORKSPACE;
      void* t = ws + w.off_t;
      rc = launch_gemm(x, d_in, false, acc_down, r_acc, false, t, r_acc, nullptr, T, r_acc, d_in, 1.f, 0.f, dtype, stream);
      if (rc) return rc;
      rc = launch_gemm(t, r_acc, false, acc_up, d_out, false, y, d_out, nullptr, T, d_out, r_acc, 1.f, 0.f, dtype, stream);
      if (rc) return rc;
      beta = 1.f;
    }
  }
  if (r_live <= 64) {
    ChainParams p{};
    p.X = x, p.Y = y, p.Hsave = h_save, p.bias = bias;
    p.M = T, p.ldx = d_in, p.ldy = d_out, p.D1 = d_in, p.D2 = d_out;
    p.F1b = A, p.ldf1b = r_live, p.F2b = B, p.ldf2b = d_out, p.rb = r_live;
    p.scale = scale, p.beta = beta;
    set_planes(p, ws, w, workspace_bytes);
    if (ws && workspace_bytes >= w.total) {
      rc = launch_chain_short(p, dtype, false, (float*)(ws + w.off_hp), stream);
      if (rc != SOW_ERR_UNSUPPORTED) return rc;
    }
    return launch_chain(p, dtype, false, stream);
  }
  // generic rank: h = x A ; y = beta*y + scale * h B + bias
  if (!h_save) return SOW_ERR_NULL;
  rc = launch_gemm(x, d_in, false, A, r_live, false, h_save, r_live, nullptr, T, r_live, d_in, 1.f, 0.f, dtype, stream);
  if (rc) return rc;
  return launch_gemm(h_save, r_live, false, B, d_out, false, y, d_out, bias, T, d_out, r_live, scale, beta, dtype, stream);
}

int sow_backward_ex(const void* dy, const void* x, const void* h_save, const void* A, const void* B, const void* acc_down,
                    const void* acc_up, void* dx, void* dA, void* dB, void* dbias, int64_t T, int d_in, int d_out,
                    int r_live, int r_acc, int acc_kind, float scale, float grad_beta, int dtype, void* workspace,
                    size_t workspace_bytes, int phases, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bool do_data = (phases & SOW_BWD_DATA) != 0;
  const bool do_partial = (phases & (SOW_BWD_WEIGHTS | SOW_BWD_WEIGHTS_PARTIAL)) != 0;
  const bool do_reduce = (phases & (SOW_BWD_WEIGHTS | SOW_BWD_WEIGHTS_REDUCE)) != 0;
  const bool do_weights = do_partial || do_reduce;
  if (!do_data && !do_weights) return SOW_ERR_SHAPE;
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (T < 0 || d_in <= 0 || d_out <= 0 || r_live <= 0) return SOW_ERR_SHAPE;
  if (T == 0) {
    // empty batch: gradients are zero (or unchanged when accumulating); x / dy / dx may be NULL
    if (!dA || !dB) return SOW_ERR_NULL;
    if (grad_beta == 0.f) {
      void* ptrs[3] = {dA, dB, dbias};
      int64_t bytes[3] = {(int64_t)d_in * r_live * (int64_t)esize(dtype), (int64_t)r_live * d_out * (int64_t)esize(dtype),
                          dbias ? (int64_t)d_out * (int64_t)esize(dtype) : 0};
      return launch_multi_zero(ptrs, bytes, 3, stream);
    }
    return SOW_OK;
  }
  if (!dy || !x || !h_save || !A || !B || !dx || !dA || !dB || !workspace) return SOW_ERR_NULL;
  if (acc_kind != SOW_ACC_NONE && !acc_down) return SOW_ERR_NULL;
  if (acc_kind == SOW_ACC_LOWRANK && (!acc_up || r_acc <= 0)) return SOW_ERR_SHAPE;
  if (acc_kind != SOW_ACC_LOWRANK) r_acc = 0;
  const WsPlan w = plan_ws(T, d_in, d_out, r_live, r_acc, acc_kind, dtype);
  if (workspace_bytes < w.total + 255) return SOW_ERR_WORKSPACE;
  char* ws = ws_base(workspace);
  void* dh = ws + w.off_dh;
  float beta = 0.f;
  int rc;
  bool data_done = false;
  if (!do_data) {
    // weights-only call: dh was produced by an earlier SOW_BWD_DATA call on the same workspace
  } else if (acc_kind == SOW_ACC_DENSE) {
    if (r_live <= 64) {
      // one product with the low-rank term as a K-extension:  dX = [dY, dh] . [W_acc^T; A^T],  dh = scale * dY . B^T
      ChainParams pd{};
      pd.X = dy, pd.Y = nullptr, pd.Hsave = dh, pd.bias = nullptr;
      pd.M = T, pd.ldx = d_out, pd.ldy = d_in, pd.D1 = d_out, pd.D2 = 0;
      pd.F1b = B, pd.ldf1b = d_out, pd.F2b = A, pd.ldf2b = r_live, pd.rb = r_live;
      pd.scale = scale, pd.beta = 0.f;
      set_planes(pd, ws, w, workspace_bytes);
      void* apad = ws + w.off_apad;
      if (gemm4h_supported(dy, d_out, acc_down, d_out, true, B, d_out, A, r_live, dx, d_in, nullptr, dh, T, d_in, d_out, r_live,
                           dtype)) {
        // one launch: dh projected by the kernel's own pass over dY, A^T read from A's own 2r-byte rows
        rc = launch_gemm4h(dy, d_out, acc_down, d_out, true, B, d_out, A, r_live, dx, d_in, nullptr, dh, T, d_in, d_out, r_live,
                           scale, stream);
        if (rc) return rc;
        data_done = true;
      } else if (gemm2h_supported(dy, d_out, acc_down, d_out, true, B, d_out, A, r_live, dx, d_in, nullptr, dh, T, d_in, d_out,
                           r_live, dtype)) {
        // one launch: dh projected inside the GEMM, A^T read from A's own 2r-byte rows
        rc = launch_gemm2h(dy, d_out, acc_down, d_out, true, B, d_out, A, r_live, dx, d_in, nullptr, dh, T, d_in, d_out,
                           r_live, scale, stream);
        if (rc) return rc;
        data_done = true;
      } else if (chain2_supported(pd, dtype) &&
          gemm2_supported(dy, d_out, acc_down, d_out, true, dh, 64, apad, 64, dx, d_in, nullptr, T, d_in, d_out, dtype)) {
        // A zero-padded to 64 columns rides along in the dh launch when that grid is large enough
        const bool pad_fused = (int64_t)ceil_div(T, 64) * 64 >= d_in;
        if (pad_fused) pd.pad_src = A, pd.pad_dst = apad, pd.pad_rows = d_in, pd.pad_r = r_live;
        rc = SOW_ERR_UNSUPPORTED;
        if (short_hp_bytes(T, d_in, d_out, r_live, dtype)) rc = launch_chain_short(pd, dtype, true, (float*)(ws + w.off_hp), stream);
        if (rc == SOW_ERR_UNSUPPORTED) rc = launch_chain2(pd, true, stream);
        if (rc) return rc;
        if (!pad_fused) {
          rc = launch_pad64(A, apad, d_in, r_live, stream);
          if (rc) return rc;
        }
        rc = launch_gemm2(dy, d_out, acc_down, d_out, true, dh, 64, apad, 64, 64, dx, d_in, nullptr, T, d_in, d_out, 1.f, 0.f,
                          stream, w.sk_bytes ? ws + w.off_sk : nullptr, w.sk_bytes);
        if (rc) return rc;
        data_done = true;
      }
    }
    // dX = dY . W_acc^T   (W_acc stored [d_in, d_out] = [N, K])
    if (!data_done) rc = gemm_auto(dy, d_out, acc_down, d_out, true, dx, d_in, nullptr, T, d_in, d_out, 1.f, 0.f, dtype, stream);
    if (rc) return rc;
    beta = 1.f;
  } else if (acc_kind == SOW_ACC_LOWRANK) {
    if (r_acc <= 64) {
      // dX = (dY . R_up^T) . Q^T : the same chain kernel with the frozen factors, scale 1
      ChainParams p{};
      p.X = dy, p.Y = dx, p.Hsave = nullptr, p.bias = nullptr;
      p.M = T, p.ldx = d_out, p.ldy = d_in, p.D1 = d_out, p.D2 = d_in;
      p.F1b = acc_up, p.ldf1b = d_out, p.F2b = acc_down, p.ldf2b = r_acc, p.rb = r_acc;
      p.scale = 1.f, p.beta = 0.f;
      set_planes(p, ws, w, workspace_bytes);
      rc = launch_chain(p, dtype, true, stream);
      if (rc) return rc;
      beta = 1.f;
    } else {
      void* t = ws + w.off_t;
      rc = launch_gemm(dy, d_out, false, acc_up, d_out, true, t, r_acc, nullptr, T, r_acc, d_out, 1.f, 0.f, dtype, stream);
      if (rc) return rc;
      rc = launch_gemm(t, r_acc, false, acc_down, r_acc, true, dx, d_in, nullptr, T, d_in, r_acc, 1.f, 0.f, dtype, stream);
      if (rc) return rc;
      beta = 1.f;
    }
  }
  if (r_live <= 64) {
    ChainParams p{};
    p.X = dy, p.Y = dx, p.Hsave = dh, p.bias = nullptr;
    p.M = T, p.ldx = d_out, p.ldy = d_in, p.D1 = d_out, p.D2 = d_in;
    p.F1b = B, p.ldf1b = d_out, p.F2b = A, p.ldf2b = r_live, p.rb = r_live;
    p.scale = scale, p.beta = beta;
    set_planes(p, ws, w, workspace_bytes);
    if (do_data && !data_done) {
      rc = launch_chain_short(p, dtype, true, short_hp_bytes(T, d_in, d_out, r_live, dtype) ? (float*)(ws + w.off_hp) : nullptr, stream);
      if (rc == SOW_ERR_UNSUPPORTED) rc = launch_chain(p, dtype, true, stream);
      if (rc) return rc;
    }
    if (!do_weights) return SOW_OK;
    // weight gradients: dA = x^T dh ; dB^T = dY^T h ; dbias = colsum(dY) via the all-ones column 63
    const bool ones_ok = dbias && r_live <= 63;
    TnParams tp{};
    tp.njobs = 2, tp.T = T, tp.ns = w.ns, tp.slab_len = w.slab_len;
    auto vec_ok = [&](const void* ptr, int D) {
      if (dtype == SOW_F32) return (D % 4 == 0 && al16p(ptr)) ? 1 : 0;
      return (D % 2 == 0 && al4p(ptr)) ? 1 : 0;
    };
    // the chain kernels write 1.0 into column 63 of h_save / dh whenever r_live <= 63
    tp.job[0] = TnJob{x, dh, (float*)(ws + w.off_p0), (int64_t)d_in, d_in, -1, (d_in + 63) / 64, vec_ok(x, d_in), 1};
    tp.job[1] = TnJob{dy, h_save, (float*)(ws + w.off_p1), (int64_t)d_out, d_out, ones_ok ? 63 : -1, (d_out + 63) / 64,
                      vec_ok(dy, d_out), 1};
    if (do_partial) {
      rc = launch_tn(tp, dtype, stream);
      if (rc) return rc;
    }
    if (!do_reduce) return SOW_OK;
    const ReduceParams rp = make_reduce_params(w, ws, dA, dB, ones_ok ? dbias : nullptr, d_in, d_out, r_live, grad_beta);
    rc = launch_tn_reduce(rp, dtype, stream);
    if (rc) return rc;
    if (dbias && !ones_ok) {
      if (dtype == SOW_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, dim3((d_out + 63) / 64), dim3(256), 0, stream, (const float*)dy, (int64_t)d_out, T, d_out, (float*)dbias, grad_beta);
      else
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((d_out + 63) / 64), dim3(256), 0, stream, (const bf16_t*)dy, (int64_t)d_out, T, d_out, (bf16_t*)dbias, grad_beta);
      SOW_CHECK_LAUNCH();
    }
    return SOW_OK;
  }
  // generic rank (GEMM composition)
  if (do_data) {
    rc = launch_gemm(dy, d_out, false, B, d_out, true, dh, r_live, nullptr, T, r_live, d_out, scale, 0.f, dtype, stream);
    if (rc) return rc;
    rc = launch_gemm(dh, r_live, false, A, r_live, true, dx, d_in, nullptr, T, d_in, r_live, 1.f, beta, dtype, stream);
    if (rc) return rc;
  }
  if (!do_partial) return SOW_OK;   // wide ranks: the PARTIAL phase does all of the weight gradients
  rc = launch_gemm(x, d_in, true, dh, r_live, false, dA, r_live, nullptr, d_in, r_live, (int)T, 1.f, grad_beta, dtype, stream);
  if (rc) return rc;
  rc = launch_gemm(h_save, r_live, true, dy, d_out, false, dB, d_out, nullptr, r_live, d_out, (int)T, scale, grad_beta, dtype, stream);
  if (rc) return rc;
  if (dbias) {
    if (dtype == SOW_F32)
      hipLaunchKernelGGL(colsum_kernel<float>, dim3((d_out + 63) / 64), dim3(256), 0, stream, (const float*)dy, (int64_t)d_out, T, d_out, (float*)dbias, grad_beta);
    else
      hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((d_out + 63) / 64), dim3(256), 0, stream, (const bf16_t*)dy, (int64_t)d_out, T, d_out, (bf16_t*)dbias, grad_beta);
    SOW_CHECK_LAUNCH();
  }
  return SOW_OK;
}

int sow_backward(const void* dy, const void* x, const void* h_save, const void* A, const void* B, const void* acc_down,
                 const void* acc_up, void* dx, void* dA, void* dB, void* dbias, int64_t T, int d_in, int d_out,
                 int r_live, int r_acc, int acc_kind, float scale, float grad_beta, int dtype, void* workspace,
                 size_t workspace_bytes, void* stream) {
  return sow_backward_ex(dy, x, h_save, A, B, acc_down, acc_up, dx, dA, dB, dbias, T, d_in, d_out, r_live, r_acc, acc_kind,
                         scale, grad_beta, dtype, workspace, workspace_bytes, SOW_BWD_DATA | SOW_BWD_WEIGHTS, stream);
}

// ---- grouped entry points ------------------------------------------------------------------------------------
// A group is n INDEPENDENT SoWLinear calls (e.g. q / k / v of one attention block, gate / up of one MLP).  Layers that
// take the plain bf16 streaming kernels (no accumulator, r <= 64, T / 64 > SHORT_NTB) share launches, C2_MAXG /
// TN_MAXG at a time; every other layer is forwarded to the single-layer entry point.  Results are bit-identical to n
// separate calls: the shared grid runs each layer's own workgroups unchanged.
static bool group_chain_params(const sow_layer_args& L, bool bwd, int dtype, const WsPlan& w, ChainParams* out) {
  if (dtype != SOW_BF16 || L.acc_kind != SOW_ACC_NONE || L.r_live > 64 || sw_on(SW_NO_GROUPED) || sw_on(SW_FORCE_CHAIN_V1))
    return false;
  if (ceil_div(L.T, 64) <= SHORT_NTB) return false;   // short inputs: K / column split, single-layer path
  ChainParams p{};
  if (!bwd) {
    p.X = L.x, p.Y = L.y, p.Hsave = L.h_save, p.bias = L.bias;
    p.M = L.T, p.ldx = L.d_in, p.ldy = L.d_out, p.D1 = L.d_in, p.D2 = L.d_out;
    p.F1b = L.A, p.ldf1b = L.r_live, p.F2b = L.B, p.ldf2b = L.d_out, p.rb = L.r_live;
  } else {
    p.X = L.dy, p.Y = L.dx, p.Hsave = ws_base(L.workspace) + w.off_dh, p.bias = nullptr;
    p.M = L.T, p.ldx = L.d_out, p.ldy = L.d_in, p.D1 = L.d_out, p.D2 = L.d_in;
    p.F1b = L.B, p.ldf1b = L.d_out, p.F2b = L.A, p.ldf2b = L.r_live, p.rb = L.r_live;
  }
  p.scale = L.scale, p.beta = 0.f;
  if (!chain2_supported(p, dtype)) return false;
  // the alignment conditions launch_chain2 would reject (it then falls back inside launch_chain)
  const void* Bp = bwd ? p.F1b : p.F2b;
  const int64_t ldB = bwd ? p.ldf1b : p.ldf2b;
  const void* Ap = bwd ? p.F2b : p.F1b;
  if ((reinterpret_cast<uintptr_t>(Bp) & 15) || ldB % 8 || (reinterpret_cast<uintptr_t>(Ap) & 3)) return false;
  *out = p;
  return true;
}

// Group-planned slab counts for the weight-gradient partial sums (skinny_tn.hip: tn_partial_rows_kernel).  A pure function
// of the layer list, so that sow_backward_group and sow_backward_group_reduce_desc agree.  Items 2 i, 2 i + 1 = the two
// operands (x with dh, dY with h) of layer i.
static bool group_rows_plan(const sow_layer_args* layers, int n, int dtype, int* ns, int* slab_len) {
  if (dtype != SOW_BF16 || n < 1 || n > TN_MAXG) return false;
  int64_t T[TNR_MAXI];
  int D[TNR_MAXI], cap[TNR_MAXI];
  for (int i = 0; i < n; ++i) {
    const sow_layer_args& L = layers[i];
    if (L.T <= 0 || L.r_live > 64 || (L.dbias && L.r_live > 63)) return false;
    const WsPlan w = plan_ws(L.T, L.d_in, L.d_out, L.r_live, L.acc_kind == SOW_ACC_LOWRANK ? L.r_acc : 0, L.acc_kind, dtype);
    if (L.workspace_bytes < w.total + 255) return false;
    if (L.d_in % 8 || L.d_out % 8 || !al16p(L.x) || !al16p(L.dy) || !al16p(L.h_save)) return false;
    T[2 * i] = T[2 * i + 1] = L.T;
    D[2 * i] = L.d_in, D[2 * i + 1] = L.d_out;
    cap[2 * i] = cap[2 * i + 1] = w.ns_cap;
  }
  return tn_rows_plan(T, D, cap, 2 * n, ns, slab_len);
}

static int check_layer(const sow_layer_args& L, bool bwd) {
  if (L.T < 0 || L.d_in <= 0 || L.d_out <= 0 || L.r_live <= 0) return SOW_ERR_SHAPE;
  if (L.T == 0) return SOW_OK;
  if (!L.x || !L.A || !L.B) return SOW_ERR_NULL;
  if (!bwd && !L.y) return SOW_ERR_NULL;
  if (bwd && (!L.dy || !L.h_save || !L.dx || !L.dA || !L.dB || !L.workspace)) return SOW_ERR_NULL;
  if (L.acc_kind != SOW_ACC_NONE && !L.acc_down) return SOW_ERR_NULL;
  return SOW_OK;
}

int sow_forward_group(const sow_layer_args* layers, int n, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!layers) return SOW_ERR_NULL;
  ChainParams batch[C2_MAXG];
  Gemm2hArgs gbatch[4];
  int nb = 0, ng = 0, rc;
  for (int i = 0; i < n; ++i) {
    const sow_layer_args& L = layers[i];
    if ((rc = check_layer(L, false))) return rc;
    if (L.T == 0) continue;
    const WsPlan w{};
    if (group_chain_params(L, false, dtype, w, &batch[nb])) {   // h_save may be NULL (no backward follows)
      if (++nb == C2_MAXG) {
        if ((rc = launch_chain2_group(batch, nb, false, stream))) return rc;
        nb = 0;
      }
      continue;
    }
    // dense accumulator: gemm4h (one launch per layer) where it applies
    if (L.acc_kind == SOW_ACC_DENSE && L.r_live <= 64 && L.h_save &&
        gemm4h_supported(L.x, L.d_in, L.acc_down, L.d_out, false, L.A, L.r_live, L.B, L.d_out, L.y, L.d_out, L.bias, L.h_save, L.T,
                         L.d_out, L.d_in, L.r_live, dtype)) {
      if ((rc = launch_gemm4h(L.x, L.d_in, L.acc_down, L.d_out, false, L.A, L.r_live, L.B, L.d_out, L.y, L.d_out, L.bias, L.h_save,
                              L.T, L.d_out, L.d_in, L.r_live, L.scale, stream)))
        return rc;
      continue;
    }
    // dense accumulator, one launch per layer (gemm2h): the layers of a group share the grid
    if (L.acc_kind == SOW_ACC_DENSE && L.r_live <= 64 && L.h_save && !sw_on(SW_NO_GROUPED) &&
        gemm2h_supported(L.x, L.d_in, L.acc_down, L.d_out, false, L.A, L.r_live, L.B, L.d_out, L.y, L.d_out, L.bias, L.h_save, L.T,
                         L.d_out, L.d_in, L.r_live, dtype)) {
      gbatch[ng] = Gemm2hArgs{L.x, L.acc_down, L.A, L.B, L.y, L.bias, L.h_save, L.T, L.d_in, L.d_out, L.r_live, L.d_out, L.d_out,
                              L.d_out, L.d_in, L.r_live, L.scale};
      if (++ng == 4) {
        if ((rc = launch_gemm2h_group(gbatch, ng, false, stream))) return rc;
        ng = 0;
      }
      continue;
    }
    rc = sow_forward(L.x, L.A, L.B, L.acc_down, L.acc_up, L.bias, L.y, L.h_save, L.T, L.d_in, L.d_out, L.r_live, L.r_acc,
                     L.acc_kind, L.scale, dtype, L.workspace, L.workspace_bytes, stream_);
    if (rc) return rc;
  }
  if (ng && (rc = launch_gemm2h_group(gbatch, ng, false, stream))) return rc;
  return nb ? launch_chain2_group(batch, nb, false, stream) : SOW_OK;
}

int sow_backward_group(const sow_layer_args* layers, int n, int dtype, int phases, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bool do_data = (phases & SOW_BWD_DATA) != 0;
  const bool do_partial = (phases & (SOW_BWD_WEIGHTS | SOW_BWD_WEIGHTS_PARTIAL)) != 0;
  const bool do_reduce = (phases & (SOW_BWD_WEIGHTS | SOW_BWD_WEIGHTS_REDUCE)) != 0;
  if (!do_data && !do_partial && !do_reduce) return SOW_ERR_SHAPE;
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!layers) return SOW_ERR_NULL;
  int rc;
  for (int i = 0; i < n; ++i)
    if ((rc = check_layer(layers[i], true))) return rc;
  auto single = [&](const sow_layer_args& L, int ph) {
    return sow_backward_ex(L.dy, L.x, L.h_save, L.A, L.B, L.acc_down, L.acc_up, L.dx, L.dA, L.dB, L.dbias, L.T, L.d_in, L.d_out,
                           L.r_live, L.r_acc, L.acc_kind, L.scale, L.grad_beta, dtype, L.workspace, L.workspace_bytes, ph,
                           stream_);
  };
  auto plan = [&](const sow_layer_args& L) {
    return plan_ws(L.T, L.d_in, L.d_out, L.r_live, L.acc_kind == SOW_ACC_LOWRANK ? L.r_acc : 0, L.acc_kind, dtype);
  };
  if (do_data) {
    ChainParams batch[C2_MAXG];
    Gemm2hArgs gbatch[4];
    int nb = 0, ng = 0;
    for (int i = 0; i < n; ++i) {
      const sow_layer_args& L = layers[i];
      if (L.T == 0) {
        if ((rc = single(L, SOW_BWD_DATA | (phases & ~SOW_BWD_DATA)))) return rc;   // empty batch: everything at once
        continue;
      }
      const WsPlan w = plan(L);
      if (L.workspace_bytes < w.total + 255) return SOW_ERR_WORKSPACE;
      void* dh = ws_base(L.workspace) + w.off_dh;
      if (group_chain_params(L, true, dtype, w, &batch[nb])) {
        if (++nb == C2_MAXG) {
          if ((rc = launch_chain2_group(batch, nb, true, stream))) return rc;
          nb = 0;
        }
      } else if (L.acc_kind == SOW_ACC_DENSE && L.r_live <= 64 &&
                 gemm4h_supported(L.dy, L.d_out, L.acc_down, L.d_out, true, L.B, L.d_out, L.A, L.r_live, L.dx, L.d_in, nullptr, dh,
                                  L.T, L.d_in, L.d_out, L.r_live, dtype)) {
        if ((rc = launch_gemm4h(L.dy, L.d_out, L.acc_down, L.d_out, true, L.B, L.d_out, L.A, L.r_live, L.dx, L.d_in, nullptr, dh,
                                L.T, L.d_in, L.d_out, L.r_live, L.scale, stream)))
          return rc;
      } else if (L.acc_kind == SOW_ACC_DENSE && L.r_live <= 64 && !sw_on(SW_NO_GROUPED) &&
                 gemm2h_supported(L.dy, L.d_out, L.acc_down, L.d_out, true, L.B, L.d_out, L.A, L.r_live, L.dx, L.d_in, nullptr, dh,
                                  L.T, L.d_in, L.d_out, L.r_live, dtype)) {
        gbatch[ng] = Gemm2hArgs{L.dy, L.acc_down, L.B, L.A, L.dx, nullptr, dh, L.T, L.d_out, L.d_out, L.d_out, L.r_live, L.d_in,
                                L.d_in, L.d_out, L.r_live, L.scale};
        if (++ng == 4) {
          if ((rc = launch_gemm2h_group(gbatch, ng, true, stream))) return rc;
          ng = 0;
        }
      } else if ((rc = single(L, SOW_BWD_DATA)))
        return rc;
    }
    if (ng && (rc = launch_gemm2h_group(gbatch, ng, true, stream))) return rc;
    if (nb && (rc = launch_chain2_group(batch, nb, true, stream))) return rc;
  }
  // row-owner weight-gradient kernel with slab counts planned over the group: when the caller asks for it (and then builds
  // the deferred reduction from sow_backward_group_reduce_desc), or when this call runs the reduction itself
  int rns[TNR_MAXI], rslab[TNR_MAXI];
  const bool rows = (do_partial || do_reduce) && ((phases & SOW_BWD_GROUP_SLABS) || (do_partial && do_reduce)) &&
                    group_rows_plan(layers, n, dtype, rns, rslab);
  if (do_partial && rows) {
    TnRowsItem items[TNR_MAXI];
    for (int i = 0; i < n; ++i) {
      const sow_layer_args& L = layers[i];
      const WsPlan w = plan(L);
      char* ws = ws_base(L.workspace);
      items[2 * i] = TnRowsItem{L.x, ws + w.off_dh, (float*)(ws + w.off_p0), (int64_t)L.d_in, L.T, L.d_in, 0, 0, 0, 0, rns[2 * i],
                                rslab[2 * i], 0};
      items[2 * i + 1] = TnRowsItem{L.dy, L.h_save, (float*)(ws + w.off_p1), (int64_t)L.d_out, L.T, L.d_out, 0, 0, 0, 0,
                                    rns[2 * i + 1], rslab[2 * i + 1], 0};
    }
    if ((rc = launch_tn_rows(items, 2 * n, stream))) return rc;
  } else if (do_partial) {
    TnParams batch[TN_MAXG];
    int nb = 0;
    for (int i = 0; i < n; ++i) {
      const sow_layer_args& L = layers[i];
      if (L.T == 0) {
        if (!do_data && (rc = single(L, phases))) return rc;
        continue;
      }
      const WsPlan w = plan(L);
      if (L.workspace_bytes < w.total + 255) return SOW_ERR_WORKSPACE;
      bool grouped = false;
      if (L.r_live <= 64 && !sw_on(SW_NO_GROUPED)) {
        char* ws = ws_base(L.workspace);
        TnParams tp{};
        tp.njobs = 2, tp.T = L.T, tp.ns = w.ns, tp.slab_len = w.slab_len;
        const bool ones_ok = L.dbias && L.r_live <= 63;
        const int vx = (L.d_in % 2 == 0 && al4p(L.x)) ? 1 : 0, vy = (L.d_out % 2 == 0 && al4p(L.dy)) ? 1 : 0;
        tp.job[0] = TnJob{L.x, ws + w.off_dh, (float*)(ws + w.off_p0), (int64_t)L.d_in, L.d_in, -1, (L.d_in + 63) / 64, vx, 1};
        tp.job[1] = TnJob{L.dy, L.h_save, (float*)(ws + w.off_p1), (int64_t)L.d_out, L.d_out, ones_ok ? 63 : -1,
                          (L.d_out + 63) / 64, vy, 1};
        if (tn_group_supported(tp, dtype)) {
          batch[nb] = tp;
          grouped = true;
          if (++nb == TN_MAXG) {
            if ((rc = launch_tn_group(batch, nb, stream))) return rc;
            nb = 0;
          }
        }
      }
      if (!grouped && (rc = single(L, SOW_BWD_WEIGHTS_PARTIAL))) return rc;
    }
    if (nb && (rc = launch_tn_group(batch, nb, stream))) return rc;
  }
  if (do_reduce && rows) {
    for (int i = 0; i < n; ++i) {
      const sow_layer_args& L = layers[i];
      ReduceParams rp = make_reduce_params(plan(L), ws_base(L.workspace), L.dA, L.dB, L.dbias, L.d_in, L.d_out, L.r_live, L.grad_beta);
      rp.job[0].ns = rns[2 * i], rp.job[1].ns = rns[2 * i + 1];
      if ((rc = launch_tn_reduce(rp, dtype, stream))) return rc;
    }
  } else if (do_reduce) {
    for (int i = 0; i < n; ++i)
      if (layers[i].T != 0 && (rc = single(layers[i], SOW_BWD_WEIGHTS_REDUCE))) return rc;
  }
  return SOW_OK;
}

int sow_backward_group_plan(const sow_layer_args* layers, int n, int dtype, int phases, int* slabs_out) {
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return 0;
  if (!layers) return SOW_ERR_NULL;
  int rc;
  for (int i = 0; i < n; ++i)
    if ((rc = check_layer(layers[i], true))) return rc;
  const bool do_partial = (phases & (SOW_BWD_WEIGHTS | SOW_BWD_WEIGHTS_PARTIAL)) != 0;
  const bool do_reduce = (phases & (SOW_BWD_WEIGHTS | SOW_BWD_WEIGHTS_REDUCE)) != 0;
  int rns[TNR_MAXI], rslab[TNR_MAXI];
  const bool rows = (do_partial || do_reduce) && ((phases & SOW_BWD_GROUP_SLABS) || (do_partial && do_reduce)) &&
                    group_rows_plan(layers, n, dtype, rns, rslab);
  if (slabs_out)
    for (int i = 0; i < n; ++i) {
      const sow_layer_args& L = layers[i];
      int ns = 0;
      if (L.T > 0 && L.r_live <= 64)
        ns = plan_ws(L.T, L.d_in, L.d_out, L.r_live, L.acc_kind == SOW_ACC_LOWRANK ? L.r_acc : 0, L.acc_kind, dtype).ns;
      slabs_out[2 * i] = rows ? rns[2 * i] : ns;
      slabs_out[2 * i + 1] = rows ? rns[2 * i + 1] : ns;
    }
  return rows ? 1 : 0;
}

int sow_backward_group_reduce_desc(const sow_layer_args* layers, int n, int dtype, int phases, void* descs_out, int* blocks_out) {
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!layers || !descs_out || !blocks_out) return SOW_ERR_NULL;
  int rc;
  for (int i = 0; i < n; ++i)
    if ((rc = check_layer(layers[i], true))) return rc;
  int rns[TNR_MAXI], rslab[TNR_MAXI];
  const bool rows = (phases & SOW_BWD_GROUP_SLABS) && group_rows_plan(layers, n, dtype, rns, rslab);
  for (int i = 0; i < n; ++i) {
    const sow_layer_args& L = layers[i];
    char* out = (char*)descs_out + (size_t)i * sizeof(ReduceParams);
    if ((rc = sow_backward_reduce_desc(L.dA, L.dB, L.dbias, L.T, L.d_in, L.d_out, L.r_live, L.r_acc, L.acc_kind, L.grad_beta, dtype,
                                       L.workspace, L.workspace_bytes, out, blocks_out + i)))
      return rc;
    if (rows) {
      ReduceParams rp;
      memcpy(&rp, out, sizeof rp);
      rp.job[0].ns = rns[2 * i], rp.job[1].ns = rns[2 * i + 1];
      memcpy(out, &rp, sizeof rp);
    }
  }
  return SOW_OK;
}

int sow_gemm(const void* A, int64_t lda, int trans_a, const void* B, int64_t ldb, int trans_b, void* C, int64_t ldc,
             const void* bias, int64_t M, int N, int K, float alpha, float beta, int dtype, void* stream) {
  return sow_gemm_ex(A, lda, trans_a, B, ldb, trans_b, C, ldc, bias, M, N, K, alpha, beta, dtype, nullptr, 0, stream);
}

size_t sow_gemm_workspace_bytes(int64_t M, int N, int K, int trans_a, int dtype) {
  if (trans_a || dtype != SOW_BF16 || M <= 0 || N <= 0 || K <= 0) return 0;
  return gemm4_splitk_bytes(M, N, K, false);
}

int sow_gemm_ex(const void* A, int64_t lda, int trans_a, const void* B, int64_t ldb, int trans_b, void* C, int64_t ldc,
                const void* bias, int64_t M, int N, int K, float alpha, float beta, int dtype, void* workspace,
                size_t workspace_bytes, void* stream) {
  if (!trans_a)
    return gemm_auto(A, lda, B, ldb, trans_b != 0, C, ldc, bias, M, N, K, alpha, beta, dtype, (hipStream_t)stream, workspace,
                     workspace_bytes);
  return launch_gemm(A, lda, trans_a != 0, B, ldb, trans_b != 0, C, ldc, bias, M, N, K, alpha, beta, dtype, (hipStream_t)stream);
}

struct QrPlan {
  int kc;
  size_t off_pt, off_qt, off_w, off_r, total;
};
static QrPlan plan_qr(int m, int n, int k, int in_dtype, int need_r, int out_dtype_is_f32) {
  QrPlan q{};
  q.kc = k < m ? k : m;
  if (n < q.kc) q.kc = n;
  size_t off = 0;
  q.off_pt = off;
  off += al256((size_t)q.kc * m * 4);
  q.off_qt = off;
  off += al256((size_t)k * m * 4);
  q.off_w = off;
  if (need_r && n > q.kc && in_dtype != SOW_F32) off += al256((size_t)m * (n - q.kc) * 4);
  q.off_r = off;
  if (need_r && n > q.kc && !out_dtype_is_f32) off += al256((size_t)k * (n - q.kc) * 4);
  q.total = off;
  return q;
}

size_t sow_qr_workspace_bytes(int m, int n, int k, int in_dtype, int need_r) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  return plan_qr(m, n, k, in_dtype, need_r, 0).total + 256;
}

int sow_qr_thin(const void* W, int64_t ldw, int m, int n, int in_dtype, int k, void* Q_out, int64_t ldq, void* R_out,
                int64_t ldr, int out_dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ok_dtype(in_dtype) || !ok_dtype(out_dtype)) return SOW_ERR_DTYPE;
  if (m <= 0 || n <= 0 || k <= 0 || k > m) return SOW_ERR_SHAPE;
  if (!W || !Q_out || !workspace) return SOW_ERR_NULL;
  const QrPlan q = plan_qr(m, n, k, in_dtype, R_out != nullptr, out_dtype == SOW_F32);
  if (workspace_bytes < q.total + 255) return SOW_ERR_WORKSPACE;
  char* ws = ws_base(workspace);
  float* Pt = (float*)(ws + q.off_pt);
  float* Qt = (float*)(ws + q.off_qt);
  int rc = launch_qr_panel(W, ldw, in_dtype, m, q.kc, k, Pt, Qt, stream);
  if (rc) return rc;
  rc = launch_qr_copy_out(Qt, Pt, Q_out, ldq, R_out, ldr, out_dtype, m, q.kc, k, k, stream);
  if (rc) return rc;
  if (R_out && n > q.kc) {
    // R[:k, kc:] = Q[:, :k]^T W[:, kc:]   (fp32 GEMM: A = Qt stored [k, m], B = W columns kc.., k-major)
    const int nt = n - q.kc;
    const void* Bp;
    int64_t ldb;
    if (in_dtype == SOW_F32) {
      Bp = (const float*)W + q.kc, ldb = ldw;
    } else {
      float* wf = (float*)(ws + q.off_w);
      rc = launch_cast_copy((const bf16_t*)W + q.kc, ldw, SOW_BF16, wf, nt, SOW_F32, m, nt, stream);
      if (rc) return rc;
      Bp = wf, ldb = nt;
    }
    if (out_dtype == SOW_F32) {
      rc = launch_gemm(Qt, m, false, Bp, ldb, false, (float*)R_out + q.kc, ldr, nullptr, k, nt, m, 1.f, 0.f, SOW_F32, stream);
      if (rc) return rc;
    } else {
      float* rt = (float*)(ws + q.off_r);
      rc = launch_gemm(Qt, m, false, Bp, ldb, false, rt, nt, nullptr, k, nt, m, 1.f, 0.f, SOW_F32, stream);
      if (rc) return rc;
      rc = launch_cast_copy(rt, nt, SOW_F32, (bf16_t*)R_out + q.kc, ldr, SOW_BF16, k, nt, stream);
      if (rc) return rc;
    }
  }
  return SOW_OK;
}

int sow_accumulate_batch(const sow_accumulate_args* items, int n, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ok_dtype(dtype)) return SOW_ERR_DTYPE;
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!items) return SOW_ERR_NULL;
  std::vector<AccItem> its((size_t)n);
  std::vector<void*> zp;
  std::vector<int64_t> zb;
  for (int i = 0; i < n; ++i) {
    const sow_accumulate_args& a = items[i];
    if (a.d_in <= 0 || a.d_out <= 0 || a.r <= 0 || a.r > 64) return SOW_ERR_SHAPE;
    if (!a.acc || !a.A || !a.B) return SOW_ERR_NULL;
    AccItem& t = its[(size_t)i];
    t.acc = a.acc, t.A = a.A, t.B = a.B, t.draw = a.draw, t.A_new = a.A_new, t.ld_draw = a.ld_draw;
    t.d_in = a.d_in, t.d_out = a.d_out, t.r = a.r, t.r_new = a.r_new, t.scale = a.scale, t.beta = a.acc_beta;
    t.Pt = t.Qt = nullptr, t.kc = 0;
    if (a.draw) {
      if (!a.A_new || !a.workspace || a.r_new <= 0 || a.r_new > a.d_in || a.draw_cols <= 0 || a.ld_draw < a.draw_cols)
        return a.A_new && a.workspace ? SOW_ERR_SHAPE : SOW_ERR_NULL;
      const QrPlan q = plan_qr(a.d_in, a.draw_cols, a.r_new, dtype, 0, 0);
      if (a.workspace_bytes < q.total + 255) return SOW_ERR_WORKSPACE;
      char* ws = ws_base(a.workspace);
      t.kc = q.kc, t.Pt = (float*)(ws + q.off_pt), t.Qt = (float*)(ws + q.off_qt);
    }
    if (a.zero && a.zero_bytes > 0) zp.push_back(a.zero), zb.push_back(a.zero_bytes);
  }
  int rc = launch_accumulate_batch(its.data(), n, dtype, stream);
  if (rc) return rc;
  return zp.empty() ? SOW_OK : launch_multi_zero(zp.data(), zb.data(), (int)zp.size(), stream);
}

int sow_zero_state(void* const* ptrs, const int64_t* bytes, int n, void* stream) {
  if (n < 0) return SOW_ERR_SHAPE;
  if (n == 0) return SOW_OK;
  if (!ptrs || !bytes) return SOW_ERR_NULL;
  return launch_multi_zero(ptrs, bytes, n, (hipStream_t)stream);
}

int sow_adamw_flat(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, int dtype, int state_dtype,
                   void* stream) {
  if (step < 1) return SOW_ERR_SHAPE;
  return launch_adamw_flat(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                           dtype, state_dtype, (hipStream_t)stream);
}

int sow_ttadam_dense(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                     float beta2, float eps, float step_size, float lr_times_wd, int clamp_v, void* stream) {
  return launch_ttadam_dense(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size, lr_times_wd, clamp_v,
                             (hipStream_t)stream);
}

int sow_tt_kron_core(const float* A, const float* B, float* out, int ra0, int rb0, int ij, int ra1, int rb1,
                     void* stream) {
  return launch_tt_kron_core(A, B, out, ra0, rb0, ij, ra1, rb1, (hipStream_t)stream);
}

int sow_absmax(const float* x, int64_t n, float* out, void* stream) {
  return launch_absmax(x, n, out, (hipStream_t)stream);
}

int sow_small_inverse(const float* A, float* out, int batch, int r, void* stream) {
  return launch_small_inverse(A, out, batch, r, (hipStream_t)stream);
}

int sow_axpby(const void* x, void* y, int64_t n, float a, float b, int dtype, void* stream) {
  return launch_axpby(x, y, n, a, b, dtype, (hipStream_t)stream);
}

int sow_cast_copy(const void* src, int64_t lds, int src_dtype, void* dst, int64_t ldd, int dst_dtype, int64_t rows,
                  int cols, void* stream) {
  if (!src || !dst) return SOW_ERR_NULL;
  return launch_cast_copy(src, lds, src_dtype, dst, ldd, dst_dtype, rows, cols, (hipStream_t)stream);
}

}  // extern "C"
