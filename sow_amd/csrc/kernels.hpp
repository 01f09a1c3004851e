// Parameter blocks and host launchers shared between the kernel translation units and api.hip.
#pragma once
#include "common.hpp"

namespace sow {
// chain.hip
struct ChainParams {
  const void* X;
  void* Y;
  const void *F1a, *F1b, *F2a, *F2b;
  void* Hsave;
  const void* bias;
  int64_t M, ldx, ldy, ldf1a, ldf1b, ldf2a, ldf2b;
  int D1, D2, ra, rb;
  float scale, beta;
  int fast_factors;
  // side job of the bf16 streaming kernel: out[rows, 64] = [in[rows, r] | 0] (the zero-padded A that the dense
  // backward's K-extension DMAs row by row); workgroup b copies rows 64 b .. 64 b + 63.  nullptr = none.
  const void* pad_src;
  void* pad_dst;
  int pad_rows, pad_r;
  // Short-T split of the streaming kernels (chain2 / chain2f) (grid = ntb token blocks x splits; ntb = 0: no split):
  //   st_per > 0: workgroup (tb, s) runs phase 1 over stages [s*st_per, ..) only and writes its fp32 partial H to
  //               Hpartial[s][M][64] (no phase 2; h_reduce sums, scales, masks and casts into Hsave);
  //   sl_per > 0: workgroup (tb, s) skips phase 1, takes H from Hload [M, 64] and runs phase 2 over slices
  //               [s*sl_per, ..) only.
  int ntb, st_per, sl_per;
  float* Hpartial;
  const void* Hload;
  int nt_store;   // chain2: write Y with the non-temporal hint
  int nt_load;    // chain2 (experiment): stream X with the non-temporal hint
  int pair_flush; // chain2: store two output slices at a time (256-byte pieces per row)
  // chain3f (fp32, T >= 8192): scratch for the pre-split factor planes (chain3f_plane_bytes(D1, D2), 16-byte aligned,
  // from the caller's workspace); nullptr = not available, the launch falls back to chain2f
  void* planes;
  size_t planes_bytes;
};
int launch_chain(ChainParams p, int dtype, bool bwd, hipStream_t stream);
// chain2.hip (bf16 streaming version)
constexpr int C2_MAXG = 4;   // layers per grouped launch
struct ChainGroup {
  ChainParams p[C2_MAXG];
  int start[C2_MAXG + 1];    // first workgroup of layer i; start[n..] = grid size
  int n;
  uint64_t* stamps;          // debug builds (SOW_STAMPS): [blocks][8] timeline, else nullptr
};
extern void* g_chain2_stamps;
bool chain2_supported(const ChainParams& p, int dtype);
int launch_chain2(const ChainParams& p, bool bwd, hipStream_t stream);
// one grid for n <= C2_MAXG independent layers of the same direction (each one chain2_supported)
int launch_chain2_group(const ChainParams* ps, int n, bool bwd, hipStream_t stream);
int launch_h_reduce(const float* Hpartial, int nsplit, void* Hsave, int64_t M, int rb, float scale, int dtype,
                    hipStream_t stream);
// chain2f.hip (fp32 streaming version)
bool chain2f_supported(const ChainParams& p, int dtype);
int launch_chain2f(const ChainParams& p, bool bwd, hipStream_t stream);
// chain3f.hip (fp32 streaming version with pre-split factor planes, 128-token workgroups)
size_t chain3f_plane_bytes(int d_in, int d_out);
bool chain3f_supported(const ChainParams& p, int dtype);
int launch_chain3f(const ChainParams& p, bool bwd, hipStream_t stream);
// skinny_tn.hip
struct TnJob {
  const void* M;
  const void* S;
  float* partial;
  int64_t ldm;
  int D;
  int ones_col;
  int ncg;
  int vec;
  int ones_col_in_s;  // S already carries 1.0 in column `ones_col` (or ones_col < 0): the DMA path cannot patch it
};
struct TnParams {
  TnJob job[2];
  int njobs;
  int64_t T;
  int ns;
  int slab_len;
};
struct ReduceJob {
  const float* partial;
  void* out;
  void* colsum;
  int64_t out_ld;
  int D, Dpad, r;
  int transpose;
  int ones_col;
  float alpha, beta;
  int ns;   // slab count of THIS job (group-planned slabs, tn_rows_plan); 0 = ReduceParams::ns
};
struct ReduceParams {
  ReduceJob job[2];
  int njobs;
  int ns;
  int blocks0;
};
// total_colgroups = 64-column groups over both jobs; total_colgroup_pairs = the same in pairs, rounded up per job
// total_colgroup_quads > 0: the caller's shape takes the fp32 quad kernel (skinny_tn_f32q.hip: four column groups per block)
int tn_pick_slabs(int64_t T, int total_colgroups, int total_colgroup_pairs, int total_colgroup_quads, int dtype, int* slab_len);
bool tn_f32q_shape_ok(int64_t T, int d_in, int d_out);
size_t tn_partial_bytes(int ns, int D);
// skinny_tn_f32q.hip
bool tn_f32q_ok(const TnParams& p);
int launch_tn_f32q(const TnParams& p, hipStream_t stream);
int launch_tn(const TnParams& p, int dtype, hipStream_t stream);
constexpr int TN_MAXG = 8;   // layers per grouped launch of the wide bf16 kernel (a decoder block has 7)
struct TnGroup {
  TnParams p[TN_MAXG];
  int start[TN_MAXG + 1];
  int n;
};
bool tn_group_supported(const TnParams& p, int dtype);
// Row-owner weight-gradient kernel (skinny_tn.hip: tn_partial_rows_kernel): one item per (layer, operand); a workgroup
// owns ALL columns of a token slab (up to 1024 per column range), so S (h / dh) is read once per range instead of once
// per 128 columns.  The slab counts are planned over the whole group (work per block equal across items).
struct TnRowsItem {
  const void* M;      // [T, D] bf16, row pitch ldm
  const void* S;      // [T, 64] bf16
  float* partial;     // [ns][ncg * 64][64] fp32
  int64_t ldm, T;
  int D, ncg;         // columns, 64-column groups
  int nr, gpr, cgw;   // column ranges, groups per range, groups per wave (1 or 2)
  int ns, slab_len;
  int start;          // first block of this item in the grid
};
constexpr int TNR_MAXI = 2 * TN_MAXG;
struct TnRowsGroup {
  TnRowsItem it[TNR_MAXI];
  int n, total;
  int nt_load;   // stream M with the non-temporal hint
};
constexpr int TNR_MAX_SLABS = 40;   // workspace capacity per operand (api.hip: plan_ws)
// n items (T[i] tokens x D[i] columns, at most cap[i] slabs): slab counts such that the blocks of the group fill one
// resident round (256 workgroups) with equal work; false = this group does not suit the kernel (too few / too many blocks)
bool tn_rows_plan(const int64_t* T, const int* D, const int* cap, int n, int* ns_out, int* slab_len_out);
int launch_tn_rows(TnRowsItem* items, int n, hipStream_t stream);
int launch_tn_group(const TnParams* ps, int n, hipStream_t stream);
int launch_tn_reduce(ReduceParams p, int dtype, hipStream_t stream);
int launch_tn_reduce_batch(const ReduceParams* descs, const int* starts, int n, int total_blocks, int dtype, hipStream_t stream);
// gemm.hip
int launch_gemm(const void* A, int64_t lda, bool transA, const void* B, int64_t ldb, bool transB, void* C, int64_t ldc,
                const void* bias, int64_t M, int N, int K, float alpha, float beta, int dtype, hipStream_t stream);
// gemm_x3.hip: the same contract for fp32 tensors on the bf16 matrix pipe (3 x bf16 splits); vector-aligned operands only
int launch_gemm_x3(const void* A, int64_t lda, bool transA, const void* B, int64_t ldb, bool transB, void* C, int64_t ldc,
                   const void* bias, int64_t M, int N, int K, float alpha, float beta, hipStream_t stream);
// gemm2.hip (bf16 streaming GEMM with K-extension: the dense-accumulator form of the layer)
bool gemm2_supported(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                     const void* B2, int64_t ldb2, const void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                     int dtype);
// ws / ws_bytes: optional scratch for split-K (gemm4_splitk_bytes; short M), nullptr = never split
int launch_gemm2(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                 const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                 float alpha, float beta, hipStream_t stream, void* ws = nullptr, size_t ws_bytes = 0);
int launch_pad64(const void* in, void* out, int rows, int r, hipStream_t stream);
// gemm3.hip (same contract as launch_gemm2; one wave per SIMD, 128x128 per wave, hand-interleaved k-loop: long K)
int launch_gemm3(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                 const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                 float alpha, float beta, hipStream_t stream);
// gemm4.hip (same contract as launch_gemm2; 256x256x64 tiles, two wave groups in anti-phase, v_mfma_f32_16x16x32_bf16)
bool gemm4_supported(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                     const void* B2, int64_t ldb2, const void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                     int dtype);
int launch_gemm4(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                 const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                 float alpha, float beta, hipStream_t stream, void* ws = nullptr, size_t ws_bytes = 0);
// split-K of gemm4 for short M (<= 128 output tiles): scratch bytes (0 = the shape does not split), and the split count a
// launch with this scratch takes (1 = none)
size_t gemm4_splitk_bytes(int64_t M, int N, int K, bool has_ext);
int gemm4_splits(int64_t M, int N, int K, bool has_ext, const void* ws, size_t ws_bytes);
// gemm4.hip, gemm4h form: the projection H = hscale * X . op(F) computed by the kernel (a streaming pass over the row panel
// ahead of the main loop) and used as the extension's A operand; F / G zero-padded to 64 columns where they are k-major
bool gemm4h_supported(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                      const void* G, int64_t ldg, const void* C, int64_t ldc, const void* bias, const void* H, int64_t M,
                      int N, int K, int r, int dtype);
int launch_gemm4h(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                  const void* G, int64_t ldg, void* C, int64_t ldc, const void* bias, void* H, int64_t M, int N, int K,
                  int r, float hscale, hipStream_t stream);
// gemm2h.hip (the same product with the projection h = hscale * X . op(F) computed in the kernel: one launch per pass)
bool gemm2h_supported(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                      const void* G, int64_t ldg, const void* C, int64_t ldc, const void* bias, const void* H, int64_t M,
                      int N, int K, int r, int dtype);
int launch_gemm2h(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                  const void* G, int64_t ldg, void* C, int64_t ldc, const void* bias, void* H, int64_t M, int N, int K,
                  int r, float hscale, hipStream_t stream);
struct Gemm2hArgs {   // the arguments of launch_gemm2h for one layer of a grouped launch (same nt for the whole group)
  const void *X, *W, *F, *G;
  void* C;
  const void* bias;
  void* H;
  int64_t M, ldx, ldw, ldf, ldg, ldc;
  int N, K, r;
  float hscale;
};
int launch_gemm2h_group(const Gemm2hArgs* a, int n, bool nt, hipStream_t stream);   // n <= 4, every layer gemm2h_supported
// gemm3s.hip (same contract; 128x128 tiles for products with few 256x256 tiles: short M)
int launch_gemm3s(const void* A, int64_t lda, const void* B, int64_t ldb, bool nt, const void* A2, int64_t lda2,
                  const void* B2, int64_t ldb2, int k2, void* C, int64_t ldc, const void* bias, int64_t M, int N, int K,
                  float alpha, float beta, hipStream_t stream);
// qr.hip
int launch_cast_copy(const void* src, int64_t lds, int src_dtype, void* dst, int64_t ldd, int dst_dtype, int64_t rows,
                     int cols, hipStream_t stream);
int launch_qr_panel(const void* W, int64_t ldw, int in_dtype, int m, int kc, int r, float* Pt, float* Qt,
                    hipStream_t stream);
int launch_qr_copy_out(const float* Qt, const float* Pt, void* Q, int64_t ldq, void* R, int64_t ldr, int out_dtype, int m,
                       int kc, int r, int k_rows, hipStream_t stream);
// accumulate.hip: batched Householder panels (one 1024-thread workgroup per matrix; Pt [kc][m] column-major panel in,
// factored in place; Qt [r_new][m] = Q[:, :r_new] out)
struct QrItem {
  float* Pt;
  float* Qt;
  int m, kc, r_new, pad;
};
int launch_qr_panel_batch(const QrItem* items, int n, hipStream_t stream);
// accumulate.hip: the periodic step of many layers, one launch per phase
struct AccItem {
  void* acc;            // [d_in, d_out] dense accumulator, acc = beta * acc + scale * A . B
  const void* A;        // [d_in, r]
  const void* B;        // [r, d_out]
  const void* draw;     // [d_in, >= kc] Gaussian draw (row stride ld_draw) or nullptr: no re-initialisation
  void* A_new;          // [d_in, r_new] <- Q[:, :r_new] of the draw (may alias A)
  float* Pt;            // workspace: kc * d_in floats
  float* Qt;            // workspace: r_new * d_in floats
  int64_t ld_draw;
  int d_in, d_out, r, r_new, kc;
  float scale, beta;
};
constexpr int ACC_MAXB = 40;   // layers per launch: the by-value argument block is 40 x 96 bytes + 8 = 3848 bytes, under HIP's 4-KiB limit
struct AccBatch {
  AccItem it[ACC_MAXB];
  int n;
};
static_assert(sizeof(AccBatch) <= 4096, "by-value kernel arguments must stay under 4 KiB");
int launch_accumulate_batch(const AccItem* items, int n, int dtype, hipStream_t stream);
// misc.hip
int launch_multi_zero(void* const* ptrs, const int64_t* bytes, int n, hipStream_t stream);
int launch_adamw_flat(void* p, const void* g, void* m, void* v, int64_t n, float lr, float b1, float b2, float eps,
                      float wd, int step, float grad_scale, int dtype, int state_dtype, hipStream_t stream);
int launch_ttadam_dense(float* p, const float* g, float* m, float* v, int64_t n, float b1, float b2, float eps,
                        float step_size, float lr_wd, int clamp_v, hipStream_t stream);
int launch_tt_kron_core(const float* A, const float* B, float* out, int ra0, int rb0, int ij, int ra1, int rb1,
                        hipStream_t stream);
int launch_absmax(const float* x, int64_t n, float* out, hipStream_t stream);
int launch_small_inverse(const float* A, float* out, int batch, int r, hipStream_t stream);
int launch_axpby(const void* x, void* y, int64_t n, float a, float b, int dtype, hipStream_t stream);
}  // namespace sow
