/* sow_amd.h -- C ABI of libsow_amd.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the SoW (Sum-of-Weights) low-rank linear hot path of
 * antoine311200/sow.  The reference has NO foreign-function interface: its hot
 * path is Python (tn_gradient/layer/sow.py, tn_gradient/utils.py,
 * tn_gradient/prepare.py, tn_gradient/tt.py) calling ATen.  Each entry point
 * below replaces the ATen call sequence of the cited reference lines; the
 * host-side mirror (the sow_amd Python package) binds them with ctypes and keeps the
 * reference's class/function surface.
 *
 * Conventions:
 *   - every function returns int: 0 ok, <0 argument error (SOW_ERR_*), >0 hipError_t;
 *   - nothing here allocates or frees device memory or synchronises the device; no mutable global state except
 *     the kernel-selection switches (sow_set_switch), which production code leaves alone;
 *     the caller passes a workspace sized by the matching *_workspace_bytes query;
 *   - all tensors are dense row-major device buffers, 16-byte aligned for the
 *     fast paths (unaligned / odd shapes take slower element-wise paths);
 *   - dtype: SOW_DTYPE_F32 (exact f32 MFMA) or SOW_DTYPE_BF16 (bf16 MFMA, f32 accumulate);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); kernels are
 *     enqueued on it, so the calls are capturable into a hipGraph;
 *   - n_iter > 1 is presented as concatenated factors A = [A_1 .. A_n] ([d_in, n*r]),
 *     B = [B_1; ..; B_n] ([n*r, d_out]); sum_i A_i B_i = A B.
 */
#ifndef SOW_AMD_H
#define SOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOW_DTYPE_F32 0
#define SOW_DTYPE_BF16 1

#define SOW_OK 0
#define SOW_ERR_NULL (-1)
#define SOW_ERR_SHAPE (-2)
#define SOW_ERR_DTYPE (-3)
#define SOW_ERR_ALIGN (-4)
#define SOW_ERR_WORKSPACE (-5)
#define SOW_ERR_UNSUPPORTED (-6)

/* accumulator kinds of SoWLinear.forward (sow.py:109-112) */
#define SOW_ACC_NONE 0    /* acc_downweight empty                                   */
#define SOW_ACC_LOWRANK 1 /* out = (x @ acc_down[d_in,vr]) @ acc_up[vr,d_out]       */
#define SOW_ACC_DENSE 2   /* out = x @ acc_down[d_in,d_out]                         */

#define SOW_H_COLS 64 /* column count of the saved h / dh buffers when r_live <= 64 */

int sow_version(void);
const char* sow_error_string(int code);

/* Kernel-selection switches -- for A/B measurements and for the tests that pin every kernel variant; production code
 * never touches them.  They are the library's ONLY process-wide state: a table of atomics initialised from the
 * environment (SOW_AMD_<NAME>) once, at first use; no launch path calls getenv.  Names: FORCE_CHAIN_V1, NO_SHORT_SPLIT,
 * NO_FUSED_H, FORCE_GEMM_V1, TN_NARROW, NO_GEMM3S, NO_GROUPED, NO_PERSIST, NO_NT_STORE, NT_LOAD, NO_PAIR_FLUSH, F32_EXACT, NO_PARK16, TN_NO_NT_LOAD, NO_TN_ROWS, NO_GEMM4H, NO_CHAIN3F, NO_TN_F32Q, NO_SPLITK
 * (value 1 = on, -1 / 0 = off) and GEMM3S, GEMM3, GEMM4
 * (1 = force, 0 = forbid, -1 = automatic).  sow_set_switch returns SOW_ERR_UNSUPPORTED for an unknown name;
 * sow_get_switch returns the value (-1 / 0 / 1).  Changing a switch while other threads launch is safe (atomic) but
 * the launches in flight may see either value. */
int sow_set_switch(const char* name, int value);
int sow_get_switch(const char* name);

/* Bytes of workspace needed by sow_forward / sow_backward for this shape. */
size_t sow_workspace_bytes(int64_t T, int d_in, int d_out, int r_live, int r_acc, int acc_kind, int dtype);
/* Bytes of workspace sow_forward itself touches: 0 for most bf16 shapes (the caller may then pass NULL / 0), else the
 * same figure as sow_workspace_bytes (a low-rank accumulator wider than 64; short inputs, whose chain is split over K
 * and over the output columns to fill the chip; fp32 inputs with T >= 8192, whose factors are pre-split into bf16 planes
 * there; a bf16 dense accumulator at short T and long K, whose product is split over K).  A caller that passes NULL / 0
 * where the query is non-zero still gets the right result from a slower kernel, except for the wide low-rank accumulator
 * (SOW_ERR_WORKSPACE). */
size_t sow_forward_workspace_bytes(int64_t T, int d_in, int d_out, int r_live, int r_acc, int acc_kind, int dtype);
/* Elements (of dtype) the caller must allocate for h_save: T*64 when r_live <= 64, else T*r_live. */
size_t sow_h_save_elems(int64_t T, int r_live);

/* SoWLinear.forward -- replaces sow.py:107-126:
 *   y = acc_term + scale * (x @ A) @ B + bias,   h_save = scale * (x @ A) for r_live <= 64 (padded to 64
 *   columns, column 63 = 1.0 when free) or x @ A for r_live > 64; opaque to the caller, kept for backward.
 * x [T,d_in], A [d_in,r_live], B [r_live,d_out], y [T,d_out]; acc_down/acc_up per acc_kind
 * (r_acc = vr for SOW_ACC_LOWRANK, ignored otherwise); bias [d_out] or NULL.
 * The accumulator term is NOT scaled (sow.py:110-112). */
int sow_forward(const void* x, const void* A, const void* B, const void* acc_down, const void* acc_up, const void* bias,
                void* y, void* h_save, int64_t T, int d_in, int d_out, int r_live, int r_acc, int acc_kind, float scale,
                int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the above (what autograd derives from sow.py:107-126):
 *   dh = scale * dY @ B^T ; dB = scale * h^T @ dY ; dA = x^T @ dh ;
 *   dX = dh @ A^T + acc-term ; dbias = sum_t dY.
 * Gradients are written as  g = grad_beta * g + new  (grad_beta = 0 overwrites, 1 accumulates).
 * dbias may be NULL (no bias).  dx must be non-NULL. */
int sow_backward(const void* dy, const void* x, const void* h_save, const void* A, const void* B, const void* acc_down,
                 const void* acc_up, void* dx, void* dA, void* dB, void* dbias, int64_t T, int d_in, int d_out,
                 int r_live, int r_acc, int acc_kind, float scale, float grad_beta, int dtype, void* workspace,
                 size_t workspace_bytes, void* stream);

/* The same in two phases, so that the weight-gradient kernels (off the critical path of backpropagation)
 * can run on another stream than the data-gradient kernel:
 *   SOW_BWD_DATA    : dX and the internal dh = scale * dY @ B^T (kept in `workspace`)
 *   SOW_BWD_WEIGHTS : dA, dB, dbias from x, dY, h_save and the dh left in the SAME workspace by a
 *                     preceding SOW_BWD_DATA call (the caller orders the two calls, e.g. with an event).
 * SOW_BWD_WEIGHTS = SOW_BWD_WEIGHTS_PARTIAL (the token-slab partial sums, left in `workspace`) followed by
 * SOW_BWD_WEIGHTS_REDUCE (their fixed-order sum into dA / dB / dbias); the two may also be requested separately, e.g.
 * to run the small reduction beside the next layer's kernels (r_live <= 64; for wider ranks PARTIAL does everything
 * and REDUCE nothing).
 * phases = SOW_BWD_DATA | SOW_BWD_WEIGHTS is sow_backward. */
#define SOW_BWD_DATA 1
#define SOW_BWD_WEIGHTS 2
#define SOW_BWD_WEIGHTS_PARTIAL 4
#define SOW_BWD_WEIGHTS_REDUCE 8
/* sow_backward_group only: the token-slab counts of the weight-gradient partial sums may be planned over the whole group
 * (row-owner kernel, see sow_backward_group) although the reduction is deferred; the deferred reduction must then be
 * built with sow_backward_group_reduce_desc(same layers, same flag).  Implied when one call runs PARTIAL and REDUCE. */
#define SOW_BWD_GROUP_SLABS 16
int sow_backward_ex(const void* dy, const void* x, const void* h_save, const void* A, const void* B,
                    const void* acc_down, const void* acc_up, void* dx, void* dA, void* dB, void* dbias, int64_t T,
                    int d_in, int d_out, int r_live, int r_acc, int acc_kind, float scale, float grad_beta, int dtype,
                    void* workspace, size_t workspace_bytes, int phases, void* stream);

/* Deferred, batched reduction of the weight gradients.  A training step calls sow_backward_ex(phases = SOW_BWD_DATA |
 * SOW_BWD_WEIGHTS_PARTIAL) for every layer, each with its OWN workspace (the slab partials stay there), and sums all of
 * them in ONE launch before the gradients are consumed (optimizer step / all-reduce): one launch instead of one 5-us
 * launch per layer, same per-element summation order as SOW_BWD_WEIGHTS_REDUCE (bit-identical results).
 *   sow_reduce_desc_bytes()    size of one opaque descriptor
 *   sow_backward_reduce_desc() writes the descriptor of one layer to HOST memory `desc_out` and its block count to
 *                              `blocks_out` (same shape / pointer arguments as the sow_backward_ex call it completes;
 *                              SOW_ERR_UNSUPPORTED for r_live > 64, or r_live = 64 with dbias: those have no separate
 *                              reduction -- use SOW_BWD_WEIGHTS)
 *   sow_reduce_batch()         `descs`: n descriptors back to back in DEVICE memory; `starts`: n ints in device memory,
 *                              starts[i] = sum of the block counts of layers < i; total_blocks = their total. */
size_t sow_reduce_desc_bytes(void);
int sow_backward_reduce_desc(void* dA, void* dB, void* dbias, int64_t T, int d_in, int d_out, int r_live, int r_acc,
                             int acc_kind, float grad_beta, int dtype, void* workspace, size_t workspace_bytes, void* desc_out,
                             int* blocks_out);
int sow_reduce_batch(const void* descs, const int* starts, int n, int total_blocks, int dtype, void* stream);

/* Grouped calls: n INDEPENDENT SoWLinear invocations in as few launches as possible -- the q / k / v projections of
 * an attention block (sow.py:107-126 called three times on the same hidden state by the HF model), gate / up of an MLP,
 * and their backward passes.  Each element carries exactly the arguments of sow_forward / sow_backward_ex for its
 * layer (fields a direction does not use are ignored).  Semantics = the n single calls, in any order; results are
 * bit-identical to them.  Layers that run the bf16 streaming kernels (no accumulator, r_live <= 64, T > 8192) share one
 * grid per kernel, up to 4 layers at a time: a launch costs ~8 us of ramp + first-load latency + write drain whatever
 * its size, which a 3-layer grid pays once.  Every other layer is forwarded to the single-layer entry point. */
typedef struct sow_layer_args {
  const void* x;        /* [T, d_in]                                      */
  const void* A;        /* [d_in, r_live]                                 */
  const void* B;        /* [r_live, d_out]                                */
  const void* acc_down; /* per acc_kind, or NULL                          */
  const void* acc_up;
  const void* bias;     /* [d_out] or NULL                                */
  void* y;              /* forward output [T, d_out]                      */
  void* h_save;         /* sow_h_save_elems(T, r_live) elements           */
  const void* dy;       /* backward: upstream gradient [T, d_out]         */
  void* dx;             /* [T, d_in]                                      */
  void* dA;
  void* dB;
  void* dbias;          /* or NULL                                        */
  int64_t T;
  int32_t d_in, d_out, r_live, r_acc, acc_kind;
  float scale, grad_beta;
  void* workspace;      /* this layer's own workspace (sow_workspace_bytes) */
  size_t workspace_bytes;
} sow_layer_args;
int sow_forward_group(const sow_layer_args* layers, int n, int dtype, void* stream);
/* phases as in sow_backward_ex; the phases run in order DATA (all layers), WEIGHTS_PARTIAL (all), WEIGHTS_REDUCE (all). */
int sow_backward_group(const sow_layer_args* layers, int n, int dtype, int phases, void* stream);
/* Weight gradients of a group: with enough layers to fill the chip (e.g. the 7 projections of a llama decoder block) the
 * partial sums run in the ROW-OWNER kernel -- a workgroup owns all columns of a token slab, so h / dh are read once instead
 * of once per 128 columns and x / dY arrive as whole rows -- with slab counts planned over the group (equal work per
 * workgroup, one resident round; a pure function of the layer list).  The sums are then added in a different (still
 * fixed) order than by n single calls: dA / dB / dbias agree with them to fp32 rounding of the slab sums, not bit for bit.
 * sow_backward_group_reduce_desc: the descriptors (n x sow_reduce_desc_bytes(), HOST memory) and block counts of the
 * deferred reductions of exactly this group, for sow_reduce_batch; `phases` = the flags of the PARTIAL call. */
int sow_backward_group_reduce_desc(const sow_layer_args* layers, int n, int dtype, int phases, void* descs_out, int* blocks_out);
/* What sow_backward_group(layers, n, dtype, phases) does for the weight gradients: returns 1 if the row-owner kernel with
 * group-planned slabs runs, 0 if every layer keeps its single-layer slab count (negative: error); slabs_out (2 n ints,
 * may be NULL) receives the token-slab counts of the x / dY operand of every layer. */
int sow_backward_group_plan(const sow_layer_args* layers, int n, int dtype, int phases, int* slabs_out);

/* General row-major GEMM  C[M,N] = alpha * op(A) op(B) + beta * C + bias[N]  (bias may be NULL).
 * trans_a: A is stored [K,M]; trans_b: B is stored [N,K].  Replaces the plain `@` / einsum call
 * sites: accumulate() sow.py:131-140 (W_acc += scale * A @ B, Q @ R), prepare.py:135, tt.py:213-237. */
int sow_gemm(const void* A, int64_t lda, int trans_a, const void* B, int64_t ldb, int trans_b, void* C, int64_t ldc,
             const void* bias, int64_t M, int N, int K, float alpha, float beta, int dtype, void* stream);

/* The same with scratch for the caller's stream: short bf16 products (M <= ~1024 rows of a wide output: fewer output tiles
 * than the chip has CUs) are split over K across workgroups, fp32 partial sums through `workspace`, summed in a fixed
 * order.  sow_gemm_workspace_bytes returns 0 when the shape does not split (workspace may then be NULL); sow_gemm is
 * sow_gemm_ex without scratch. */
size_t sow_gemm_workspace_bytes(int64_t M, int N, int K, int trans_a, int dtype);
int sow_gemm_ex(const void* A, int64_t lda, int trans_a, const void* B, int64_t ldb, int trans_b, void* C, int64_t ldc,
                const void* bias, int64_t M, int N, int K, float alpha, float beta, int dtype, void* workspace,
                size_t workspace_bytes, void* stream);

/* Truncated Householder QR -- replaces qr_weight (utils.py:8-30) and the truncated complete-mode QR
 * of TensorTrain.decompose (tt.py:128-136):  Q_out[m,k] = Q[:, :k], R_out[k,n] = R[:k, :]
 * with LAPACK's sign convention.  W [m,n] (ldw) of in_dtype; outputs of out_dtype; internals fp32.
 * R_out may be NULL (sow.py:168-172 only needs Q).  k <= m. */
size_t sow_qr_workspace_bytes(int m, int n, int k, int in_dtype, int need_r);
int sow_qr_thin(const void* W, int64_t ldw, int m, int n, int in_dtype, int k, void* Q_out, int64_t ldq, void* R_out,
                int64_t ldr, int out_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* The periodic step of MANY layers -- replaces the loop of tn_gradient/prepare.py:219-222 over
 * SoWLinear.accumulate (sow.py:128-178) for layers on the dense-accumulator branch (the one every script takes:
 * prepare.py:120 forces virtual_rank = min(in, out)).  Per item, in stream order:
 *     acc   = acc_beta * acc + scale * A . B          (sow.py:131-140; acc_beta = 0 materialises a first accumulator)
 *     A_new = Q[:, :r_new] of the Householder QR of `draw` [d_in, draw_cols]   (sow.py:161-172; skipped when draw = NULL)
 *     `zero` buffer <- 0                                (B <- 0, sow.py:159)
 * One launch per phase for all items (the per-layer QR panel is latency-bound on one CU; n of them run side by side).
 * A_new may alias A: it is written after the update has consumed A.  Workspace per item:
 * sow_qr_workspace_bytes(d_in, draw_cols, r_new, dtype, 0).  r <= 64. */
typedef struct sow_accumulate_args {
  void* acc;
  const void* A;
  const void* B;
  const void* draw;
  int64_t ld_draw;
  void* A_new;
  void* zero;
  int64_t zero_bytes;
  int32_t d_in, d_out, r, r_new, draw_cols;
  float scale, acc_beta;
  void* workspace;
  size_t workspace_bytes;
} sow_accumulate_args;
int sow_accumulate_batch(const sow_accumulate_args* items, int n, int dtype, void* stream);

/* Multi-tensor zero fill -- replaces the per-parameter torch.zeros_like of reset_optimizer
 * (scripts/utils/training_utils.py:257-277) and B <- 0 of sow.py:159.  ptrs/bytes are HOST arrays. */
int sow_zero_state(void* const* ptrs, const int64_t* bytes, int n, void* stream);

/* AdamW step over one flat parameter buffer (the factor param group of simple_train.py:502-506).
 * state_dtype = dtype of exp_avg / exp_avg_sq.  step is the 1-based step count. */
int sow_adamw_flat(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, int dtype, int state_dtype,
                   void* stream);

/* TTAdam dense section (ttadam.py:84-111), fp32 buffers. */
int sow_ttadam_dense(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                     float beta2, float eps, float step_size, float lr_times_wd, int clamp_v, void* stream);

/* TT Hadamard product of two cores (tt.py:469-475): out[(a,c),ij,(b,d)] = A[a,ij,b] * B[c,ij,d], fp32. */
int sow_tt_kron_core(const float* A, const float* B, float* out, int ra0, int rb0, int ij, int ra1, int rb1,
                     void* stream);

/* Tensor-train optimizer state on device, batched over trains.  A train is described by its fp32 cores
 * (core k: [ranks[k], in_dims[k], out_dims[k], ranks[k+1]], contiguous, ranks[0] = ranks[order] = 1) and the shape
 * [rows, cols] of the matrix it represents (rows <= prod(in_dims), cols <= prod(out_dims): TensorTrain.from_matrix pads,
 * to_matrix un-pads -- tt.py:48-67, 242-247, utils.py:78-87).  ranks <= 32, order <= SOW_TT_MAX_ORDER, ranks[k+1] <=
 * ranks[k] * in_dims[k] * out_dims[k]; anything else returns SOW_ERR_UNSUPPORTED (the caller keeps the per-train path).
 *   sow_tt_reconstruct_batch  out_i[rows, cols] = TensorTrain.to_matrix (tt.py:213-247), one launch per 16 trains;
 *   sow_tt_decompose_batch    cores_i <- TensorTrain.from_matrix(mat_i, ranks, padding=True) (tt.py:48-67 + decompose
 *                             :111-140: sequential truncated complete-mode QR, LAPACK sign convention): per bond ONE
 *                             Householder-panel launch for all trains; workspace per train:
 *                             sow_tt_decompose_workspace_bytes;
 *   sow_ttadam_batch          TTAdam.step (ttadam.py:68-115) for n parameters: reconstruct m and v, clamp v < 0, Adam
 *                             update of the parameter (+ the decoupled weight-decay line :110-111), re-decompose both
 *                             moments into the cores given (in place over the old ones); has_state = 0 on the first step
 *                             (m = v = 0).  step_size already carries the bias correction of :95-100.  Workspace per
 *                             item: sow_ttadam_workspace_bytes(&item.m). */
#define SOW_TT_MAX_ORDER 6
typedef struct sow_tt_desc {
  void* cores[SOW_TT_MAX_ORDER];
  int32_t order;
  int32_t ranks[SOW_TT_MAX_ORDER + 1];
  int32_t in_dims[SOW_TT_MAX_ORDER];
  int32_t out_dims[SOW_TT_MAX_ORDER];
  int32_t rows, cols;
} sow_tt_desc;
typedef struct sow_ttadam_item {
  sow_tt_desc m, v;        /* exp_avg / exp_avg_sq trains (read when has_state, always written) */
  float* param;            /* [rows, cols] fp32, row pitch ld_param */
  const float* grad;
  int64_t ld_param, ld_grad;
  float step_size, lr_times_wd;
  int32_t has_state;
  void* workspace;
  size_t workspace_bytes;
} sow_ttadam_item;
size_t sow_tt_decompose_workspace_bytes(const sow_tt_desc* tt);
size_t sow_ttadam_workspace_bytes(const sow_tt_desc* tt);
int sow_tt_reconstruct_batch(const sow_tt_desc* tts, void* const* out, const int64_t* ld_out, int n, void* stream);
int sow_tt_decompose_batch(const sow_tt_desc* tts, const void* const* mats, const int64_t* ld, int n, void* const* workspaces,
                           const size_t* workspace_bytes, void* stream);
int sow_ttadam_batch(const sow_ttadam_item* items, int n, float beta1, float beta2, float eps, void* stream);

/* out[0] = max |x[i]| over n fp32 elements (TensorTrain.sqrt / sqrtinv scaling, tt.py:288, 322). */
int sow_absmax(const float* x, int64_t n, float* out, void* stream);

/* Batched inverse of `batch` small [r, r] fp32 matrices, r <= 16 (TensorTrain.reciprocal, tt.py:480-494). */
int sow_small_inverse(const float* A, float* out, int batch, int r, void* stream);

/* y = a*x + b*y over n elements. */
int sow_axpby(const void* x, void* y, int64_t n, float a, float b, int dtype, void* stream);

/* Strided 2-D cast copy between f32 / bf16. */
int sow_cast_copy(const void* src, int64_t lds, int src_dtype, void* dst, int64_t ldd, int dst_dtype, int64_t rows,
                  int cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SOW_AMD_H */
