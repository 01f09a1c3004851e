// Fused low-rank chain, bf16 streaming version (gfx950):  Y = beta*Y + (scale * X.F1).F2 + bias
//
// Same contract as chain.hip (reference tn_gradient/layer/sow.py:107-126 forward, and its autograd
// backward with F1 = B^T, F2 = A^T), rebuilt around the CDNA4 features that matter for an HBM-bound
// kernel:
//   * LDS-DMA (`global_load_lds_dwordx4`) for EVERYTHING that is read: each compute wave streams its own
//     32 token rows of X through a private ring of [32 x 64] stages (no VGPR staging, no barrier on
//     the X path, counted s_waitcnt vmcnt); four loader waves stream the factors as 64-row chunks
//     into an 8-slot LDS ring, running 6 chunks ahead of the consumers.
//   * `ds_read_b64_tr_b16`: the factors stay in their storage layout (A is [d_in, r], B is [r, d_out]);
//     in the forward direction both have the contraction index as their ROW index and are read
//     transposed; in the backward direction both are k-contiguous and read with ds_read_b128.
//   * 16-byte loads need only 4-byte alignment on gfx950 (tools/probe2.hip): the 100-byte rows of a
//     rank-50 A are DMA'd as 128-byte rows whose tail is the head of the next row ("padding by
//     overlap"); the garbage lands in rank columns >= r, which always meet an explicit zero (rows >= r of
//     B come from a zero page, H columns >= r are masked), so it never reaches a result.  The last row of
//     A, whose tail would cross the end of the buffer, is rewritten by the loader from a guarded load.
// Workgroup = 4 waves: compute waves 0-1 (32 tokens each, 64 per workgroup), loader waves 2-3; 80 KiB of
// LDS, so two workgroups share a CU and are staggered by one phase (see the kernel body).
// One raw s_barrier per chunk hands chunk c to the consumers and frees slot (c-2) % 4 for the loader.
// Per compute wave: phase 1 accumulates H[32,64] over K; H is scaled, rounded and parked in the
// wave's own LDS (it leaves the CU only as the saved copy for backward); phase 2 produces Y in
// 64-column slices written as 16-byte row segments.
// All LDS reads of the compute waves are inline asm: for a compiler-visible LDS read hipcc emits
// `s_waitcnt vmcnt(0)` while LDS-DMA is outstanding, which would drain the rings every step.
//
// LDS images are 64 rows x 128 B; 16-byte chunk c of a row sits at physical chunk
//   c ^ ((row >> 1) & 7)           for ds_read_b128 consumers (X stage, H, backward factors)
//   c ^ (((row >> 1) & 1) << 2)    for transposed-read consumers (forward factors)
// DMA writes LDS lane-linearly, so the XOR is applied to the per-lane SOURCE address.
#include <stdlib.h>

#include "kernels.hpp"

#ifndef SOW_CHAIN2_DIAG
#define SOW_CHAIN2_DIAG 0   // 1: honour SOW_AMD_CHAIN2_DEBUG / _DBGBUF (timing experiments, tools/chain_*.py)
#endif

namespace sow {

constexpr int C2_NCW = 2;             // compute waves per workgroup (32 tokens each)
constexpr int C2_NLW = 2;             // loader waves per workgroup
constexpr int C2_BM = 32 * C2_NCW;    // tokens per workgroup
constexpr int C2_DEPTH = 6;           // X stages in flight per compute wave
constexpr int C2_STAGE = 4096;        // [32 tok][64 k] bf16
constexpr int C2_NSLOT = 4;           // factor chunk slots
constexpr int C2_AHEAD = 2;           // chunks the loaders run ahead (slot c+AHEAD held chunk c+AHEAD-NSLOT <= c-2)
constexpr int C2_FSLOT = 8192;        // [64][64] bf16
constexpr int C2_LPW = 8 / C2_NLW;    // 1-KiB DMA instructions per loader wave per chunk
constexpr int C2_RING0 = C2_NSLOT * C2_FSLOT;
constexpr int C2_RING = C2_DEPTH * C2_STAGE;  // 24 KiB per compute wave
constexpr int C2_LDS = C2_RING0 + C2_NCW * C2_RING;   // 80 KiB: two workgroups per CU
constexpr int C2_THREADS = 64 * (C2_NCW + C2_NLW);

__device__ __attribute__((aligned(256))) uint32_t g_zero_page2[64];

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
#define DS_READ_B128(dst, addr, off) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DS_READ_B64(dst, addr, off) \
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DS_READ_TR(dst, addr, off) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define LGKM_WAIT0()                                  \
  do {                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)

__device__ __forceinline__ void raw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// wait until all but the `newer` most recent groups of PER instructions have completed
template <int PER> __device__ __forceinline__ void wait_groups(int newer) {
  switch (newer) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PER) : "memory"); break;
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
template <bool TR> __device__ __forceinline__ int img_chunk(int row, int c) {
  return TR ? (c ^ (((row >> 1) & 1) << 2)) : (c ^ ((row >> 1) & 7));
}

// =================================================================================================
// BWD = false: forward  (F1 = A [D1, r] rows = k, F2 = B [r, D2] rows = k  -> transposed reads)
// BWD = true : backward (F1 = B [r, D1] rows = rank, F2 = A [D2, r] rows = n -> b128 reads)
template <bool BWD> __global__ __launch_bounds__(C2_THREADS, 2) void chain2_kernel(const ChainParams p) {
  constexpr bool TR = !BWD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * C2_BM;
  const int D1 = p.D1, D2 = p.D2, rb = p.rb;
  const int dflags = SOW_CHAIN2_DIAG ? p.fast_factors : 0;   // diagnostics only; constant 0 in production
  // diagnostic timeline (SOW_AMD_CHAIN2_DEBUG bit 32 + SOW_AMD_CHAIN2_DBGBUF): never set in production
  unsigned long long* tl_buf = (dflags & 32) ? (unsigned long long*)p.F1a + ((int64_t)blockIdx.x * 4 + w) * 16 : nullptr;
  unsigned long long t_wait = 0, t_mark = 0;
  auto stamp = [&](int slot) {
    if (tl_buf && lane == 0) tl_buf[slot] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  if (tl_buf && lane == 0) {
    tl_buf[10] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    tl_buf[11] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
  }
  const int nst = (D1 + 63) / 64;   // phase-1 chunks = X stages
  const int nsl = (D2 + 63) / 64;   // phase-2 chunks = output slices
  const int total = nst + nsl;
  const bf16_t* Amat = (const bf16_t*)(BWD ? p.F2b : p.F1b);   // [rows_a, rb] contiguous
  const bf16_t* Bmat = (const bf16_t*)(BWD ? p.F1b : p.F2b);   // [rb, cols_b], ld = ldb
  const int64_t ldb = BWD ? p.ldf1b : p.ldf2b;
  const int rows_a = BWD ? D2 : D1, cols_b = BWD ? D1 : D2;

  // Stagger: the two workgroups that share a CU (ids b and b + grid/2 under round-robin placement; only
  // speed depends on that, never correctness) should be half a kernel apart, so that one streams X in
  // while the other streams Y out.  The second half of the grid sleeps for about one phase-1 duration:
  // its X bytes at the CU's fair share of HBM bandwidth (~10 B/clk).
  if ((dflags & 64) && 2 * blockIdx.x >= gridDim.x) {   // measured: no gain (per-wave issue-bound), off by default
    const unsigned long long t_go = __builtin_amdgcn_s_memtime() + (unsigned long long)(C2_BM * D1 * 2) / 10u;
    while (__builtin_amdgcn_s_memtime() < t_go) __builtin_amdgcn_s_sleep(8);
  }
  if (w >= C2_NCW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - C2_NCW;
    const char* a_end = (const char*)(Amat + (int64_t)rows_a * rb);
    // the last row of A, kept in a register for the fix-up by the loader wave that DMAs that row
    // (rows 16*lw .. 16*lw+15 of a chunk belong to loader wave lw, so its own counted wait orders the
    // fix-up after its DMA)
    const bool own_last = lw == ((rows_a - 1) & 63) / (8 * C2_LPW);
    uint32_t last_row_dw = 0u;
    if (own_last && lane < 32 && 2 * lane < rb) last_row_dw = *((const uint32_t*)(Amat + (int64_t)(rows_a - 1) * rb) + lane);
    asm volatile("" : "+v"(last_row_dw));  // consume now: the compiler's wait for this load lands here, not mid-pipeline
    auto chunk_is_a = [&](int c) { return BWD ? (c >= nst) : (c < nst); };
    // per-lane source pointers of chunk 0, advanced by a constant per chunk (address arithmetic is the
    // loaders' critical path at one wave per SIMD, so it is hoisted out of the loop)
    const char* zp = (const char*)(g_zero_page2 + (lane & 7) * 4);
    const char* a_ptr[C2_LPW];
    const char* b_ptr[C2_LPW];
    int b_stride[C2_LPW], b_lc[C2_LPW];
#pragma unroll
    for (int ii = 0; ii < C2_LPW; ++ii) {
      const int i = C2_LPW * lw + ii;
      const int row = 8 * i + (lane >> 3), pc = lane & 7;
      const int lc = img_chunk<TR>(row, pc);
      a_ptr[ii] = (const char*)(Amat + (int64_t)row * rb) + 16 * lc;
      const bool bv = row < rb;
      b_ptr[ii] = bv ? (const char*)(Bmat + (int64_t)row * ldb + 8 * lc) : zp;
      b_stride[ii] = bv ? 128 : 0;
      b_lc[ii] = lc;
    }
    const int a_chunk_bytes = 128 * rb;   // 64 rows of 2*rb bytes
    const bool b_ragged = (cols_b & 63) != 0;
    const int nb_chunks = (cols_b + 63) / 64;
    auto issue = [&](int c) {
      if (dflags & 8) return;   // timing experiment: no factor DMA
      char* slot = smem + (c % C2_NSLOT) * C2_FSLOT;
      const int ci = c < nst ? c : c - nst;   // chunk index inside its matrix
      if (chunk_is_a(c)) {
#pragma unroll
        for (int ii = 0; ii < C2_LPW; ++ii) {
          // A rows: 2*rb bytes each, read as 128-byte rows (tail = head of the next row); pieces that
          // would cross the end of the buffer read zeros (the last row is rewritten by the fix-up)
          const char* q = a_ptr[ii] + (int64_t)ci * a_chunk_bytes;
          dma16(q + 16 <= a_end ? (const void*)q : (const void*)zp, slot + (C2_LPW * lw + ii) * 1024);
        }
      } else {
#pragma unroll
        for (int ii = 0; ii < C2_LPW; ++ii) {
          const char* q = b_ptr[ii] + ci * b_stride[ii];
          if (b_ragged && ci == nb_chunks - 1 && ci * 64 + 8 * b_lc[ii] >= cols_b) q = zp;
          dma16((const void*)q, slot + (C2_LPW * lw + ii) * 1024);
        }
      }
    };
    const int pre = total < C2_AHEAD ? total : C2_AHEAD;
    for (int c = 0; c < pre; ++c) issue(c);
    stamp(1);
    for (int c = 0; c < total; ++c) {
      const int newer = (total - 1 - c) < (C2_AHEAD - 1) ? (total - 1 - c) : (C2_AHEAD - 1);
      if (tl_buf) t_mark = __builtin_amdgcn_s_memtime();
      if (!(dflags & 8)) wait_groups<C2_LPW>(newer);
      if (tl_buf) t_wait += __builtin_amdgcn_s_memtime() - t_mark;
      if (c == nst) stamp(2);
      // fix-up: rewrite the last row of A (its DMA pieces past the end of the buffer were zero-filled)
      if (chunk_is_a(c)) {
        const int base = (c < nst ? c : c - nst) * 64;
        const int lr = rows_a - 1 - base;
        if (own_last && lr >= 0 && lr < 64 && lane < 32) {
          const int cc = lane >> 2;
          *(uint32_t*)(smem + (c % C2_NSLOT) * C2_FSLOT + lr * 128 + img_chunk<TR>(lr, cc) * 16 + (lane & 3) * 4) = last_row_dw;
        }
      }
      raw_barrier();   // chunk c visible to the consumers; they have finished chunk c-1
      if (c + C2_AHEAD < total) issue(c + C2_AHEAD);   // slot held chunk c-2: free
    }
    stamp(4);
    if (tl_buf && lane == 0) tl_buf[8] = t_wait;
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int li = lane & 31, lh = lane >> 5;
  const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3;  // transposed-read geometry
  char* ring = smem + C2_RING0 + w * C2_RING;
  const uint32_t ring_a = lds_addr(ring);
  const uint32_t slot_a = lds_addr(smem);
  const bf16_t* X = (const bf16_t*)p.X;
  const int64_t tok0 = m0 + 32 * w;

  const int drow = lane >> 3, dpc = lane & 7;
  const char* zp = (const char*)(g_zero_page2 + (lane & 7) * 4);
  const char* xsrc[4];
  int xstride[4], xlc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + drow;
    const int lc = dpc ^ ((row >> 1) & 7);
    const int64_t tk = tok0 + row;
    const bool v = tk < p.M;
    xsrc[i] = v ? (const char*)(X + tk * p.ldx + lc * 8) : zp;
    xstride[i] = v ? 128 : 0;
    xlc[i] = lc;
  }
  const bool x_ragged = (D1 & 63) != 0;
  auto issue_x = [&](int st) {
    char* dst = ring + (st % C2_DEPTH) * C2_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* q = xsrc[i] + st * xstride[i];
      if (x_ragged && st == nst - 1 && st * 64 + xlc[i] * 8 >= D1) q = zp;
      dma16((const void*)q, dst + i * 1024);
    }
  };

  // per-lane LDS offsets
  const uint32_t xoff = (uint32_t)(li * 128);   // this lane's row in an X stage
  const int xsw = (li >> 1) & 7;                // b128 row swizzle
  // Factor fragments.  Both products are computed TRANSPOSED (H^T = F1^T X^T, Y^T = F2^T H^T) so that the
  // token index sits on the MFMA lane: H^T's accumulator registers are then directly the B operand
  // of phase 2 (k order permuted: element j of lane-half h is rank 16s + 8(j>>2) + 4h + (j&3), cdna
  // guide "accumulator tile as the next MFMA's operand") and Y^T leaves the wave as 8-byte row pieces
  // -- no LDS round trip for H, no epilogue scratch.
  //   phase 1 (natural k):  TR rows 16ks + 8h + q (+4)      | B128 chunk 2ks + h of row (tile*32 + li)
  //   phase 2 (permuted k): TR rows 16ks + 4h + q (+8)      | two B64 at k = 16ks + 4h and 16ks + 8 + 4h
  const int h2 = g >> 1;
  uint32_t foff1[2], foff2[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    if constexpr (TR) {
      const int col = tl * 32 + 16 * (g & 1) + 4 * pp;
      const int r1 = 8 * h2 + q, r2 = 4 * h2 + q;
      foff1[tl] = (uint32_t)(r1 * 128 + img_chunk<true>(r1, col >> 3) * 16 + (col & 7) * 2);
      foff2[tl] = (uint32_t)(r2 * 128 + img_chunk<true>(r2, col >> 3) * 16 + (col & 7) * 2);
    } else {
      foff1[tl] = (uint32_t)((tl * 32 + li) * 128);
      foff2[tl] = (uint32_t)((tl * 32 + li) * 128 + 8 * lh);
    }
  }
  // Operand fragments of one step.  Reads are issued WITHOUT waiting (software pipeline: the reads of
  // step s+1 are in flight while the MFMAs of step s execute); `frag` assembles after the wait.
  struct Frags {
    u32x4 x[4];            // X fragments (phase 1)
    u32x2 f2[4][2][2];     // factor fragments as two 8-byte halves: transposed reads, b64 pairs
    u32x4 f4[4][2];        // factor fragments as one 16-byte read (backward phase 1)
  };
  auto issue_reads_p1 = [&](int st, Frags& F) {
    const uint32_t xs = ring_a + (uint32_t)((st % C2_DEPTH) * C2_STAGE) + xoff;
    const uint32_t fs = slot_a + (uint32_t)((st % C2_NSLOT) * C2_FSLOT);
    DS_READ_B128(F.x[0], xs + (uint32_t)(((0 + lh) ^ xsw) * 16), 0);
    DS_READ_B128(F.x[1], xs + (uint32_t)(((2 + lh) ^ xsw) * 16), 0);
    DS_READ_B128(F.x[2], xs + (uint32_t)(((4 + lh) ^ xsw) * 16), 0);
    DS_READ_B128(F.x[3], xs + (uint32_t)(((6 + lh) ^ xsw) * 16), 0);
    if constexpr (TR) {
      const uint32_t b0 = fs + foff1[0], b1 = fs + foff1[1];
#define F_TR(ks)                                  \
  DS_READ_TR(F.f2[ks][0][0], b0, ks * 2048);      \
  DS_READ_TR(F.f2[ks][0][1], b0, ks * 2048 + 512); \
  DS_READ_TR(F.f2[ks][1][0], b1, ks * 2048);      \
  DS_READ_TR(F.f2[ks][1][1], b1, ks * 2048 + 512);
      F_TR(0) F_TR(1) F_TR(2) F_TR(3)
#undef F_TR
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint32_t o = (uint32_t)(((2 * ks + lh) ^ xsw) * 16);
        DS_READ_B128(F.f4[ks][0], fs + foff1[0] + o, 0);
        DS_READ_B128(F.f4[ks][1], fs + foff1[1] + o, 0);
      }
    }
  };
  auto issue_reads_p2 = [&](int sl, Frags& F) {
    const uint32_t fs = slot_a + (uint32_t)(((nst + sl) % C2_NSLOT) * C2_FSLOT);
    if constexpr (TR) {
      const uint32_t b0 = fs + foff2[0], b1 = fs + foff2[1];
#define F_TR(ks)                                   \
  DS_READ_TR(F.f2[ks][0][0], b0, ks * 2048);       \
  DS_READ_TR(F.f2[ks][0][1], b0, ks * 2048 + 1024); \
  DS_READ_TR(F.f2[ks][1][0], b1, ks * 2048);       \
  DS_READ_TR(F.f2[ks][1][1], b1, ks * 2048 + 1024);
      F_TR(0) F_TR(1) F_TR(2) F_TR(3)
#undef F_TR
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint32_t o0 = (uint32_t)(((2 * ks) ^ xsw) * 16), o1 = (uint32_t)(((2 * ks + 1) ^ xsw) * 16);
        DS_READ_B64(F.f2[ks][0][0], fs + foff2[0] + o0, 0);
        DS_READ_B64(F.f2[ks][0][1], fs + foff2[0] + o1, 0);
        DS_READ_B64(F.f2[ks][1][0], fs + foff2[1] + o0, 0);
        DS_READ_B64(F.f2[ks][1][1], fs + foff2[1] + o1, 0);
      }
    }
  };
  auto frag2 = [&](const Frags& F, int ks, int tl) {
    return (u32x4){F.f2[ks][tl][0][0], F.f2[ks][tl][0][1], F.f2[ks][tl][1][0], F.f2[ks][tl][1][1]};
  };

  f32x16 hacc[2];   // H^T tiles: lane = token, registers = rank rows of tile rt
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[a][i] = 0.f;

  const int pre = nst < C2_DEPTH ? nst : C2_DEPTH;
  for (int st = 0; st < pre; ++st) issue_x(st);

  // ================================================================== phase 1: H^T = F1^T . X^T
  // step(st): [wait X(st+1), barrier(st+1)] -> wait reads(st) -> issue reads(st+1) -> MFMA(st) -> DMA X(st+DEPTH)
  unsigned long long t_bar = 0;
  auto wait_stage = [&](int st, int ahead) {   // `ahead` = stages issued after stage st at this point (upper bound)
    const int newer = (nst - 1 - st) < ahead ? (nst - 1 - st) : ahead;
    if (tl_buf) t_mark = __builtin_amdgcn_s_memtime();
    wait_groups<4>(newer);   // this wave's X stage `st` has landed
    if (tl_buf) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_wait += t2 - t_mark; t_mark = t2; }
    raw_barrier();           // factor chunk `st` is in slot st % 8
    if (tl_buf) t_bar += __builtin_amdgcn_s_memtime() - t_mark;
  };
  auto step1 = [&](int st, Frags& cur, Frags& nxt) {
    if (st + 1 < nst) wait_stage(st + 1, C2_DEPTH - 2);   // X(st+DEPTH) is issued at the END of this step
    LGKM_WAIT0();                                   // reads(st) have returned
    if (dflags & 16) {                       // timing experiment: barriers + X DMA only
      if (st + C2_DEPTH < nst) issue_x(st + C2_DEPTH);
      return;
    }
    if (st + 1 < nst) issue_reads_p1(st + 1, nxt);  // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 f0 = TR ? frag2(cur, ks, 0) : cur.f4[ks][0];
      const u32x4 f1 = TR ? frag2(cur, ks, 1) : cur.f4[ks][1];
      hacc[0] = mfma32(as_bf16x8(f0), as_bf16x8(cur.x[ks]), hacc[0]);
      hacc[1] = mfma32(as_bf16x8(f1), as_bf16x8(cur.x[ks]), hacc[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (st + C2_DEPTH < nst) issue_x(st + C2_DEPTH);  // stage st's reads returned before its MFMAs were issued
  };
  stamp(1);
  {
    Frags FA, FB;
    wait_stage(0, C2_DEPTH - 1);
    if (!(dflags & 16)) issue_reads_p1(0, FA);
    int st = 0;
#pragma unroll 1
    for (; st + 1 < nst; st += 2) {
      step1(st, FA, FB);
      step1(st + 1, FB, FA);
    }
    if (st < nst) step1(st, FA, FB);
  }

  stamp(2);
  if (tl_buf && lane == 0) tl_buf[8] = t_wait, tl_buf[9] = t_bar;
  const int dbg = dflags;
  // ================================================================== hand-off (registers only)
  // scale, mask rank rows >= r (overlap garbage / zeros), round to bf16: hf[s] is the phase-2 B operand
  // of k-step s; the same values go to the saved copy [M, 64] as 8-byte row pieces.
  const int64_t tok = tok0 + li;
  u32x4 hf[4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    float hv[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
      hv[reg] = r < rb ? hacc[rt][reg] * p.scale : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
      hf[2 * rt + a] = (u32x4){pack_bf16x2(hv[8 * a + 0], hv[8 * a + 1]), pack_bf16x2(hv[8 * a + 2], hv[8 * a + 3]),
                               pack_bf16x2(hv[8 * a + 4], hv[8 * a + 5]), pack_bf16x2(hv[8 * a + 6], hv[8 * a + 7])};
    if (p.Hsave && !(dbg & 2) && tok < p.M) {
      bf16_t* Hs = (bf16_t*)p.Hsave + tok * 64 + rt * 32 + 4 * lh;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        u32x2 v = {pack_bf16x2(hv[4 * rq + 0], hv[4 * rq + 1]), pack_bf16x2(hv[4 * rq + 2], hv[4 * rq + 3])};
        // column 63 <- 1.0 when free (dbias trick of the skinny-TN kernel): rt = 1, rq = 3, lh = 1, element 3
        if (rt == 1 && rq == 3 && lh == 1 && rb < 64) v[1] = (v[1] & 0xffffu) | 0x3F800000u;
        *(u32x2*)(Hs + 8 * rq) = v;
      }
    }
  }

  stamp(3);
  // ================================================================== phase 2: Y^T = F2^T . H^T
  // Epilogue: Y^T has one token per lane, so a direct store writes 8-byte pieces of 32 different rows
  // (measured: 2x the time of full-row stores).  The slice is therefore transposed through a
  // wave-private fp32 LDS tile ([32 tok][64 col], 16-byte chunks XOR-swizzled by the row) and written
  // as 16-byte row segments by the NEXT iteration, so the LDS round trip overlaps the next slice's
  // factor reads and MFMAs.  The X ring of this wave is free by now: two tiles of 8 KiB.
  bf16_t* Y = (bf16_t*)p.Y;
  const bf16_t* bias = (const bf16_t*)p.bias;
  const int ksteps = (rb + 15) / 16;
  auto tile_addr = [&](int buf, int row, int chunk) {   // chunk = 16-byte (4 fp32) index 0..15
    return ring_a + (uint32_t)(buf * 8192 + row * 256 + ((chunk ^ (row & 15)) * 16));
  };
  auto flush = [&](int sl_prev) {   // write slice sl_prev from tile (sl_prev & 1)
    const int buf = sl_prev & 1;
    u32x4 v0[4], v1[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 8 + (lane >> 3), c8 = lane & 7;   // row, 8-column group
      DS_READ_B128(v0[pass], tile_addr(buf, r, 2 * c8), 0);
      DS_READ_B128(v1[pass], tile_addr(buf, r, 2 * c8 + 1), 0);
    }
    LGKM_WAIT0();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 8 + (lane >> 3), c8 = lane & 7;
      const int64_t tk = tok0 + r;
      const int col = sl_prev * 64 + c8 * 8;
      if (tk < p.M && col < D2 && !(dbg & 4)) {
        float v[8];
        const float* f0 = (const float*)&v0[pass];
        const float* f1 = (const float*)&v1[pass];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f0[e], v[4 + e] = f1[e];
        bf16_t* dst = Y + tk * p.ldy + col;
        if (p.beta != 0.f) {
          const u32x4 old = *(const u32x4*)dst;
          const bf16_t* o = (const bf16_t*)&old;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += p.beta * (float)o[e];
        }
        if (bias) {
          const u32x4 bv = *(const u32x4*)(bias + col);
          const bf16_t* bb = (const bf16_t*)&bv;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)bb[e];
        }
        *(u32x4*)dst = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                pack_bf16x2(v[6], v[7])};
      }
    }
  };
  // step2(sl): [barrier(chunk sl+1)] -> wait reads(sl) -> issue reads(sl+1) -> MFMA(sl) -> flush(sl-1) -> park(sl)
  auto step2 = [&](int sl, Frags& cur, Frags& nxt) {
    if (sl + 1 < nsl) raw_barrier();                 // factor chunk nst + sl + 1 is in its slot
    LGKM_WAIT0();                                    // reads(sl) returned; tile writes of slice sl-1 done
    if (dbg & 16) return;
    if (sl + 1 < nsl) issue_reads_p2(sl + 1, nxt);
    f32x16 yacc[2];   // Y^T tiles: lane = token, registers = output columns of tile nt
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) yacc[a][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < ksteps) {
        yacc[0] = mfma32(as_bf16x8(frag2(cur, ks, 0)), as_bf16x8(hf[ks]), yacc[0]);
        yacc[1] = mfma32(as_bf16x8(frag2(cur, ks, 1)), as_bf16x8(hf[ks]), yacc[1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (sl > 0) flush(sl - 1);   // previous slice: LDS -> global while this slice's MFMAs drain
    // park this slice: register quad rq of tile nt holds columns nt*32 + 8*rq + 4*lh .. +3 of token li
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int chunk = nt * 8 + 2 * rq + lh;
        f32x4 v = {yacc[nt][4 * rq + 0], yacc[nt][4 * rq + 1], yacc[nt][4 * rq + 2], yacc[nt][4 * rq + 3]};
        *(f32x4*)(ring + (sl & 1) * 8192 + li * 256 + ((chunk ^ (li & 15)) * 16)) = v;
      }
    __builtin_amdgcn_wave_barrier();
  };
  if (nsl > 0) {
    Frags FA, FB;
    raw_barrier();                                   // factor chunk nst is in its slot
    if (!(dbg & 16)) issue_reads_p2(0, FA);
    int sl = 0;
#pragma unroll 1
    for (; sl + 1 < nsl; sl += 2) {
      step2(sl, FA, FB);
      step2(sl + 1, FB, FA);
    }
    if (sl < nsl) step2(sl, FA, FB);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(dbg & 16)) flush(nsl - 1);
  }
  stamp(4);
  if (tl_buf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
  }
}

// =================================================================================================
bool chain2_supported(const ChainParams& p, int dtype) {
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (dtype != SOW_BF16 || p.ra != 0 || p.rb <= 0 || p.rb > 64 || (p.rb & 1)) return false;
  if (p.D1 % 8 || p.D2 % 8 || p.ldx % 8 || p.ldy % 8) return false;
  if (!a16(p.X) || !a16(p.Y) || (p.bias && !a16(p.bias)) || (p.Hsave && !a16(p.Hsave))) return false;
  if (p.M < 4096) return false;  // short inputs: the 64-row generic kernel fills the chip better
  return true;
}

int launch_chain2(const ChainParams& p_in, bool bwd, hipStream_t stream) {
  ChainParams p = p_in;
  p.fast_factors = 0;
#if SOW_CHAIN2_DIAG
  {
    const char* e = getenv("SOW_AMD_CHAIN2_DEBUG");
    p.fast_factors = e ? atoi(e) : 0;
    const char* b = getenv("SOW_AMD_CHAIN2_DBGBUF");
    if ((p.fast_factors & 32) && b) p.F1a = (const void*)strtoull(b, nullptr, 0); else p.fast_factors &= ~32;
  }
#endif
  // B is DMA'd in aligned 16-byte pieces along its rows; A must be contiguous [rows, r] and 4-byte aligned
  const void* Bp = bwd ? p.F1b : p.F2b;
  const int64_t ldB = bwd ? p.ldf1b : p.ldf2b;
  const void* Ap = bwd ? p.F2b : p.F1b;
  const int64_t ldA = bwd ? p.ldf2b : p.ldf1b;
  if ((reinterpret_cast<uintptr_t>(Bp) & 15) || ldB % 8 || (reinterpret_cast<uintptr_t>(Ap) & 3) || ldA != p.rb)
    return SOW_ERR_ALIGN;
  const int grid = ceil_div(p.M, C2_BM);
  if (bwd) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)chain2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
      attr_set = true;
    }
    hipLaunchKernelGGL(chain2_kernel<true>, dim3(grid), dim3(C2_THREADS), C2_LDS, stream, p);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)chain2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
      attr_set = true;
    }
    hipLaunchKernelGGL(chain2_kernel<false>, dim3(grid), dim3(C2_THREADS), C2_LDS, stream, p);
  }
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
