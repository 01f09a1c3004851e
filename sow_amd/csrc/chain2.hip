// Fused low-rank chain, bf16 streaming version (gfx950):  Y = beta*Y + (scale * X.F1).F2 + bias
//
// Same contract as chain.hip (reference tn_gradient/layer/sow.py:107-126 forward, and its autograd
// backward with F1 = B^T, F2 = A^T), rebuilt around the CDNA4 features that matter for an HBM-bound
// kernel:
//   * LDS-DMA (`global_load_lds_dwordx4`) for EVERYTHING that is read: X streams through per-token-group
//     rings of [32 tok x 64 k] stages (no VGPR staging, counted s_waitcnt vmcnt); two loader waves
//     stream the factors as 64-row chunks into a 6-slot LDS ring, 5 chunks ahead of the consumers.
//   * `ds_read_b64_tr_b16`: the factors stay in their storage layout (A is [d_in, r], B is [r, d_out]);
//     in the forward direction both have the contraction index as their ROW index and are read
//     transposed; in the backward direction both are k-contiguous and read with ds_read_b128 / b64.
//   * 16-byte loads need only 4-byte alignment on gfx950 (tools/probe2.hip): the 100-byte rows of a
//     rank-50 A are DMA'd as 128-byte rows whose tail is the head of the next row ("padding by
//     overlap"); the garbage lands in rank columns >= r, which always meet an explicit zero (rows >= r of
//     B come from a zero page, H columns >= r are masked), so it never reaches a result.  The last row of
//     A, whose tail would cross the end of the buffer, is rewritten by the loader from a guarded load.
//   * Both products are computed TRANSPOSED (H^T = F1^T X^T, Y^T = F2^T H^T) so that the token index
//     sits on the MFMA lane: H^T's accumulator registers are directly the B operand of phase 2 (k order
//     permuted: element j of lane-half h is rank 16s + 8(j>>2) + 4h + (j&3), CDNA guide "accumulator
//     tile as the next MFMA's operand"); H never goes through LDS or HBM except as the saved copy.
//
// Measured with in-kernel stamps: a wave that does everything for its 32 tokens is bound by its own
// instruction issue (each LDS-DMA / global store occupies the issuing wave ~100-180 cycles), not by HBM.
// The work of one 32-token group is therefore split over TWO compute waves: in phase 1 each takes
// half of every stage's DMA rows and one 32-rank tile of H^T over the whole K range (final accumulators; the two tiles
// are exchanged as bf16 through LDS at the hand-off);
// in phase 2 each takes one 32-column tile of every 64-column slice and half of the store rows.
// Workgroup = 64 tokens = 6 waves: compute waves 0-3 (token group = w & 1, half = w >> 1) and loader
// waves 4-5; 80 KiB of LDS, two workgroups per CU (12 waves, 3 per SIMD).  One raw s_barrier per
// chunk hands chunk c to the consumers and frees the slot of chunk c-1 for the loaders (chunk c+5).
// All LDS reads of the compute waves are inline asm: for a compiler-visible LDS read hipcc emits
// `s_waitcnt vmcnt(0)` while LDS-DMA is outstanding, which would drain the rings every step.
//
// LDS images are 64 (or 32) rows x 128 B; 16-byte chunk c of a row sits at physical chunk
//   c ^ ((row >> 1) & 7)           for ds_read_b128 / b64 consumers (X stage, backward factors)
//   c ^ (((row >> 1) & 1) << 2)    for transposed-read consumers (forward factors)
// DMA writes LDS lane-linearly, so the XOR is applied to the per-lane SOURCE address.
#include "kernels.hpp"
#include "lds_dma.hpp"

namespace sow {

// In-kernel timeline (debug builds only: `make STAMPS=1` -> libsow_amd_stamps.so, tools/chain_stamps.py): wave 0 of every
// workgroup writes s_memrealtime (100 MHz) at the phase boundaries of every block it runs into grp.stamps[block][16]
// (slots 0-6); slots 8-12 hold accumulated wait times (10-ns ticks): 8 loader wave 4 in its counted DMA wait, 9 the same wave at the chunk barrier, 10 / 11 compute wave 0 in its X wait / at the stage barrier of
// phase 1, 12 compute wave 0 at the slice barrier of phase 2.
#ifdef SOW_STAMPS
#define C2_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (stamps && w == 0 && lane == 0) stamps[i] = __builtin_amdgcn_s_memrealtime();                \
  } while (0)
#else
#define C2_STAMP(i) \
  do {              \
  } while (0)
#endif
#ifdef SOW_STAMPS
#define C2_TICK() __builtin_amdgcn_s_memrealtime()
#define C2_ACC(var, t0) var += __builtin_amdgcn_s_memrealtime() - (t0)
#define C2_PUT(cond, i, var)                     \
  do {                                           \
    if (stamps && (cond) && lane == 0) stamps[i] = (var); \
  } while (0)
#else
#define C2_TICK() 0ull
#define C2_ACC(var, t0) (void)(t0)
#define C2_PUT(cond, i, var) \
  do {                       \
  } while (0)
#endif

constexpr int C2_NTG = 2;             // token groups (32 tokens) per workgroup
constexpr int C2_NCW = 2 * C2_NTG;    // compute waves: (token group, half)
constexpr int C2_NLW = 2;             // loader waves
constexpr int C2_BM = 32 * C2_NTG;    // tokens per workgroup
// Ring split of the 80 KiB (A/B builds: make VARIANT=... DEFS="-DC2_DEPTH_X=6 -DC2_NSLOT_X=4 -DC2_AHEAD_X=3"; slots + depth = 10).
// The factor chunks come from L2, but under a saturated HBM stream an L2 hit takes ~2 us, and a late chunk stalls all six
// waves at the stage barrier (tools/chain_stamps.py: the loader waves spend 1.2 us of a 14-us block waiting for their DMA;
// with the factor DMA compiled out the kernel is 12 % faster).  Round 2 moved two X slots to the factor ring: 4 X slots
// (3 in flight) + 6 chunk slots (5 ahead) against 6 + 4 (3 ahead): step 3.92 -> 3.84 ms on one box (5 + 5: 3.87);
// q + k + v forward 52.7 -> 50.1 us, the K = 1376 forward 34.9 -> 35.8 us.
#ifndef C2_DEPTH_X
#define C2_DEPTH_X 4
#define C2_NSLOT_X 6
#define C2_AHEAD_X 5
#endif
constexpr int C2_DEPTH = C2_DEPTH_X;   // X stage slots per token group (DEPTH - 1 in flight)
constexpr int C2_STAGE = 4096;        // [32 tok][64 k] bf16
constexpr int C2_NSLOT = C2_NSLOT_X;   // factor chunk slots
constexpr int C2_AHEAD = C2_AHEAD_X;   // chunks the loaders run ahead: the slot of chunk c + AHEAD held chunk c - 1, drained before barrier c
#ifdef C2_EXPERIMENT_SHALLOW_X   // timing experiment: 3 X slots hold the bf16 park tiles but not the fp32 ones (bias / beta launches break)
static_assert(C2_NSLOT == C2_AHEAD + 1 && C2_NSLOT + C2_DEPTH == 10 && C2_DEPTH >= 3 && C2_DEPTH <= 6, "80 KiB");
#else
static_assert(C2_NSLOT == C2_AHEAD + 1 && C2_NSLOT + C2_DEPTH == 10 && C2_DEPTH >= 4 && C2_DEPTH <= 6,
              "80 KiB: 8 KiB per chunk slot + 2 x 4 KiB per X slot; two 8-KiB fp32 park tiles per token group must fit its X ring");
#endif
constexpr int C2_FSLOT = 8192;        // [64][64] bf16
constexpr int C2_LPW = 8 / C2_NLW;    // 1-KiB DMA instructions per loader wave per chunk
constexpr int C2_RING0 = C2_NSLOT * C2_FSLOT;
constexpr int C2_RING = C2_DEPTH * C2_STAGE;  // 16 KiB per token group
constexpr int C2_LDS = C2_RING0 + C2_NTG * C2_RING;   // 80 KiB: two workgroups per CU
constexpr int C2_THREADS = 64 * (C2_NCW + C2_NLW);
constexpr int C2_RESIDENT = 512;      // resident workgroups of a persistent grid (2 per CU x 256 CUs)


template <bool TR> __device__ __forceinline__ int img_chunk(int row, int c) {
  return TR ? (c ^ (((row >> 1) & 1) << 2)) : (c ^ ((row >> 1) & 7));
}

// =================================================================================================
// BWD = false: forward  (F1 = A [D1, r] rows = k, F2 = B [r, D2] rows = k  -> transposed reads)
// BWD = true : backward (F1 = B [r, D1] rows = rank, F2 = A [D2, r] rows = n -> b128 / b64 reads)
//
// Grouped launch: the grid is the concatenation of the grids of up to C2_MAXG independent layers (same direction; shapes
// may differ) -- e.g. the q / k / v projections of an attention block, or gate / up of an MLP.  A launch costs ~8 us of
// ramp, first-DMA round trip, hand-off and write drain whatever its size (tools/chain_sweep.py: t = 8.6 us + bytes /
// 5.1 TB/s), so three 512-workgroup layers in one 1536-workgroup grid pay it once; later rounds start while earlier
// workgroups drain.  Every workgroup runs exactly the single-layer code on its own layer's parameter block, so the
// results are bit-identical to separate launches.
template <bool BWD, bool P16>
__device__ __forceinline__ void chain2_block(const ChainParams& p, const int bid, char* smem, const int t, const int lane, const int w,
                                             uint64_t* stamps) {
  constexpr bool TR = !BWD;
  C2_STAMP(0);
  // short-T split (kernels.hpp): workgroup = (token block, split); a split owns a range of phase-1 stages OR of
  // phase-2 slices.  Without a split every workgroup owns all of both.
  const int tb = p.ntb > 0 ? bid % p.ntb : bid;
  const int split = p.ntb > 0 ? bid / p.ntb : 0;
  const int64_t m0 = (int64_t)tb * C2_BM;
  const int D1 = p.D1, D2 = p.D2, rb = p.rb;
  const int nst_all = (D1 + 63) / 64, nsl_all = (D2 + 63) / 64;
  const int st0 = p.st_per > 0 ? split * p.st_per : 0;   // first phase-1 stage of this workgroup
  const int sl0 = p.sl_per > 0 ? split * p.sl_per : 0;   // first phase-2 slice
  const int nst = p.sl_per > 0 ? 0 : (p.st_per > 0 ? (nst_all - st0 < p.st_per ? nst_all - st0 : p.st_per) : nst_all);
  const int nsl = p.st_per > 0 ? 0 : (p.sl_per > 0 ? (nsl_all - sl0 < p.sl_per ? nsl_all - sl0 : p.sl_per) : nsl_all);
  const int total = nst + nsl;
  const bf16_t* Amat = (const bf16_t*)(BWD ? p.F2b : p.F1b);   // [rows_a, rb] contiguous
  const bf16_t* Bmat = (const bf16_t*)(BWD ? p.F1b : p.F2b);   // [rb, cols_b], ld = ldb
  const int64_t ldb = BWD ? p.ldf1b : p.ldf2b;
  const int rows_a = BWD ? D2 : D1, cols_b = BWD ? D1 : D2;
  const char* zp = zero_page_for(lane);

  if (w >= C2_NCW) {
    // ------------------------------------------------------------------ loader waves
    // phase 1: the loaders' chunk DMA goes ahead of the compute waves' instruction streams (a late chunk stalls everybody at the
    // stage barrier); phase 2: back to normal, the stores of the compute waves come first.  K = 1376 forward 36.0 -> 33.4 us,
    // q + k + v 50.4 -> 49.2 us; with the priority kept through phase 2 the wide-output launches lose 1-3 us.
    __builtin_amdgcn_s_setprio(3);
    const int lw = w - C2_NCW;
    const char* a_end = (const char*)(Amat + (int64_t)rows_a * rb);
    // the last row of A, kept in a register for the fix-up by the loader wave that DMAs that row
    // (rows 32*lw .. 32*lw+31 of a chunk belong to loader wave lw, so its own counted wait orders the
    // fix-up after its DMA)
    const bool own_last = rows_a > 0 && lw == ((rows_a - 1) & 63) / (8 * C2_LPW);
    uint32_t last_row_dw = 0u;
    if (own_last && lane < 32 && 2 * lane < rb) last_row_dw = *((const uint32_t*)(Amat + (int64_t)(rows_a - 1) * rb) + lane);
    asm volatile("" : "+v"(last_row_dw));  // consume now: the compiler's wait for this load lands here, not mid-pipeline
    auto chunk_is_a = [&](int c) { return BWD ? (c >= nst) : (c < nst); };
    // per-lane source pointers of chunk 0, advanced by a constant per chunk
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
#ifdef C2_EXPERIMENT_NO_FACTOR_DMA   // timing experiment only (wrong results): what the factor re-reads from L2 cost
      if (c >= C2_NSLOT) return;
#endif
#ifdef C2_EXPERIMENT_HALF_FACTOR_DMA   // timing experiment only (wrong results): every other block re-uses stale chunks
      if (bid & 1) return;
#endif
      char* slot = smem + (c % C2_NSLOT) * C2_FSLOT;
      const int ci = c < nst ? st0 + c : sl0 + (c - nst);   // chunk index inside its matrix
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
    uint64_t tw_dma = 0, tw_bar = 0;
    for (int c = 0; c < total; ++c) {
      const int newer = (total - 1 - c) < (C2_AHEAD - 1) ? (total - 1 - c) : (C2_AHEAD - 1);
      const uint64_t tk0 = C2_TICK();
      wait_groups<C2_LPW>(newer);
      C2_ACC(tw_dma, tk0);
      // fix-up: rewrite the last row of A (its DMA pieces past the end of the buffer were zero-filled)
      if (chunk_is_a(c)) {
        const int base = (c < nst ? st0 + c : sl0 + (c - nst)) * 64;
        const int lr = rows_a - 1 - base;
        if (own_last && lr >= 0 && lr < 64 && lane < 32) {
          const int cc = lane >> 2;
          *(uint32_t*)(smem + (c % C2_NSLOT) * C2_FSLOT + lr * 128 + img_chunk<TR>(lr, cc) * 16 + (lane & 3) * 4) = last_row_dw;
        }
      }
      if (c == nst) __builtin_amdgcn_s_setprio(0);
      if (c == nst) raw_barrier();   // the hand-off barrier of the compute waves (exchange of the rank tiles)
      const uint64_t tk1 = C2_TICK();
      raw_barrier();   // chunk c visible to the consumers; they have finished chunk c-1
      C2_ACC(tw_bar, tk1);
      if (c + C2_AHEAD < total) issue(c + C2_AHEAD);   // its slot held chunk c-1: every consumer finished it before barrier c
    }
    C2_PUT(w == C2_NCW, 8, tw_dma);
    C2_PUT(w == C2_NCW, 9, tw_bar);
    if (nsl > 0) {
      raw_barrier();   // matches the compute waves' final "last slice parked" barrier
    } else {           // H-only call (D2 == 0): the hand-off barrier was not met inside the loop
      raw_barrier();
    }
    raw_barrier();     // end of block: the compute waves have read the last parked slice out of the rings
    return;
  }

  // -------------------------------------------------------------------- compute waves
  if (p.pad_dst && bid * 64 < p.pad_rows) {
    // side job (first ceil(rows / 64) workgroups): two 8-column groups per thread, plain guarded loads
    const bf16_t* src = (const bf16_t*)p.pad_src;
    bf16_t* dst = (bf16_t*)p.pad_dst;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = t + 256 * it, row = bid * 64 + (idx >> 3), c0 = (idx & 7) * 8;
      if (row < p.pad_rows) {
        u32x4 v;
        bf16_t* e = (bf16_t*)&v;
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (c0 + j < p.pad_r) ? src[(int64_t)row * p.pad_r + c0 + j] : (bf16_t)0.f;
        *(u32x4*)(dst + (int64_t)row * 64 + c0) = v;
      }
    }
  }
  const int tg = w & 1, hh = w >> 1;   // token group, half (rank tile in phase 1, column tile in phase 2)
  const int li = lane & 31, lh = lane >> 5;
  const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3;  // transposed-read geometry
  const int h2 = g >> 1;
  char* ring = smem + C2_RING0 + tg * C2_RING;
  const uint32_t ring_a = lds_addr(ring);
  const uint32_t slot_a = lds_addr(smem);
  const bf16_t* X = (const bf16_t*)p.X;
  const int64_t tok0 = m0 + 32 * tg;

  // X DMA: this wave issues instructions i = 2*hh, 2*hh+1 (token rows 16*hh .. 16*hh+15) of every stage
  const int drow = lane >> 3, dpc = lane & 7;
  const char* xsrc[2];
  int xstride[2], xlc[2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int row = 8 * (2 * hh + ii) + drow;
    const int lc = dpc ^ ((row >> 1) & 7);
    const int64_t tk = tok0 + row;
    const bool v = tk < p.M;
    xsrc[ii] = v ? (const char*)(X + tk * p.ldx + lc * 8) : zp;
    xstride[ii] = v ? 128 : 0;
    xlc[ii] = lc;
  }
  const bool x_ragged = (D1 & 63) != 0;
  auto issue_x = [&](int st) {
    char* dst = ring + (st % C2_DEPTH) * C2_STAGE;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const char* qq = xsrc[ii] + (st0 + st) * xstride[ii];
      if (x_ragged && (st0 + st) * 64 + xlc[ii] * 8 >= D1) qq = zp;
      if (p.nt_load) dma16_nt((const void*)qq, dst + (2 * hh + ii) * 1024);
      else dma16((const void*)qq, dst + (2 * hh + ii) * 1024);
    }
  };

  // per-lane LDS offsets
  const uint32_t xoff = (uint32_t)(li * 128);   // this lane's row in an X stage
  const int xsw = (li >> 1) & 7;                // b128 row swizzle
  //   phase 1 (natural k):  TR rows 16ks + 8h + q (+4)      | B128 chunk 2ks + h of row (tile*32 + li)
  //   phase 2 (permuted k): TR rows 16ks + 4h + q (+8)      | two B64 at k = 16ks + 4h and 16ks + 8 + 4h
  uint32_t foff1[1], foff2;   // phase 1: rank tile hh of the chunk image
  if constexpr (TR) {
    const int col = hh * 32 + 16 * (g & 1) + 4 * pp;
    const int r1 = 8 * h2 + q;
    foff1[0] = (uint32_t)(r1 * 128 + img_chunk<true>(r1, col >> 3) * 16 + (col & 7) * 2);
  } else {
    foff1[0] = (uint32_t)((hh * 32 + li) * 128);
  }
  if constexpr (TR) {
    const int col = hh * 32 + 16 * (g & 1) + 4 * pp;
    const int r2 = 4 * h2 + q;
    foff2 = (uint32_t)(r2 * 128 + img_chunk<true>(r2, col >> 3) * 16 + (col & 7) * 2);
  } else {
    foff2 = (uint32_t)((hh * 32 + li) * 128 + 8 * lh);
  }

  f32x16 hacc;   // H^T tile hh (ranks 32 hh .. 32 hh + 31) over the whole K range: lane = token, registers = rank rows
#pragma unroll
  for (int i = 0; i < 16; ++i) hacc[i] = 0.f;

  // stage s+DEPTH-1 is issued after barrier(s): by then BOTH waves of the token group are done with stage s-1
  const int pre = nst < (C2_DEPTH - 1) ? nst : (C2_DEPTH - 1);
  for (int st = 0; st < pre; ++st) issue_x(st);

  // ================================================================== phase 1: H^T = F1^T . X^T (rank tile hh)
  uint64_t tw_x = 0, tw_b1 = 0, tw_b2 = 0;
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    // stages issued after `st` at this point: st+1 .. st+DEPTH-2
    const int newer = (nst - 1 - st) < (C2_DEPTH - 2) ? (nst - 1 - st) : (C2_DEPTH - 2);
    const uint64_t tk0 = C2_TICK();
    wait_groups<2>(newer);   // this wave's half of X stage `st` has landed
    C2_ACC(tw_x, tk0);
    const uint64_t tk1 = C2_TICK();
    raw_barrier();           // ... and so have the partner's half and factor chunk `st`
    C2_ACC(tw_b1, tk1);
    if (st == 0) C2_STAMP(1);
    if (st + C2_DEPTH - 1 < nst) issue_x(st + C2_DEPTH - 1);   // into the slot of stage st-1
    const uint32_t xs = ring_a + (uint32_t)((st % C2_DEPTH) * C2_STAGE) + xoff;
    const uint32_t fs = slot_a + (uint32_t)((st % C2_NSLOT) * C2_FSLOT);
    // the two waves of a token group split the RANKS (tile hh each), not K: every wave contracts the whole stage, so its
    // accumulator is final and the hand-off is an exchange of 2 KiB of bf16 instead of a sum of 8 KiB of fp32 partials
    u32x4 xf[4], ff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) DS_READ_B128(xf[ks], xs + (uint32_t)(((2 * ks + lh) ^ xsw) * 16), 0);
    if constexpr (TR) {
      u32x2 bl[4], bh[4];
      const uint32_t b0 = fs + foff1[0];
      DS_READ_TR(bl[0], b0, 0);
      DS_READ_TR(bh[0], b0, 512);
      DS_READ_TR(bl[1], b0, 2048);
      DS_READ_TR(bh[1], b0, 2048 + 512);
      DS_READ_TR(bl[2], b0, 4096);
      DS_READ_TR(bh[2], b0, 4096 + 512);
      DS_READ_TR(bl[3], b0, 6144);
      DS_READ_TR(bh[3], b0, 6144 + 512);
      LGKM_WAIT0();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ff[ks] = join2(bl[ks], bh[ks]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) DS_READ_B128(ff[ks], fs + foff1[0] + (uint32_t)(((2 * ks + lh) ^ xsw) * 16), 0);
      LGKM_WAIT0();
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) hacc = mfma32(as_bf16x8(ff[ks]), as_bf16x8(xf[ks]), hacc);
    __builtin_amdgcn_sched_barrier(0);
  }

  C2_STAMP(2);
  // ================================================================== hand-off: exchange the two rank tiles
  // Each wave scales, masks and rounds its own 32 ranks (k-steps 2 hh, 2 hh + 1 of phase 2), parks them in the X slot that
  // follows the last stage (free: the partner can only still be reading the slot of stage nst - 1), and after ONE barrier
  // reads the partner's two k-steps.
  u32x4 hf[4];
  const int64_t tok = tok0 + li;
  if (p.Hload) {
    // phase-2-only workgroup: H comes from memory.  hf[s] of lane (li, lh) = ranks 16s + 4lh + (0..3) and
    // 16s + 8 + 4lh + (0..3) of token li; the 1.0 of column 63 (dbias trick) must not reach the product.
    raw_barrier();
    const bf16_t* Hl = (const bf16_t*)p.Hload + tok * 64 + 4 * lh;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      u32x2 lo = {0u, 0u}, hi = {0u, 0u};
      if (tok < p.M) {
        lo = *(const u32x2*)(Hl + 16 * s4);
        hi = *(const u32x2*)(Hl + 16 * s4 + 8);
      }
      if (s4 == 3 && lh == 1 && rb < 64) hi[1] &= 0xffffu;
      hf[s4] = join2(lo, hi);
    }
  } else {
    u32x4 own[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if (p.Hpartial) {
      // phase-1 slab of a short-T split: the raw fp32 sums of this K range, one 32-rank tile per wave
      if (tok < p.M) {
        float* Hp = p.Hpartial + ((int64_t)split * p.M + tok) * 64 + hh * 32 + 4 * lh;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
          *(f32x4*)(Hp + 8 * rq) = (f32x4){hacc[4 * rq + 0], hacc[4 * rq + 1], hacc[4 * rq + 2], hacc[4 * rq + 3]};
      }
    } else {
      // scale, mask rank rows >= r (overlap garbage / zeros), round to bf16; the saved copy [M, 64] (scaled live columns,
      // zeros, and 1.0 in column 63 when free -- the dbias trick of the skinny-TN kernel) is written as 8-byte row pieces
      float hv[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = hh * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        hv[reg] = r < rb ? hacc[reg] * p.scale : 0.f;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
        own[a] = (u32x4){pack_bf16x2(hv[8 * a + 0], hv[8 * a + 1]), pack_bf16x2(hv[8 * a + 2], hv[8 * a + 3]),
                         pack_bf16x2(hv[8 * a + 4], hv[8 * a + 5]), pack_bf16x2(hv[8 * a + 6], hv[8 * a + 7])};
      if (p.Hsave && tok < p.M) {
        bf16_t* Hs = (bf16_t*)p.Hsave + tok * 64 + hh * 32 + 4 * lh;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          u32x2 v = {pack_bf16x2(hv[4 * rq + 0], hv[4 * rq + 1]), pack_bf16x2(hv[4 * rq + 2], hv[4 * rq + 3])};
          if (hh == 1 && rq == 3 && lh == 1 && rb < 64) v[1] = (v[1] & 0xffffu) | 0x3F800000u;  // column 63 <- 1.0
          *(u32x2*)(Hs + 8 * rq) = v;
        }
      }
    }
    char* xch = ring + (nst % C2_DEPTH) * C2_STAGE;   // [half][k-step][lane] x 16 bytes
    *(u32x4*)(xch + hh * 2048 + lane * 16) = own[0];
    *(u32x4*)(xch + hh * 2048 + 1024 + lane * 16) = own[1];
    raw_barrier();   // both rank tiles of the token group are parked
    u32x4 oth[2];
    const uint32_t pa = ring_a + (uint32_t)((nst % C2_DEPTH) * C2_STAGE + (hh ^ 1) * 2048 + lane * 16);
    DS_READ_B128(oth[0], pa, 0);
    DS_READ_B128(oth[1], pa, 1024);
    LGKM_WAIT0();
    hf[0] = hh ? oth[0] : own[0], hf[1] = hh ? oth[1] : own[1];   // (hh is wave-uniform; no dynamic register indexing)
    hf[2] = hh ? own[0] : oth[0], hf[3] = hh ? own[1] : oth[1];
  }
  C2_STAMP(3);
  // ================================================================== phase 2: Y^T = F2^T . H^T (column tile hh)
  // Epilogue: Y^T has one token per lane, so a direct store would write 8-byte pieces of 32 different
  // rows (measured: 2x the time of full-row stores).  Each slice is transposed through a per-token-group
  // fp32 LDS tile ([32 tok][64 col], 16-byte chunks XOR-swizzled by the row; this wave fills its 32
  // columns) and written one step later as 16-byte row segments (this wave stores rows 16*hh..+15).
  bf16_t* Y = (bf16_t*)p.Y;
  const bf16_t* bias = (const bf16_t*)p.bias;
  const int ksteps = (rb + 15) / 16;
  auto tile_addr = [&](int buf, int row, int chunk) {   // chunk = 16-byte (4 fp32) index 0..15
    return ring_a + (uint32_t)(buf * 8192 + row * 256 + ((chunk ^ (row & 15)) * 16));
  };
  const bool pairf = p.pair_flush != 0 && C2_RING >= 3 * 8192;   // park tiles rotate through 3 buffers and two slices are stored at a time
  auto flush = [&](int sl_prev) {   // store rows 16*hh .. 16*hh+15 of slice sl_prev from its park tile
    const int buf = pairf ? sl_prev % 3 : (sl_prev & 1);
    u32x4 v0[2], v1[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = 16 * hh + pass * 8 + (lane >> 3), c8 = lane & 7;   // row, 8-column group
      DS_READ_B128(v0[pass], tile_addr(buf, r, 2 * c8), 0);
      DS_READ_B128(v1[pass], tile_addr(buf, r, 2 * c8 + 1), 0);
    }
    LGKM_WAIT0();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = 16 * hh + pass * 8 + (lane >> 3), c8 = lane & 7;
      const int64_t tk = tok0 + r;
      const int col = (sl0 + sl_prev) * 64 + c8 * 8;
      if (tk < p.M && col < D2) {
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
        // streaming (non-temporal) store: Y is not read again by this kernel.  Without the hint up to 32 MB of dirty lines sit
        // in the eight L2s when the last workgroup ends and are written back before the next kernel may start (the XCDs'
        // L2s are not coherent with each other): 512 -> 512 forward 23.0 -> 19.7 us per launch, step 4.48 -> 4.23 ms.
        // ("sc0 sc1" write-through: 22.3 us; "sc0 sc1 nt": 19.2 us -- no better than nt alone.)
        const u32x4 ov = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if (p.nt_store) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(ov) : "memory");
        else *(u32x4*)dst = ov;
      }
    }
  };
  // pair flush: slices sl_a and sl_a + 1 as ONE 256-byte piece per row (16 lanes x 16 B) -- HBM absorbs the store stream
  // of the wide layers (2752-byte pitch) faster in 256-byte than in 128-byte pieces (tools/probe7.hip: +11 % for a pure
  // write; here gate + up forward 78.3 -> 73.0 us, down backward 39.2 -> 35.8 us, step 4.25 -> 4.18 ms; NO_PAIR_FLUSH switch)
  auto flush_pair = [&](int sl_a) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      u32x4 v0[2], v1[2];
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int r = 16 * hh + (2 * half + ps) * 4 + (lane >> 4), c16 = lane & 15;
        const int buf = (sl_a + (c16 >> 3)) % 3, c8 = c16 & 7;
        DS_READ_B128(v0[ps], tile_addr(buf, r, 2 * c8), 0);
        DS_READ_B128(v1[ps], tile_addr(buf, r, 2 * c8 + 1), 0);
      }
      LGKM_WAIT0();
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int r = 16 * hh + (2 * half + ps) * 4 + (lane >> 4), c16 = lane & 15;
        const int64_t tk = tok0 + r;
        const int col = (sl0 + sl_a) * 64 + c16 * 8;
        if (tk < p.M && col < D2) {
          float v[8];
          const float* f0 = (const float*)&v0[ps];
          const float* f1 = (const float*)&v1[ps];
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
          const u32x4 ov = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
          if (p.nt_store) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(ov) : "memory");
          else *(u32x4*)dst = ov;
        }
      }
    }
  };
  // bf16 park tiles ([32 tok][64 col] bf16, 4 KiB, three of them in ring slots 3..5): with no bias and beta = 0 the value
  // stored is exactly bf16(acc), so the rounding can be done before the transposition -- half the LDS traffic of the
  // epilogue, one ds_read_b128 per 16-byte store, and ring slots 0..2 stay free during phase 2
  constexpr bool park16 = P16;   // host-selected instantiation: every layer of the launch has pair_flush, beta = 0 and no bias
  auto tile16_addr = [&](int buf, int row, int chunk) {   // chunk = 16-byte (8 bf16) index 0..7
    return ring_a + (uint32_t)((C2_DEPTH - 3) * C2_STAGE + buf * 4096 + row * 128 + ((chunk ^ ((row >> 1) & 7)) * 16));
  };
  auto store16 = [&](int r, int col, u32x4 ov) {
    const int64_t tk = tok0 + r;
    if (tk < p.M && col < D2) {
      bf16_t* dst = Y + tk * p.ldy + col;
      if (p.nt_store) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(ov) : "memory");
      else *(u32x4*)dst = ov;
    }
  };
  auto flush16 = [&](int sl_prev) {   // one slice: rows 16*hh .. 16*hh+15, 128 bytes per row
    u32x4 v[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) DS_READ_B128(v[pass], tile16_addr(sl_prev % 3, 16 * hh + pass * 8 + (lane >> 3), lane & 7), 0);
    LGKM_WAIT0();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) store16(16 * hh + pass * 8 + (lane >> 3), (sl0 + sl_prev) * 64 + (lane & 7) * 8, v[pass]);
  };
  auto flush_pair16 = [&](int sl_a) {   // two slices: 256 bytes per row
    u32x4 v[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int r = 16 * hh + ps * 4 + (lane >> 4), c16 = lane & 15;
      DS_READ_B128(v[ps], tile16_addr((sl_a + (c16 >> 3)) % 3, r, c16 & 7), 0);
    }
    LGKM_WAIT0();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) store16(16 * hh + ps * 4 + (lane >> 4), (sl0 + sl_a) * 64 + (lane & 15) * 8, v[ps]);
  };
#pragma unroll 1
  for (int sl = 0; sl < nsl; ++sl) {
    const uint64_t tk2 = C2_TICK();
    raw_barrier();   // factor chunk nst + sl is in its slot; the partner has parked slice sl-1
    C2_ACC(tw_b2, tk2);
    const uint32_t fs = slot_a + (uint32_t)(((nst + sl) % C2_NSLOT) * C2_FSLOT) + foff2;
    u32x2 bl[4], bh[4];
    if constexpr (TR) {
      DS_READ_TR(bl[0], fs, 0);
      DS_READ_TR(bh[0], fs, 1024);
      DS_READ_TR(bl[1], fs, 2048);
      DS_READ_TR(bh[1], fs, 2048 + 1024);
      DS_READ_TR(bl[2], fs, 4096);
      DS_READ_TR(bh[2], fs, 4096 + 1024);
      DS_READ_TR(bl[3], fs, 6144);
      DS_READ_TR(bh[3], fs, 6144 + 1024);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        DS_READ_B64(bl[ks], fs + (uint32_t)(((2 * ks) ^ xsw) * 16), 0);
        DS_READ_B64(bh[ks], fs + (uint32_t)(((2 * ks + 1) ^ xsw) * 16), 0);
      }
    }
    LGKM_WAIT0();
    f32x16 yacc;   // Y^T tile: lane = token, registers = output columns of tile hh
#pragma unroll
    for (int i = 0; i < 16; ++i) yacc[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (ks < ksteps) yacc = mfma32(as_bf16x8(join2(bl[ks], bh[ks])), as_bf16x8(hf[ks]), yacc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (park16) {
      if (sl >= 2 && !(sl & 1)) flush_pair16(sl - 2);
      // park: register quad rq = columns hh*32 + 8*rq + 4*lh .. +3 of token li = half lh of 16-byte chunk hh*4 + rq
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const u32x2 v = {pack_bf16x2(yacc[4 * rq + 0], yacc[4 * rq + 1]), pack_bf16x2(yacc[4 * rq + 2], yacc[4 * rq + 3])};
        *(u32x2*)(ring + (C2_DEPTH - 3) * C2_STAGE + (sl % 3) * 4096 + li * 128 + (((hh * 4 + rq) ^ ((li >> 1) & 7)) * 16) + lh * 8) = v;
      }
    } else {
    if (pairf) {
      if (sl >= 2 && !(sl & 1)) flush_pair(sl - 2);   // the previous two slices, while this slice's MFMAs drain
    } else if (sl > 0) {
      flush(sl - 1);   // previous slice: LDS -> global while this slice's MFMAs drain
    }
    // park this slice: register quad rq holds columns hh*32 + 8*rq + 4*lh .. +3 of token li
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int chunk = hh * 8 + 2 * rq + lh;
      f32x4 v = {yacc[4 * rq + 0], yacc[4 * rq + 1], yacc[4 * rq + 2], yacc[4 * rq + 3]};
      *(f32x4*)(ring + (pairf ? sl % 3 : (sl & 1)) * 8192 + li * 256 + ((chunk ^ (li & 15)) * 16)) = v;
    }
    }
  }
  C2_STAMP(4);
  if (nsl > 0) {
    raw_barrier();   // the partner has parked the last slice
    if constexpr (park16) {
      if (nsl >= 2 && !(nsl & 1)) flush_pair16(nsl - 2);
      else flush16(nsl - 1);
    } else {
      if (pairf && nsl >= 2 && !(nsl & 1)) flush_pair(nsl - 2);
      else flush(nsl - 1);   // (pair mode, odd count: the pairs before it went out inside the loop)
    }
  }
  C2_STAMP(5);
  raw_barrier();     // end of block: every LDS read of this block has returned -- the next block's DMA may overwrite the rings
  C2_STAMP(6);
  C2_PUT(w == 0, 10, tw_x);
  C2_PUT(w == 0, 11, tw_b1);
  C2_PUT(w == 0, 12, tw_b2);
}

// Grouped, persistent launch: the grid is min(total, C2_RESIDENT) workgroups (two per CU); workgroup g runs token blocks
// g, g + grid, g + 2 grid, ... of the concatenated block list of up to C2_MAXG independent layers (same direction; shapes
// may differ) -- e.g. the q / k / v projections of an attention block, or gate / up of an MLP.  Why: a 64-token
// workgroup that is launched, streams 64 KB in, 64 KB out and exits costs ~20 us per round at 512 -> 512 against
// ~14 us of streaming (tools/chain_sweep.py: t = 8.6 us + bytes / 5.1 TB/s) -- workgroup launch, first-load latency and
// the drain of its stores before the slot is free again are per-workgroup costs that a multi-round grid pays every
// round.  A resident workgroup issues the next block's first loads right after its last stores: the store drain and
// the load latency overlap, and nothing is relaunched.  Every block runs exactly the single-layer code on its own
// layer's parameter block, so the results are bit-identical to separate launches.
template <bool BWD, bool P16> __global__ __launch_bounds__(C2_THREADS, 4) void chain2_kernel(const ChainGroup grp) {   // 4 waves per SIMD: caps the allocation at 128 VGPRs, above which a CU cannot place two of these workgroups
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int total = grp.start[C2_MAXG];
  for (int blk = (int)blockIdx.x; blk < total; blk += (int)gridDim.x) {
    int layer = 0;
#pragma unroll
    for (int i = 1; i < C2_MAXG; ++i)
      if (i < grp.n && blk >= grp.start[i]) layer = i;
    // opaque per iteration: keeps the per-lane address arithmetic of a block inside its iteration -- hoisted out of the
    // loop by LICM it stays live across the whole block (168 VGPRs instead of ~100; above 128 a CU holds one workgroup)
    int tt = t;
    asm volatile("" : "+v"(tt));
    chain2_block<BWD, P16>(grp.p[layer], blk - grp.start[layer], smem, tt, tt & 63, w, grp.stamps ? grp.stamps + (int64_t)blk * 16 : nullptr);
  }
}

// =================================================================================================
bool chain2_supported(const ChainParams& p, int dtype) {
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // rb >= 4: the overlap trick zero-fills every 16-byte piece that would cross the end of A and repairs only
  // the LAST row; with 4-byte rows (rb = 2) the last three rows lose their data (found by tests/test_gpu_fuzz.py)
  if (dtype != SOW_BF16 || p.ra != 0 || p.rb < 4 || p.rb > 64 || (p.rb & 1)) return false;
  if (p.D1 % 8 || p.D2 % 8 || p.ldx % 8 || p.ldy % 8) return false;
  if (!a16(p.X) || !a16(p.Y) || (p.bias && !a16(p.bias)) || (p.Hsave && !a16(p.Hsave))) return false;
  if (p.M < 64) return false;   // (short inputs run T/64 workgroups either way; measured 1.4x faster than the generic kernel at T = 1024)
  return true;
}

// Hsave[t][c] = bf16(scale * sum_s Hpartial[s][t][c]) for c < rb, 0 above, 1.0 in column 63 when rb < 64
template <typename T>
__global__ __launch_bounds__(256) void h_reduce_kernel(const float* __restrict__ Hp, int nsplit, T* __restrict__ Hs, int64_t M,
                                                       int rb, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one 8-column group per thread
  const int64_t tok = idx >> 3;
  const int c0 = (int)(idx & 7) * 8;
  if (tok >= M) return;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nsplit; ++s) {
    const f32x4 a = *(const f32x4*)(Hp + ((int64_t)s * M + tok) * 64 + c0);
    const f32x4 b = *(const f32x4*)(Hp + ((int64_t)s * M + tok) * 64 + c0 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += a[j], v[4 + j] += b[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    v[j] = c < rb ? v[j] * scale : ((c == 63 && rb < 64) ? 1.0f : 0.f);
  }
  if constexpr (sizeof(T) == 2) {
    *(u32x4*)(Hs + tok * 64 + c0) = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  } else {
    *(f32x4*)(Hs + tok * 64 + c0) = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(Hs + tok * 64 + c0 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  }
}

int launch_h_reduce(const float* Hpartial, int nsplit, void* Hsave, int64_t M, int rb, float scale, int dtype,
                    hipStream_t stream) {
  if (M <= 0) return SOW_OK;
  const dim3 grid((unsigned)((M * 8 + 255) / 256));
  if (dtype == SOW_BF16)
    hipLaunchKernelGGL(h_reduce_kernel<bf16_t>, grid, dim3(256), 0, stream, Hpartial, nsplit, (bf16_t*)Hsave, M, rb, scale);
  else
    hipLaunchKernelGGL(h_reduce_kernel<float>, grid, dim3(256), 0, stream, Hpartial, nsplit, (float*)Hsave, M, rb, scale);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

void* g_chain2_stamps = nullptr;   // debug builds: device buffer [blocks][8] of u64 (sow_debug_set_stamps)

static int chain2_grid(const ChainParams& p) {
  // B is DMA'd in aligned 16-byte pieces along its rows; A must be contiguous [rows, r] and 4-byte aligned
  if (p.ntb > 0) {
    const int nsplit = p.st_per > 0 ? ceil_div((p.D1 + 63) / 64, p.st_per) : ceil_div((p.D2 + 63) / 64, p.sl_per);
    return p.ntb * nsplit;
  }
  return ceil_div(p.M, C2_BM);
}

int launch_chain2_group(const ChainParams* ps, int n, bool bwd, hipStream_t stream) {
  if (n <= 0) return SOW_OK;
  if (n > C2_MAXG) return SOW_ERR_SHAPE;
  ChainGroup g{};
  g.n = n;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    const ChainParams& p = ps[i];
    const void* Bp = bwd ? p.F1b : p.F2b;
    const int64_t ldB = bwd ? p.ldf1b : p.ldf2b;
    const void* Ap = bwd ? p.F2b : p.F1b;
    const int64_t ldA = bwd ? p.ldf2b : p.ldf1b;
    if ((reinterpret_cast<uintptr_t>(Bp) & 15) || ldB % 8 || (reinterpret_cast<uintptr_t>(Ap) & 3) || ldA != p.rb)
      return SOW_ERR_ALIGN;
    g.p[i] = p;
    g.p[i].nt_store = sw_on(SW_NO_NT_STORE) ? 0 : 1;
    g.p[i].nt_load = sw_on(SW_NT_LOAD) ? 1 : 0;
    g.p[i].pair_flush = sw_on(SW_NO_PAIR_FLUSH) ? 0 : 1;
    g.start[i] = (int)total;
    total += chain2_grid(p);
  }
  for (int i = n; i <= C2_MAXG; ++i) g.start[i] = (int)total;
  g.stamps = (uint64_t*)g_chain2_stamps;
  if (total <= 0) return SOW_OK;
  if (total > 0x7fffffff) return SOW_ERR_SHAPE;
  // persistent when the block list exceeds one resident round (two 80-KiB workgroups per CU x 256 CUs)
  const int64_t grid = (sw_on(SW_NO_PERSIST) || total < C2_RESIDENT) ? total : C2_RESIDENT;
  bool p16 = !sw_on(SW_NO_PARK16);   // bf16 park tiles: exact only when the epilogue adds nothing to the rounded product
  for (int i = 0; i < n; ++i) p16 = p16 && g.p[i].pair_flush && g.p[i].beta == 0.f && !g.p[i].bias;
#define C2_LAUNCH(B, P)                                                                                              \
  do {                                                                                                               \
    SOW_SET_MAX_LDS_ONCE(C2_LDS, (chain2_kernel<B, P>));                                                             \
    hipLaunchKernelGGL((chain2_kernel<B, P>), dim3((unsigned)grid), dim3(C2_THREADS), C2_LDS, stream, g);            \
  } while (0)
  if (bwd) {
    if (p16) C2_LAUNCH(true, true);
    else C2_LAUNCH(true, false);
  } else {
    if (p16) C2_LAUNCH(false, true);
    else C2_LAUNCH(false, false);
  }
#undef C2_LAUNCH
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

int launch_chain2(const ChainParams& p, bool bwd, hipStream_t stream) { return launch_chain2_group(&p, 1, bwd, stream); }

}  // namespace sow
