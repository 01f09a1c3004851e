// Fused low-rank chain, fp32 streaming version (gfx950):  Y = beta*Y + (scale * X.F1).F2 + bias
//
// Same contract as chain.hip / chain2.hip (reference tn_gradient/layer/sow.py:107-126 forward, and
// its autograd backward with F1 = B^T, F2 = A^T), for the exact-fp32 path (`v_mfma_f32_32x32x2_f32`,
// 64 cycles per 32x32x2).  With r = 50 the fp32 contraction sits ON the ridge (AI ~ 30 flop/B against
// 157 TF / ~5.3 TB/s): phase 1 is MFMA-bound, phase 2 HBM-write-bound, so the kernel needs both the
// LDS-DMA streaming of chain2.hip and a dense MFMA schedule.
//
// Structure (identical roles to chain2.hip): workgroup = 64 tokens = 6 waves -- compute waves 0-3
// (token group = w & 1, half = w >> 1) and loader waves 4-5; 80 KiB LDS, two workgroups per CU.
//   * X streams by LDS-DMA through per-token-group rings of [32 tok x 64 k] fp32 stages (8 KiB, 3 slots,
//     2 in flight); the factors arrive as 16-KiB chunks ([64 rows] x 256 B) in a 2-slot ring filled by
//     the loader waves one chunk ahead; one raw s_barrier per chunk.
//   * The two products are computed TRANSPOSED (H^T = F1^T X^T, Y^T = F2^T H^T): the token index is the
//     MFMA lane, so H^T's accumulator registers ARE the B operand of phase 2 -- register `reg` of rank
//     tile rt holds rank rt*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5), exactly the k pair {rho, rho+4} one
//     32x32x2 MFMA contracts.  Phase 2 therefore issues only the MFMAs whose rank pair is < r
//     (26 instead of 32 for r = 50) and H never touches LDS or HBM except as the saved copy.
//   * Phase 1: each of the two waves of a token group multiplies half of the stage's K range (two
//     partial H^T, summed through LDS once at the hand-off).  Phase 2: each owns one 32-column tile of
//     every 64-column slice, parks it in a wave-private fp32 LDS tile and stores it one slice later
//     as 16-byte row segments (full 128-byte lines).
//   * Operand fetch: with one k per lane-half per MFMA, a ds_read_b128 along k feeds 4 MFMAs (lane-half
//     lh takes k = 8m + 4lh + j, the same map on both operands); operands whose contraction index is
//     the ROW index of their storage (forward factors) are read by ds_read_b32 along a row (32 lanes =
//     128 contiguous bytes, conflict-free).
//   * A's rows are 4r bytes (200 for r = 50): DMA'd as 256-byte rows whose tail is the head of the next
//     row; the garbage lands in rank columns >= r, which always meet an explicit zero (H ranks >= r are
//     masked); pieces that would cross the end of the buffer read a zero page and the last row is
//     rewritten by the loader from guarded loads.
// Measured (r = 50, d = 768, T = 32768, in-kernel s_memtime stamps): a stage / slice of one workgroup takes
// ~5000 cycles with two workgroups per CU against 4096 / 3584 cycles of MFMA work for the two waves that
// share a SIMD, while the kernel moves 201 MB in ~70 us: both the matrix pipe and HBM run at ~75 %.
// Two things that matter: (1) VGPRs <= 128 -- at 129..170 the compiler still reports occupancy 3, but the
// second 6-wave workgroup no longer finds a 2/2/1/1 SIMD placement and the CU runs ONE workgroup (two
// rounds, +25 % time); (2) a phase-2 loop with one runtime predicate per MFMA compiles into a web of taken
// branches; the rank range is cut at quad (8-rank) granularity with forward branches only.
// LDS images are [64 rows][256 B]; images read by ds_read_b128 keep 16-byte chunk c of a row at physical
// chunk c ^ (row & 15) (the XOR is applied to the per-lane DMA SOURCE address).
#include "kernels.hpp"
#include "lds_dma.hpp"

namespace sow {

#ifdef C3_EXPERIMENT_CHEAP_FSPLIT   // timing experiment only (bf16-accurate results): what the per-use split of the FACTOR fragments costs
__device__ __forceinline__ void fsplit3v(float x0, float x1, u32x4 (&pl)[3], int e) {
  const uint32_t h = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, x0), 0x07060302u);
  pl[0][e] = h, pl[1][e] = h, pl[2][e] = h;
}
#else
#define fsplit3v split3v
#endif
constexpr int C3_NTG = 2;
constexpr int C3_NCW = 2 * C3_NTG;
constexpr int C3_NLW = 2;
constexpr int C3_BM = 32 * C3_NTG;
constexpr int C3_DEPTH = 3;            // X stage slots per token group (2 in flight)
constexpr int C3_STAGE = 8192;         // [32 tok][64 k] fp32
constexpr int C3_NSLOT = 2;            // factor chunk slots
constexpr int C3_FSLOT = 16384;        // [64][64] fp32
constexpr int C3_LPW = 16 / C3_NLW;    // 1-KiB DMA instructions per loader wave per chunk
constexpr int C3_XPW = 4;              // X DMA instructions per compute wave per stage
constexpr int C3_RING0 = C3_NSLOT * C3_FSLOT;          // 32 KiB
constexpr int C3_RING = C3_DEPTH * C3_STAGE;           // 24 KiB per token group
constexpr int C3_LDS = C3_RING0 + C3_NTG * C3_RING;    // 80 KiB
constexpr int C3_THREADS = 64 * (C3_NCW + C3_NLW);
constexpr int C3_PARK0 = 32768;        // wave-private Y tiles live behind the 32-KiB exchange area


// X3 = true: both products on the bf16 matrix pipe as 3 x bf16 splits (lds_dma.hpp: split3 / mfma_x3) -- fp32 accuracy
// (the dropped cross terms are <= 2^-23 of each product) at 6 x 32 cycles per 16 k instead of 8 x 64, and, unlike the
// fp32 MFMA, without blocking the other instructions of the SIMD.  Operand fragments are split in registers right
// after they are read (11 VALU instructions per pair of elements).  X3 = false: the exact v_mfma_f32_32x32x2_f32 form
// (F32_EXACT switch; the reference of the A/B test).
template <bool BWD, bool X3> __global__ __launch_bounds__(C3_THREADS, 4) void chain2f_kernel(const ChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  // short-T split (kernels.hpp): workgroup = (token block, split); a split owns a range of phase-1 stages OR of
  // phase-2 slices.  Without a split every workgroup owns all of both.
  const int tb = p.ntb > 0 ? (int)blockIdx.x % p.ntb : (int)blockIdx.x;
  const int split = p.ntb > 0 ? (int)blockIdx.x / p.ntb : 0;
  const int64_t m0 = (int64_t)tb * C3_BM;
  const int D1 = p.D1, D2 = p.D2, rb = p.rb;
  const int nst_all = (D1 + 63) / 64, nsl_all = (D2 + 63) / 64;
  const int st0 = p.st_per > 0 ? split * p.st_per : 0;   // first phase-1 stage of this workgroup
  const int sl0 = p.sl_per > 0 ? split * p.sl_per : 0;   // first phase-2 slice
  const int nst = p.sl_per > 0 ? 0 : (p.st_per > 0 ? (nst_all - st0 < p.st_per ? nst_all - st0 : p.st_per) : nst_all);
  const int nsl = p.st_per > 0 ? 0 : (p.sl_per > 0 ? (nsl_all - sl0 < p.sl_per ? nsl_all - sl0 : p.sl_per) : nsl_all);
  const int total = nst + nsl;
  const float* Amat = (const float*)(BWD ? p.F2b : p.F1b);   // [rows_a, rb] contiguous
  const float* Bmat = (const float*)(BWD ? p.F1b : p.F2b);   // [rb, cols_b], ld = ldb
  const int64_t ldb = BWD ? p.ldf1b : p.ldf2b;
  const int rows_a = BWD ? D2 : D1, cols_b = BWD ? D1 : D2;
  const char* zp = zero_page_for(lane);

  if (w >= C3_NCW) {
    // ------------------------------------------------------------------ loader waves
    // every image is [64 rows][256 B]; instruction i (0..15) covers rows 4i .. 4i+3; loader wave lw issues
    // instructions 8 lw .. 8 lw + 7 (rows 32 lw .. 32 lw + 31)
    const int lw = w - C3_NCW;
    const char* a_end = (const char*)(Amat + (int64_t)rows_a * rb);
    const bool own_last = rows_a > 0 && lw == ((rows_a - 1) & 63) / 32;
    float last_row = 0.f;
    if (own_last && lane < rb) last_row = Amat[(int64_t)(rows_a - 1) * rb + lane];
    asm volatile("" : "+v"(last_row));
    auto chunk_is_a = [&](int c) { return BWD ? (c >= nst) : (c < nst); };
    // the A image is read by ds_read_b128 only in the backward direction (phase 2); the B image only in
    // the backward direction too (phase 1): forward images are read by ds_read_b32 along rows, unswizzled.
    const int rsub = lane >> 4, pc = lane & 15;
    auto issue = [&](int c) {
      char* slot = smem + (c % C3_NSLOT) * C3_FSLOT;
      const int ci = c < nst ? st0 + c : sl0 + (c - nst);
      if (chunk_is_a(c)) {
#pragma unroll
        for (int ii = 0; ii < C3_LPW; ++ii) {
          const int row = 4 * (C3_LPW * lw + ii) + rsub;
          const int lc = BWD ? (pc ^ (row & 15)) : pc;
          const char* q = (const char*)(Amat + (int64_t)(ci * 64 + row) * rb) + 16 * lc;
          dma16(q + 16 <= a_end ? (const void*)q : (const void*)zp, slot + (C3_LPW * lw + ii) * 1024);
        }
      } else {
#pragma unroll
        for (int ii = 0; ii < C3_LPW; ++ii) {
          const int row = 4 * (C3_LPW * lw + ii) + rsub;
          const int lc = BWD ? (pc ^ (row & 15)) : pc;
          const int col = ci * 64 + 4 * lc;
          const void* q = (row < rb && col < cols_b) ? (const void*)(Bmat + (int64_t)row * ldb + col) : (const void*)zp;
          dma16(q, slot + (C3_LPW * lw + ii) * 1024);
        }
      }
    };
    if (total > 0) issue(0);
    for (int c = 0; c < total; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c (the only one in flight) has landed
      __builtin_amdgcn_sched_barrier(0);
      if (chunk_is_a(c)) {   // fix-up: rewrite the last row of A (pieces past the end of the buffer were zero-filled)
        const int base = (c < nst ? st0 + c : sl0 + (c - nst)) * 64;
        const int lr = rows_a - 1 - base;
        if (own_last && lr >= 0 && lr < 64) {
          const int cc = lane >> 2;
          const int phys = BWD ? (cc ^ (lr & 15)) : cc;
          *(float*)(smem + (c % C3_NSLOT) * C3_FSLOT + lr * 256 + phys * 16 + (lane & 3) * 4) = lane < rb ? last_row : 0.f;
        }
      }
      if (c == nst) {      // the two hand-off barriers of the compute waves (partial-H exchange)
        raw_barrier();
        raw_barrier();
      }
      raw_barrier();   // chunk c visible to the consumers; they have finished chunk c-1
      if (c + 1 < total) issue(c + 1);   // slot held chunk c-1: free
    }
    if (nsl == 0) {   // H-only call: the hand-off barriers were not met inside the loop
      raw_barrier();
      raw_barrier();
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int tg = w & 1, hh = w >> 1;
  const int li = lane & 31, lh = lane >> 5;
  char* ring = smem + C3_RING0 + tg * C3_RING;
  const uint32_t ring_a = lds_addr(ring);
  const uint32_t slot_a = lds_addr(smem);
  const float* X = (const float*)p.X;
  const int64_t tok0 = m0 + 32 * tg;

  // X DMA: a stage is 8 instructions of 4 token rows; this wave issues i = 4 hh .. 4 hh + 3 (rows 16 hh ..)
  const int drow = lane >> 4, dpc = lane & 15;
  auto issue_x_one = [&](int st, int ii) {
    char* dst = ring + (st % C3_DEPTH) * C3_STAGE;
    const int row = 4 * (C3_XPW * hh + ii) + drow;
    const int lc = dpc ^ (row & 15);
    const int64_t tk = tok0 + row;
    const int col = (st0 + st) * 64 + 4 * lc;
    const void* q = (tk < p.M && col < D1) ? (const void*)(X + tk * p.ldx + col) : (const void*)zp;
    dma16(q, dst + (C3_XPW * hh + ii) * 1024);
  };
  auto issue_x = [&](int st) {
#pragma unroll
    for (int ii = 0; ii < C3_XPW; ++ii) issue_x_one(st, ii);
  };

  const int xsw = li & 15;                       // b128 row swizzle of this lane's row (rows li and 32 + li alike)
  const uint32_t xrow = (uint32_t)(li * 256);

  f32x16 hacc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[a][i] = 0.f;
  const int pre = nst < (C3_DEPTH - 1) ? nst : (C3_DEPTH - 1);
  for (int st = 0; st < pre; ++st) issue_x(st);

  // ================================================================== phase 1: H^T = F1^T . X^T (K half hh)
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int newer = (nst - 1 - st) < (C3_DEPTH - 2) ? (nst - 1 - st) : (C3_DEPTH - 2);
    wait_groups<C3_XPW>(newer);   // this wave's rows of X stage `st` have landed
    raw_barrier();                // ... and so have the partner's rows and factor chunk `st`
    if (st + C3_DEPTH - 1 < nst) issue_x(st + C3_DEPTH - 1);   // into the slot of stage st-1
    const uint32_t xs = ring_a + (uint32_t)((st % C3_DEPTH) * C3_STAGE) + xrow;
    const uint32_t fs = slot_a + (uint32_t)((st % C3_NSLOT) * C3_FSLOT);
    if constexpr (X3) {
      // two k-steps of 16: lane-half lh multiplies k = 32 hh + 16 s + 8 lh + j, j = 0 .. 7
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f32x4 xa, xb, fa[2], fb[2];
        const int c0 = 8 * hh + 4 * s2 + 2 * lh;   // first 16-byte chunk of this lane's 8 k
        DS_READ_B128(xa, xs + (uint32_t)(((c0) ^ xsw) * 16), 0);
        DS_READ_B128(xb, xs + (uint32_t)(((c0 + 1) ^ xsw) * 16), 0);
        if constexpr (BWD) {
          const uint32_t fr = fs + (uint32_t)(li * 256);
          DS_READ_B128(fa[0], fr + (uint32_t)(((c0) ^ xsw) * 16), 0);
          DS_READ_B128(fb[0], fr + (uint32_t)(((c0 + 1) ^ xsw) * 16), 0);
          DS_READ_B128(fa[1], fr + (uint32_t)(((c0) ^ xsw) * 16), 32 * 256);
          DS_READ_B128(fb[1], fr + (uint32_t)(((c0 + 1) ^ xsw) * 16), 32 * 256);
        } else {
          const uint32_t fr = fs + (uint32_t)((32 * hh + 16 * s2 + 8 * lh) * 256 + li * 4);
          DS_READ_B32(fa[0][0], fr, 0 * 256);
          DS_READ_B32(fa[0][1], fr, 1 * 256);
          DS_READ_B32(fa[0][2], fr, 2 * 256);
          DS_READ_B32(fa[0][3], fr, 3 * 256);
          DS_READ_B32(fb[0][0], fr, 4 * 256);
          DS_READ_B32(fb[0][1], fr, 5 * 256);
          DS_READ_B32(fb[0][2], fr, 6 * 256);
          DS_READ_B32(fb[0][3], fr, 7 * 256);
          DS_READ_B32(fa[1][0], fr, 0 * 256 + 128);
          DS_READ_B32(fa[1][1], fr, 1 * 256 + 128);
          DS_READ_B32(fa[1][2], fr, 2 * 256 + 128);
          DS_READ_B32(fa[1][3], fr, 3 * 256 + 128);
          DS_READ_B32(fb[1][0], fr, 4 * 256 + 128);
          DS_READ_B32(fb[1][1], fr, 5 * 256 + 128);
          DS_READ_B32(fb[1][2], fr, 6 * 256 + 128);
          DS_READ_B32(fb[1][3], fr, 7 * 256 + 128);
        }
        LGKM_WAIT0();
        u32x4 xp[3], fp[2][3];
        split3v(xa[0], xa[1], xp, 0);
        split3v(xa[2], xa[3], xp, 1);
        split3v(xb[0], xb[1], xp, 2);
        split3v(xb[2], xb[3], xp, 3);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          fsplit3v(fa[tl][0], fa[tl][1], fp[tl], 0);
          fsplit3v(fa[tl][2], fa[tl][3], fp[tl], 1);
          fsplit3v(fb[tl][0], fb[tl][1], fp[tl], 2);
          fsplit3v(fb[tl][2], fb[tl][3], fp[tl], 3);
        }
        hacc[0] = mfma_x3(fp[0], xp, hacc[0]);
        hacc[1] = mfma_x3(fp[1], xp, hacc[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    // four groups of 8 MFMAs (k = 32 hh + 8 m + 4 lh + j); the operands of group m+1 are fetched while
    // group m multiplies (double-buffered, counted lgkmcnt)
    f32x4 xf[2];
    f32x4 fq[2][2];   // [buffer][rank tile]: BWD one b128 each; FWD four dwords each
    auto fetch = [&](int m, int bsel) {
      DS_READ_B128(xf[bsel], xs + (uint32_t)(((8 * hh + 2 * m + lh) ^ xsw) * 16), 0);
      if constexpr (BWD) {
        // F1 = B, image [64 rank rows][64 k]: b128 along k, same k map as X
        const uint32_t fa = fs + (uint32_t)(li * 256 + (((8 * hh + 2 * m + lh) ^ xsw) * 16));
        DS_READ_B128(fq[bsel][0], fa, 0);
        DS_READ_B128(fq[bsel][1], fa, 32 * 256);
      } else {
        // F1 = A, image [64 k rows][64 rank]: one dword per lane along a row
        const uint32_t fa = fs + (uint32_t)((32 * hh + 8 * m + 4 * lh) * 256 + li * 4);
        DS_READ_B32(fq[bsel][0][0], fa, 0 * 256);
        DS_READ_B32(fq[bsel][1][0], fa, 0 * 256 + 128);
        DS_READ_B32(fq[bsel][0][1], fa, 1 * 256);
        DS_READ_B32(fq[bsel][1][1], fa, 1 * 256 + 128);
        DS_READ_B32(fq[bsel][0][2], fa, 2 * 256);
        DS_READ_B32(fq[bsel][1][2], fa, 2 * 256 + 128);
        DS_READ_B32(fq[bsel][0][3], fa, 3 * 256);
        DS_READ_B32(fq[bsel][1][3], fa, 3 * 256 + 128);
      }
    };
    constexpr int RPG = BWD ? 3 : 9;   // LDS reads per group
    fetch(0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < 3) {
        fetch(m + 1, (m + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RPG) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hacc[0] = mfma32(fq[m & 1][0][j], xf[m & 1][j], hacc[0]);
        hacc[1] = mfma32(fq[m & 1][1][j], xf[m & 1][j], hacc[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ================================================================== hand-off: sum the two K halves
  const int64_t tok = tok0 + li;
  if (p.Hload) {
    // phase-2-only workgroup: H comes from memory; register `reg` of tile rt is rank rt*32 + (reg&3) + 8(reg>>2) + 4lh
    // of token li.  The 1.0 of column 63 (dbias trick) must not reach the product.
    raw_barrier();
    raw_barrier();
    const float* Hl = (const float*)p.Hload + tok * 64 + 4 * lh;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tok < p.M) v = *(const f32x4*)(Hl + rt * 32 + 8 * rq);
        if (rt == 1 && rq == 3 && lh == 1 && rb < 64) v[3] = 0.f;
        hacc[rt][4 * rq + 0] = v[0], hacc[rt][4 * rq + 1] = v[1], hacc[rt][4 * rq + 2] = v[2], hacc[rt][4 * rq + 3] = v[3];
      }
  } else {
  raw_barrier();   // every X read of the workgroup is done: the rings become the exchange buffers
  {
    float* xch = (float*)(smem + C3_RING0) + w * 2048;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) xch[(rt * 16 + reg) * 64 + lane] = hacc[rt][reg];
  }
  raw_barrier();   // partials visible to the partner
  {
    const uint32_t pa = lds_addr(smem + C3_RING0) + (uint32_t)((w ^ 2) * 8192 + lane * 4);
    uint32_t pv[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) DS_READ_B32(pv[i], pa, i * 256);
    LGKM_WAIT0();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) hacc[rt][reg] += __builtin_bit_cast(float, pv[rt * 16 + reg]);
  }
  if (p.Hpartial) {
    // phase-1 slab of a short-T split: the raw fp32 sums of this K range, one 32-rank tile per half
    if (tok < p.M) {
      float* Hp = p.Hpartial + ((int64_t)split * p.M + tok) * 64 + hh * 32 + 4 * lh;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *(f32x4*)(Hp + 8 * rq) = hh ? (f32x4){hacc[1][4 * rq + 0], hacc[1][4 * rq + 1], hacc[1][4 * rq + 2], hacc[1][4 * rq + 3]}
                                    : (f32x4){hacc[0][4 * rq + 0], hacc[0][4 * rq + 1], hacc[0][4 * rq + 2], hacc[0][4 * rq + 3]};
    }
  } else {
  // scale, mask rank rows >= r; the saved copy [M, 64] carries 1.0 in column 63 when free (the dbias
  // trick of the skinny-TN kernel); one 32-rank tile per half, 16-byte pieces.
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
      hacc[rt][reg] = r < rb ? hacc[rt][reg] * p.scale : 0.f;
    }
    if (p.Hsave && hh == rt && tok < p.M) {
      float* Hs = (float*)p.Hsave + tok * 64 + rt * 32 + 4 * lh;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {hacc[rt][4 * rq + 0], hacc[rt][4 * rq + 1], hacc[rt][4 * rq + 2], hacc[rt][4 * rq + 3]};
        if (rt == 1 && rq == 3 && lh == 1 && rb < 64) v[3] = 1.0f;   // column 63 <- 1.0
        *(f32x4*)(Hs + 8 * rq) = v;
      }
    }
  }
  }
  }
  // ================================================================== phase 2: Y^T = F2^T . H^T (column tile hh)
  [[maybe_unused]] u32x4 hp[4][3];   // X3: H^T as the B operand of k-step s, split once
  [[maybe_unused]] const int ksteps = (rb + 15) / 16;
  if constexpr (X3) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int pr = 0; pr < 4; ++pr)
        split3v(hacc[s4 >> 1][8 * (s4 & 1) + 2 * pr], hacc[s4 >> 1][8 * (s4 & 1) + 2 * pr + 1], hp[s4], pr);
  }
  float* Y = (float*)p.Y;
  const float* bias = (const float*)p.bias;
  const int nq = (rb + 7) / 8;
  char* park = smem + C3_RING0 + C3_PARK0 + w * 4096;   // [32 tok][32 col] fp32, chunk c of row at c ^ ((row >> 1) & 7)
  const uint32_t park_a = lds_addr(park);
  f32x4 fv[4];   // the parked tile of the previous slice on its way to global memory
  auto flush_read = [&]() {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 8 + (lane >> 3), c = lane & 7;
      DS_READ_B128(fv[pass], park_a + (uint32_t)(r * 128 + ((c ^ ((r >> 1) & 7)) * 16)), 0);
    }
  };
  auto flush_store = [&](int sl_prev, int pass) {   // 8 rows of 128 B per pass
    const int r = pass * 8 + (lane >> 3), c = lane & 7;
    const int64_t tk = tok0 + r;
    const int col = (sl0 + sl_prev) * 64 + hh * 32 + 4 * c;
    if (tk < p.M && col < D2) {
      f32x4 o = fv[pass];
      float* dst = Y + tk * p.ldy + col;
      if (p.beta != 0.f) {
        const f32x4 old = *(const f32x4*)dst;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += p.beta * old[e];
      }
      if (bias) {
        const f32x4 bv = *(const f32x4*)(bias + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += bv[e];
      }
      // streaming store (see chain2.hip): Y is not re-read here; keeps the L2s clean for the end-of-kernel write-back
      asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(o) : "memory");
    }
  };
#pragma unroll 1
  for (int sl = 0; sl < nsl; ++sl) {
    raw_barrier();   // factor chunk nst + sl is in its slot
    const uint32_t fs = slot_a + (uint32_t)(((nst + sl) % C3_NSLOT) * C3_FSLOT);
    // Rank quads q = 0 .. nq-1 (8 ranks each: registers 4g .. 4g+3 of tile rt, q = 4 rt + g).  A quad whose
    // ranks are partly >= r still multiplies exact zeros (H is masked).  The factor operand of quad q+1 is
    // fetched while quad q multiplies.
    f32x16 yacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) yacc[i] = 0.f;
    if constexpr (X3) {
      // k-step s contracts ranks 16 s + 8 (j >> 2) + 4 lh + (j & 3), j = 0 .. 7 -- the order in which H^T's accumulator
      // registers hold them (hp[s] was split once, after the hand-off)
      if (sl > 0) flush_read();
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 < ksteps) {
          f32x4 fa, fb;
          if constexpr (BWD) {
            const uint32_t fr = fs + (uint32_t)((hh * 32 + li) * 256);
            DS_READ_B128(fa, fr + (uint32_t)(((4 * s4 + lh) ^ xsw) * 16), 0);
            DS_READ_B128(fb, fr + (uint32_t)(((4 * s4 + 2 + lh) ^ xsw) * 16), 0);
          } else {
            const uint32_t fr = fs + (uint32_t)((16 * s4 + 4 * lh) * 256 + (hh * 32 + li) * 4);
            DS_READ_B32(fa[0], fr, 0 * 256);
            DS_READ_B32(fa[1], fr, 1 * 256);
            DS_READ_B32(fa[2], fr, 2 * 256);
            DS_READ_B32(fa[3], fr, 3 * 256);
            DS_READ_B32(fb[0], fr, 8 * 256);
            DS_READ_B32(fb[1], fr, 9 * 256);
            DS_READ_B32(fb[2], fr, 10 * 256);
            DS_READ_B32(fb[3], fr, 11 * 256);
          }
          LGKM_WAIT0();
          u32x4 f2p[3];
          fsplit3v(fa[0], fa[1], f2p, 0);
          fsplit3v(fa[2], fa[3], f2p, 1);
          fsplit3v(fb[0], fb[1], f2p, 2);
          fsplit3v(fb[2], fb[3], f2p, 3);
          yacc = mfma_x3(f2p, hp[s4], yacc);
        }
        if (sl > 0) flush_store(sl - 1, s4);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    f32x4 fq[2];
    auto fetch2 = [&](int q, int bsel) {
      if constexpr (BWD) {
        // F2 = A, image [64 n rows][64 rank]: b128 along rank; quad q = chunk 2q + lh
        DS_READ_B128(fq[bsel], fs + (uint32_t)((hh * 32 + li) * 256 + (((2 * q + lh) ^ xsw) * 16)), 0);
      } else {
        // F2 = B, image [64 rank rows][64 n]: one dword per lane along a row; rank = 8 q + 4 lh + i
        const uint32_t fa = fs + (uint32_t)((8 * q + 4 * lh) * 256 + (hh * 32 + li) * 4);
        DS_READ_B32(fq[bsel][0], fa, 0);
        DS_READ_B32(fq[bsel][1], fa, 256);
        DS_READ_B32(fq[bsel][2], fa, 512);
        DS_READ_B32(fq[bsel][3], fa, 768);
      }
    };
    constexpr int RPQ = BWD ? 1 : 4;
    {
      if (sl > 0) flush_read();   // previous slice: LDS -> registers now, one store pass after each of quads 0..3
      fetch2(0, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < nq) {
          if (q + 1 < nq) {
            fetch2(q + 1, (q + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RPQ) : "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
          yacc = mfma32(fq[q & 1][0], hacc[q >> 2][4 * (q & 3) + 0], yacc);
          yacc = mfma32(fq[q & 1][1], hacc[q >> 2][4 * (q & 3) + 1], yacc);
          yacc = mfma32(fq[q & 1][2], hacc[q >> 2][4 * (q & 3) + 2], yacc);
          yacc = mfma32(fq[q & 1][3], hacc[q >> 2][4 * (q & 3) + 3], yacc);
          if (q < 4 && sl > 0) flush_store(sl - 1, q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (sl > 0 && nq < 4) {   // fewer than four quads (r <= 24): the remaining store passes
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q >= nq) flush_store(sl - 1, q);
      }
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    // park this slice: register quad rq holds columns 8 rq + 4 lh .. +3 of token li
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int chunk = 2 * rq + lh;
      f32x4 v = {yacc[4 * rq + 0], yacc[4 * rq + 1], yacc[4 * rq + 2], yacc[4 * rq + 3]};
      *(f32x4*)(park + li * 128 + ((chunk ^ ((li >> 1) & 7)) * 16)) = v;
    }
  }
  if (nsl > 0) {
    flush_read();
    LGKM_WAIT0();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) flush_store(nsl - 1, pass);
  }
}

// =================================================================================================
bool chain2f_supported(const ChainParams& p, int dtype) {
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // rb >= 2: with 4-byte rows (rb = 1) the zero-filled end-crossing pieces would cost the last three rows of A
  if (dtype != SOW_F32 || p.ra != 0 || p.rb < 2 || p.rb > 64) return false;
  if (p.D1 % 4 || p.D2 % 4 || p.ldx % 4 || p.ldy % 4) return false;
  if (!a16(p.X) || !a16(p.Y) || (p.bias && !a16(p.bias)) || (p.Hsave && !a16(p.Hsave))) return false;
  if (p.M < 64) return false;   // (short inputs run T/64 workgroups either way; measured 1.4x faster than the generic kernel at T = 1024)
  return true;
}

int launch_chain2f(const ChainParams& p, bool bwd, hipStream_t stream) {
  // B is DMA'd in aligned 16-byte pieces along its rows; A must be contiguous [rows, r] (4-byte aligned rows)
  const void* Bp = bwd ? p.F1b : p.F2b;
  const int64_t ldB = bwd ? p.ldf1b : p.ldf2b;
  const void* Ap = bwd ? p.F2b : p.F1b;
  const int64_t ldA = bwd ? p.ldf2b : p.ldf1b;
  if ((reinterpret_cast<uintptr_t>(Bp) & 15) || ldB % 4 || (reinterpret_cast<uintptr_t>(Ap) & 3) || ldA != p.rb)
    return SOW_ERR_ALIGN;
  int grid = ceil_div(p.M, C3_BM);
  if (p.ntb > 0) {
    const int nsplit = p.st_per > 0 ? ceil_div((p.D1 + 63) / 64, p.st_per) : ceil_div((p.D2 + 63) / 64, p.sl_per);
    grid = p.ntb * nsplit;
  }
  const bool x3 = !sw_on(SW_F32_EXACT);
#define C3_LAUNCH(...)                                                                  \
  do {                                                                                  \
    SOW_SET_MAX_LDS_ONCE(C3_LDS, __VA_ARGS__);                                          \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(grid), dim3(C3_THREADS), C3_LDS, stream, p); \
  } while (0)
  if (bwd && x3) C3_LAUNCH(chain2f_kernel<true, true>);
  else if (bwd) C3_LAUNCH(chain2f_kernel<true, false>);
  else if (x3) C3_LAUNCH(chain2f_kernel<false, true>);
  else C3_LAUNCH(chain2f_kernel<false, false>);
#undef C3_LAUNCH
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
