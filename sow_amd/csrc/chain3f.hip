// Fused low-rank chain for fp32 tensors, second generation (gfx950):  Y = beta*Y + (scale * X.F1).F2 + bias
//
// Same contract as chain2f.hip (reference tn_gradient/layer/sow.py:107-126 forward, and its autograd backward with
// F1 = B^T, F2 = A^T); products on the bf16 matrix pipe as 3 x bf16 splits (lds_dma.hpp: fp32-equivalent).  What the
// counters said about chain2f at the north-star point (profiles/r03_pmc_sq_fp32_before.txt): matrix pipe 27 % busy, 12.8 M
// VALU instructions per launch of which 7 M split FACTOR fragments -- the same 64 x 64 chunk split again by every wave that
// uses it -- and the compute waves stalled at issue for half of their lifetime (fragment reads, splits and MFMAs of a
// k-step run one after the other).  Here:
//   * the factors are split ONCE per launch by a pre-pass (chain3f_planes_kernel: ~40 K elements) into ready-to-read
//     LDS images -- per 64-wide chunk three bf16 planes [64 rows][64 k], 16-byte chunk c of row r at c ^ ((r >> 1) & 7) --
//     kept in the caller's workspace (L2-resident, 24 KiB per chunk); a chunk reaches LDS as 24 lane-linear 1-KiB DMA
//     instructions and its fragments are plain ds_read_b128, no VALU.  The kernel is direction-agnostic: forward and
//     backward differ only in what the pre-pass gathers;
//   * workgroup = 128 tokens = 8 waves (token group tg = w >> 1 of 32 tokens, half hh = w & 1), one per CU (144 KiB of
//     LDS: 2 factor slots of 24 KiB + 4 X rings of 3 x 8 KiB), no loader waves: every wave issues 3 of a chunk's 24 DMA
//     instructions and its 4 of the 8 of its token group's X stage, counted vmcnt;
//   * phase 1 (H^T = F1^T X^T, K half hh of every 64-wide stage = two k-steps of 16) is software-pipelined by k-step: the
//     fragment reads of k-step j + 1 are issued before the 6 x NT MFMAs of k-step j, and the split of its X fragment (36
//     VALU) is interleaved with them; one raw s_barrier per stage, passed with the last k-step's operands already in
//     registers, so that the matrix pipe has work across it;
//   * phase 2 is computed as Y = H . F2 (H^T's accumulator registers are the A operand: lane = token, the 8 k of a k-step
//     are the ranks the registers hold; the pre-pass stores F2's ranks in that order): a lane of the result holds ONE output
//     column of 16 tokens, so register `reg` of the wave is a 128-byte row segment per token and goes straight to memory with
//     one global_store_dword -- no LDS park, no flush pass;
//   * rank tiles are a template parameter: r <= 32 runs one 32-rank tile (half of the phase-1 MFMAs).
// Launch condition: T >= 8192, no short-T split (api.hip keeps chain2f for those and for the exact-fp32 form).
#include "kernels.hpp"
#include "lds_dma.hpp"

namespace sow {

constexpr int C4_BM = 128;
constexpr int C4_THREADS = 512;
constexpr int C4_XSTAGE = 8192;                       // [32 tok][64 k] fp32
constexpr int C4_XDEPTH = 3;
constexpr int C4_PLANE = 8192;                        // [64 rows][64 k] bf16
constexpr int C4_FSLOT = 3 * C4_PLANE;                // 24 KiB
constexpr int C4_RING0 = 2 * C4_FSLOT;                // 48 KiB
constexpr int C4_XRING = C4_XDEPTH * C4_XSTAGE;       // 24 KiB per token group
constexpr int C4_LDS = C4_RING0 + 4 * C4_XRING;       // 144 KiB

struct Chain3fParams {
  const float* X;
  float* Y;
  float* Hsave;
  const float* bias;
  const char* planes;   // (nst + nsl) images of 24 KiB
  int64_t M, ldx, ldy;
  int D1, D2, rb;
  float scale, beta;
};

// ---- pre-pass: factor planes --------------------------------------------------------------------------------------------
struct PlaneParams {
  const float* Amat;    // [rows_a, rb] contiguous: forward F1 (rows = k), backward F2 (rows = output column)
  const float* Bmat;    // [rb, cols_b], ld ldb: forward F2, backward F1
  int64_t ldb;
  int rb, D1, D2, nst, bwd;
  char* out;
};
__device__ __forceinline__ void split3_scalar(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  h = u >> 16;
  const float r = x - __builtin_bit_cast(float, u & 0xffff0000u);       // exact
  const uint32_t ur = __builtin_bit_cast(uint32_t, r);
  m = ur >> 16;
  const float q = r - __builtin_bit_cast(float, ur & 0xffff0000u);      // exact, <= 8 bits
  l = __builtin_bit_cast(uint32_t, q) >> 16;
}
// block c < nst: phase-1 image of chunk c, row = rank, position j = k c*64 + j;
// block c >= nst: phase-2 image of slice c - nst, row = output column within the slice, position q = rank
//   rho(q) = 16 (q >> 4) + 8 ((q >> 2) & 1) + 4 ((q >> 3) & 1) + (q & 3)   (the order H^T's accumulators hold them)
// (four blocks of 128 threads per image, one 16-byte piece of every plane per thread: the launch is a single chain of
// gather -> split -> store, ~3 us, and sits on the critical path of the chain launch behind it)
__global__ __launch_bounds__(128) void chain3f_planes_kernel(const PlaneParams q) {
  const int c = blockIdx.x >> 2;
  char* img = q.out + (size_t)c * C4_FSLOT;
  {
    const int item = threadIdx.x + 128 * (blockIdx.x & 3);
    const int row = item >> 3, c16 = item & 7;
    u32x4 pl[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int pos = 8 * c16 + e;
      float v = 0.f;
      if (c < q.nst) {
        const int k = c * 64 + pos, rho = row;
        if (rho < q.rb && k < q.D1) v = q.bwd ? q.Bmat[(int64_t)rho * q.ldb + k] : q.Amat[(int64_t)k * q.rb + rho];
      } else {
        const int col = (c - q.nst) * 64 + row;
        const int rho = 16 * (pos >> 4) + 8 * ((pos >> 2) & 1) + 4 * ((pos >> 3) & 1) + (pos & 3);
        if (rho < q.rb && col < q.D2) v = q.bwd ? q.Amat[(int64_t)col * q.rb + rho] : q.Bmat[(int64_t)rho * q.ldb + col];
      }
      uint32_t h, m, l;
      split3_scalar(v, h, m, l);
      const int sh = 16 * (e & 1);
      pl[0][e >> 1] |= h << sh, pl[1][e >> 1] |= m << sh, pl[2][e >> 1] |= l << sh;
    }
    const int off = row * 128 + ((c16 ^ ((row >> 1) & 7)) * 16);
#pragma unroll
    for (int p = 0; p < 3; ++p) *(u32x4*)(img + p * C4_PLANE + off) = pl[p];
  }
}

// ---- main kernel --------------------------------------------------------------------------------------------------------
template <int NT> struct C4Frag {
  u32x4 xp[3];        // X^T fragment of one k-step, three planes
  u32x4 fp[NT][3];    // F1^T fragments of the rank tiles
};

// The LDS reads are inline asm whose results land later than the compiler believes; every counted wait therefore takes the
// registers it releases as in/out operands, so that no consumer can be scheduled (or hoisted out of a branch) above it.
template <int NT> __device__ __forceinline__ void c4_tie(C4Frag<NT>& f) {
  if constexpr (NT == 2)
    asm volatile("" : "+v"(f.fp[0][0]), "+v"(f.fp[0][1]), "+v"(f.fp[0][2]), "+v"(f.fp[1][0]), "+v"(f.fp[1][1]), "+v"(f.fp[1][2]));
  else
    asm volatile("" : "+v"(f.fp[0][0]), "+v"(f.fp[0][1]), "+v"(f.fp[0][2]));
}
__device__ __forceinline__ void c4_store_nt(uint32_t voff, float v, const float* base) {
  asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(voff), "v"(v), "s"(base) : "memory");
}

// NT = rank tiles of phase 1 (ceil(r / 32)), KS = k-steps of phase 2 (ceil(r / 16)): both compile-time, so that no LDS read sits
// under a runtime condition (hipcc copies the registers of conditionally executed asm reads right behind them -- before the
// data has landed).
template <int NT, int KS> __global__ __launch_bounds__(C4_THREADS, 2) void chain3f_kernel(const Chain3fParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tg = w >> 1, hh = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * C4_BM;
  const int64_t tok0 = m0 + 32 * tg;
  const int D1 = p.D1, D2 = p.D2, rb = p.rb;
  const int nst = (D1 + 63) / 64, nsl = (D2 + 63) / 64, ntot = nst + nsl;
  const char* zp = zero_page_for(lane);
  char* xring = smem + C4_RING0 + tg * C4_XRING;
  const uint32_t lds0 = lds_addr(smem);

  // factor DMA: chunk c -> slot c & 1; this wave's instructions 3 w .. 3 w + 2 (lane-linear 1-KiB pieces)
  const char* fsrc = p.planes + (3 * w) * 1024 + lane * 16;
  auto issue_f = [&](int c) {
    char* dst = smem + (c & 1) * C4_FSLOT + (3 * w) * 1024;
    const char* src = fsrc + (size_t)c * C4_FSLOT;
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) dma16(src + ii * 1024, dst + ii * 1024);
  };
  // X DMA: a stage is 8 instructions of 4 token rows; this wave issues 4 hh .. 4 hh + 3 (rows 16 hh ..); 16-byte chunk c of
  // a row at c ^ (row & 15) (XOR on the per-lane SOURCE address)
  // (running per-lane source pointers; rows beyond M read the zero page with a zero step, so that a full stage -- every one but
  // possibly the last -- is four DMA instructions and four pointer increments)
  const char* xsrc[4];
  int xcol[4];
  int64_t xstep[4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int row = 4 * (4 * hh + ii) + (lane >> 4);
    const int lc = (lane & 15) ^ (row & 15);
    const int64_t tk = tok0 + row;
    xcol[ii] = 4 * lc;
    xsrc[ii] = tk < p.M ? (const char*)(p.X + tk * p.ldx + 4 * lc) : zp;
    xstep[ii] = tk < p.M ? 256 : 0;
  }
  auto issue_x = [&](int st) {   // strictly in order of st
    char* dst = xring + (st % C4_XDEPTH) * C4_XSTAGE + (4 * hh) * 1024;
    if (st * 64 + 64 <= D1) {   // wave-uniform
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        dma16(xsrc[ii], dst + ii * 1024);
        xsrc[ii] += xstep[ii];
      }
    } else {                    // the K tail: columns beyond D1 read zeros
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const void* src = (st * 64 + xcol[ii] < D1) ? (const void*)xsrc[ii] : (const void*)zp;
        dma16(src, dst + ii * 1024);
        xsrc[ii] += xstep[ii];
      }
    }
  };

  // fragment addresses.  X: lane li reads its row's chunks 8 hh + 4 s2 + 2 lh (+ 1); the k-step and the second chunk are
  // XORs of the byte offset (bits 6 and 4).  F1: row 32 tl + li of every plane, chunk 4 hh + 2 s2 + lh (k-step: bit 5).
  const uint32_t xo = (uint32_t)(li * 256 + (((8 * hh + 2 * lh) ^ (li & 15)) * 16));
  const uint32_t fo = (uint32_t)(li * 128 + (((4 * hh + lh) ^ ((li >> 1) & 7)) * 16));
  const uint32_t xbase = lds0 + (uint32_t)(C4_RING0 + tg * C4_XRING);

  f32x16 hacc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[a][i] = 0.f;

  C4Frag<NT> P, Q;   // P: k-step 1 of a stage, Q: k-step 0
  f32x4 xr[2];
  // (before the first stage P holds zeros: its MFMAs add nothing, and the loop body needs no first-iteration branch)
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    P.xp[pl] = (u32x4){0, 0, 0, 0};
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) P.fp[tl][pl] = (u32x4){0, 0, 0, 0};
  }

#define C4_READ(FR, st_, S2)                                                                   \
  do {                                                                                         \
    const uint32_t xs__ = xbase + (uint32_t)(((st_) % C4_XDEPTH) * C4_XSTAGE);                 \
    const uint32_t fs__ = lds0 + (uint32_t)(((st_) & 1) * C4_FSLOT);                           \
    DS_READ_B128(xr[0], xs__ + (xo ^ ((S2) ? 64u : 0u)), 0);                                   \
    DS_READ_B128(xr[1], xs__ + (xo ^ ((S2) ? 80u : 16u)), 0);                                  \
    _Pragma("unroll") for (int tl__ = 0; tl__ < NT; ++tl__) {                                  \
      const uint32_t fa__ = fs__ + (fo ^ ((S2) ? 32u : 0u));                                   \
      DS_READ_B128(FR.fp[tl__][0], fa__, 0 * C4_PLANE + tl__ * 4096);                          \
      DS_READ_B128(FR.fp[tl__][1], fa__, 1 * C4_PLANE + tl__ * 4096);                          \
      DS_READ_B128(FR.fp[tl__][2], fa__, 2 * C4_PLANE + tl__ * 4096);                          \
    }                                                                                          \
  } while (0)
  // wait until the two X reads of the latest C4_READ have landed (its 3 NT factor reads may still be in flight)
#define C4_WAIT_X() asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xr[0]), "+v"(xr[1]) : "n"(3 * NT) : "memory")
  // the 6 NT MFMAs of fragment CUR with the split of xr into NXT.xp spread over them (one element pair per MFMA slot)
#define C4_MFMA_SPLIT(CUR, NXT, DO_SPLIT)                                                      \
  do {                                                                                         \
    _Pragma("unroll") for (int i__ = 0; i__ < 6 * NT; ++i__) {                                 \
      const int tl__ = i__ / 6, k__ = i__ % 6;                                                 \
      constexpr int ia__[6] = {2, 0, 1, 1, 0, 0}, ib__[6] = {0, 2, 1, 0, 1, 0};                \
      hacc[tl__] = mfma32(as_bf16x8(CUR.fp[tl__][ia__[k__]]), as_bf16x8(CUR.xp[ib__[k__]]), hacc[tl__]); \
      if ((DO_SPLIT) && i__ >= 1 && i__ <= 4) {                                                \
        const int e__ = i__ - 1;                                                               \
        split3v(xr[e__ >> 1][2 * (e__ & 1)], xr[e__ >> 1][2 * (e__ & 1) + 1], NXT.xp, e__);    \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }                                                                                          \
  } while (0)

  // ================================================================== phase 1
  issue_f(0);
  issue_x(0);
  if (nst > 1) issue_x(1);
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    // this wave's pieces of X stage st and of factor chunk st have landed (only X stage st + 1 may still be in flight)
    if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    raw_barrier();   // ... and everyone's; every wave holds the fragments of stage st - 1 in registers: its slots are free
    if (st + 1 < ntot) issue_f(st + 1);
    if (st + 2 < nst) issue_x(st + 2);
    c4_tie<NT>(P);   // P's factor fragments (k-step 1 of the previous stage) landed with the barrier's lgkmcnt(0)
    C4_READ(Q, st, 0);
    C4_WAIT_X();
    __builtin_amdgcn_sched_barrier(0);
    C4_MFMA_SPLIT(P, Q, true);
    __builtin_amdgcn_sched_barrier(0);
    C4_READ(P, st, 1);
    // Q's factor fragments: everything but the 2 + 3 NT reads just issued
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + 3 * NT) : "memory");
    c4_tie<NT>(Q);
    __builtin_amdgcn_sched_barrier(0);
    // (the X reads of k-step 1 have had the first MFMA's time to land: the split starts one slot in)
    {
      _Pragma("unroll") for (int i = 0; i < 6 * NT; ++i) {
        const int tl = i / 6, k = i % 6;
        constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};
        hacc[tl] = mfma32(as_bf16x8(Q.fp[tl][ia[k]]), as_bf16x8(Q.xp[ib[k]]), hacc[tl]);
        if (i == 1) {
          C4_WAIT_X();
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i >= 1 && i <= 4) {
          const int e = i - 1;
          split3v(xr[e >> 1][2 * (e & 1)], xr[e >> 1][2 * (e & 1) + 1], P.xp, e);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // the last k-step
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  c4_tie<NT>(P);
  __builtin_amdgcn_sched_barrier(0);
  C4_MFMA_SPLIT(P, Q, false);
#undef C4_READ
#undef C4_WAIT_X
#undef C4_MFMA_SPLIT

  // ================================================================== hand-off: sum the two K halves
  const int64_t tok = tok0 + li;
  raw_barrier();   // every X read of the workgroup is done: the X rings become the exchange buffers
  {
    float* xch = (float*)(smem + C4_RING0) + w * 2048;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) xch[(rt * 16 + reg) * 64 + lane] = hacc[rt][reg];
  }
  raw_barrier();   // partials visible to the partner
  {
    const uint32_t pa = lds0 + (uint32_t)(C4_RING0 + (w ^ 1) * 8192 + lane * 4);
    uint32_t pv[16 * NT];
#pragma unroll
    for (int i = 0; i < 16 * NT; ++i) DS_READ_B32(pv[i], pa, i * 256);
    LGKM_WAIT0();
#pragma unroll
    for (int i = 0; i < 16 * NT; i += 8)
      asm volatile("" : "+v"(pv[i]), "+v"(pv[i + 1]), "+v"(pv[i + 2]), "+v"(pv[i + 3]), "+v"(pv[i + 4]), "+v"(pv[i + 5]),
                   "+v"(pv[i + 6]), "+v"(pv[i + 7]));
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) hacc[rt][reg] += __builtin_bit_cast(float, pv[rt * 16 + reg]);
  }
  // scale, mask rank rows >= r; the saved copy [M, 64] carries 1.0 in column 63 when free (the dbias trick of the
  // weight-gradient kernels); one 32-rank tile per half, 16-byte pieces
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    if (rt < NT) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        hacc[rt][reg] = r < rb ? hacc[rt][reg] * p.scale : 0.f;
      }
    }
    if (p.Hsave && hh == rt && tok < p.M) {
      float* Hs = p.Hsave + tok * 64 + rt * 32 + 4 * lh;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rt < NT) v = (f32x4){hacc[rt][4 * rq + 0], hacc[rt][4 * rq + 1], hacc[rt][4 * rq + 2], hacc[rt][4 * rq + 3]};
        if (rt == 1 && rq == 3 && lh == 1 && rb < 64) v[3] = 1.0f;   // column 63 <- 1.0
        *(f32x4*)(Hs + 8 * rq) = v;
      }
    }
  }
  if (nsl == 0) return;

  // ================================================================== phase 2: Y = H . F2 (column tile hh of every slice)
  // k-step s contracts ranks 16 s + 8 (j >> 2) + 4 lh + (j & 3), j = 0 .. 7 -- the order the accumulator registers hold them
  u32x4 hp[KS][3];
#pragma unroll
  for (int s4 = 0; s4 < KS; ++s4)
#pragma unroll
    for (int pr = 0; pr < 4; ++pr)
      split3v(hacc[s4 >> 1][8 * (s4 & 1) + 2 * pr], hacc[s4 >> 1][8 * (s4 & 1) + 2 * pr + 1], hp[s4], pr);
  const uint32_t fo2 = (uint32_t)((32 * hh + li) * 128 + ((lh ^ ((li >> 1) & 7)) * 16));
  const bool rows_full = m0 + C4_BM <= p.M;
  const bool plain = rows_full;                            // unpredicated stores wherever the slice is full
  const bool has_beta = p.beta != 0.f;                     // Y is accumulated onto (accumulator term written by an earlier launch)
  // per-lane byte offset of (token 4 lh, column 32 hh + li) from the wave's row base; register reg adds row (reg & 3) + 8 (reg >> 2)
  const float* ybase = p.Y + tok0 * p.ldy;                 // uniform per wave
  const uint32_t voff0 = (uint32_t)((4 * lh * p.ldy + 32 * hh + li) * 4);
  f32x16 yacc;
  bool prev_plain = false;
  auto stores = [&](int sl, bool fast) {
    const int col = sl * 64 + 32 * hh + li;
    float bv = 0.f;
    if (p.bias && col < D2) bv = p.bias[col];
    if (fast) {
      const float* sb = ybase + sl * 64;
      if (has_beta) {   // the sixteen old values first (one 128-byte row segment per register, as the stores), then the stores
        float old[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2);
          old[reg] = *(const float*)((const char*)sb + voff0 + (uint32_t)(r * (int)p.ldy * 4));
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) yacc[reg] += p.beta * old[reg];
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = (reg & 3) + 8 * (reg >> 2);
        c4_store_nt(voff0 + (uint32_t)(r * (int)p.ldy * 4), yacc[reg] + bv, sb);
      }
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        const int64_t tk = tok0 + r;
        if (tk < p.M && col < D2) {
          float* dst = p.Y + tk * p.ldy + col;
          float o = yacc[reg] + bv;
          if (p.beta != 0.f) o += p.beta * *dst;
          __builtin_nontemporal_store(o, dst);
        }
      }
    }
  };
#pragma unroll 1
  for (int sl = 0; sl < nsl; ++sl) {
    const int c = nst + sl;
    // this wave's pieces of chunk c have landed; the 16 stores of the previous slice (issued after that DMA) may still fly
    if (sl >= 2 && prev_plain) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    raw_barrier();
    if (c + 1 < ntot) issue_f(c + 1);
    const uint32_t fs = lds0 + (uint32_t)((c & 1) * C4_FSLOT);
    u32x4 f2[KS][3];
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) {
      const uint32_t fa = fs + (fo2 ^ (uint32_t)(32 * s4));   // k-step: bits 5-6 of the chunk offset
      DS_READ_B128(f2[s4][0], fa, 0 * C4_PLANE);
      DS_READ_B128(f2[s4][1], fa, 1 * C4_PLANE);
      DS_READ_B128(f2[s4][2], fa, 2 * C4_PLANE);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (sl > 0) {
      const bool fast = plain && (sl * 64 <= D2);   // slice sl - 1 is full
      stores(sl - 1, fast);
      prev_plain = fast && !p.bias && !has_beta;    // (bias / old-Y loads add compiler-counted operations: wait for everything)
    }
    LGKM_WAIT0();
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) asm volatile("" : "+v"(f2[s4][0]), "+v"(f2[s4][1]), "+v"(f2[s4][2]));
#pragma unroll
    for (int i = 0; i < 16; ++i) yacc[i] = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) yacc = mfma_x3(hp[s4], f2[s4], yacc);
    __builtin_amdgcn_sched_barrier(0);
  }
  stores(nsl - 1, plain && (nsl * 64 <= D2));
}

// =================================================================================================
size_t chain3f_plane_bytes(int d_in, int d_out) { return (size_t)((d_in + 63) / 64 + (d_out + 63) / 64) * C4_FSLOT; }

bool chain3f_supported(const ChainParams& p, int dtype) {
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  auto a4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; };
  if (dtype != SOW_F32 || p.ra != 0 || p.rb < 1 || p.rb > 64) return false;
  if (sw_on(SW_F32_EXACT) || sw_on(SW_NO_CHAIN3F) || sw_on(SW_FORCE_CHAIN_V1)) return false;
  if (!p.planes || p.planes_bytes < chain3f_plane_bytes(p.D1, p.D2) || !a16(p.planes)) return false;
  if (p.ntb > 0 || p.Hpartial || p.Hload || p.pad_dst) return false;
  if (p.D1 % 4 || p.ldx % 4 || !a16(p.X) || (p.D2 > 0 && !a4(p.Y)) || (p.Hsave && !a16(p.Hsave))) return false;
  if (p.M < 8192) return false;   // (shorter inputs: chain2f with the K / column split of api.hip fills the chip better)
  if (p.ldy * 4 * 32 >= (int64_t)1 << 31) return false;   // 32-bit store offsets inside a wave's 32 rows
  return true;
}

int launch_chain3f(const ChainParams& p, bool bwd, hipStream_t stream) {
  const void* Bp = bwd ? p.F1b : p.F2b;
  const int64_t ldB = bwd ? p.ldf1b : p.ldf2b;
  const void* Ap = bwd ? p.F2b : p.F1b;
  const int64_t ldA = bwd ? p.ldf2b : p.ldf1b;
  if ((reinterpret_cast<uintptr_t>(Bp) & 3) || (reinterpret_cast<uintptr_t>(Ap) & 3) || ldA != p.rb) return SOW_ERR_ALIGN;
  const int nst = (p.D1 + 63) / 64, nsl = (p.D2 + 63) / 64;
  PlaneParams q;
  q.Amat = (const float*)Ap, q.Bmat = (const float*)Bp, q.ldb = ldB;
  q.rb = p.rb, q.D1 = p.D1, q.D2 = p.D2, q.nst = nst, q.bwd = bwd ? 1 : 0;
  q.out = (char*)p.planes;
  hipLaunchKernelGGL(chain3f_planes_kernel, dim3(4 * (nst + nsl)), dim3(128), 0, stream, q);
  SOW_CHECK_LAUNCH();
  Chain3fParams k;
  k.X = (const float*)p.X, k.Y = (float*)p.Y, k.Hsave = (float*)p.Hsave, k.bias = (const float*)p.bias;
  k.planes = (const char*)p.planes;
  k.M = p.M, k.ldx = p.ldx, k.ldy = p.ldy, k.D1 = p.D1, k.D2 = p.D2, k.rb = p.rb;
  k.scale = p.scale, k.beta = p.beta;
  const int grid = ceil_div(p.M, C4_BM);
#define C4_LAUNCH(NT_, KS_)                                                                                  \
  do {                                                                                                       \
    SOW_SET_MAX_LDS_ONCE(C4_LDS, chain3f_kernel<NT_, KS_>);                                                  \
    hipLaunchKernelGGL((chain3f_kernel<NT_, KS_>), dim3(grid), dim3(C4_THREADS), C4_LDS, stream, k);         \
  } while (0)
  if (p.rb <= 16) C4_LAUNCH(1, 1);
  else if (p.rb <= 32) C4_LAUNCH(1, 2);
  else if (p.rb <= 48) C4_LAUNCH(2, 3);
  else C4_LAUNCH(2, 4);
#undef C4_LAUNCH
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
