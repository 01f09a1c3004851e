// fp32 weight-gradient partial sums, "quad" variant (3 x bf16 products, gfx950):  G[D, 64] = sum_t M[t, D]^T . S[t, 64]
//
// Same contract and partial-buffer layout as the kernels of skinny_tn.hip (dA = x^T . dh, dB^T = dY^T . h, dbias through the
// all-ones column 63 of S; autograd of tn_gradient/layer/sow.py:117); a different work split.  The wide fp32 kernel there
// gives a workgroup TWO 64-column groups and its own copy of the S stream: at d = 768 the h / dh rows are re-read (from L2)
// six times per operand -- a third of everything the DMA path moves -- and every wave reads, splits and multiplies one
// after the other (counters at the north-star point: matrix pipe 30 %, VALU 32 %, 6 waves on 4 SIMDs).  Here
//   * a workgroup = 8 waves owns FOUR column groups (256 columns) of a token slab: wave (cg = w & 3, th = w >> 2) takes column
//     group cg and the token half th of every 32-token stage -- the two waves of a SIMD work on the same columns, two k-steps
//     apart -- so S is fetched once per 256 columns (3 x at d = 768) and shared through LDS;
//   * a stage is 32 tokens: eight M images [16 tok][64 col] fp32 (one per wave, 4 DMA instructions each) and one S image
//     [32 tok][64] (one instruction per wave), 40 KiB, three slots, one raw s_barrier per stage, counted vmcnt;
//   * software pipeline across the barrier: after barrier i a wave issues the LDS reads of its k-step of stage i, then runs the
//     24 MFMAs of stage i - 1 with the 3 x bf16 split of the new fragments spread over the MFMA slots; the S fragment of a
//     token half is split ONCE -- a quarter by each of its four waves, exchanged through an LDS image -- not once per wave
//     (90 instead of 144 split instructions per wave and stage);
//   * accumulators as in the wide kernel (lane li owns columns 2 li, 2 li + 1 of both operands' images); the two token halves
//     are summed through LDS at the end of the slab, in a fixed order (deterministic).
// Taken for fp32 inputs with T >= 4096 and at least 3 column groups per operand (tn_f32q_shape_ok); tn_pick_slabs then plans
// the slab count for one resident round of these workgroups.
#include "kernels.hpp"
#include "lds_dma.hpp"

namespace sow {

constexpr int TN_BD = 64;            // columns per column group (as in skinny_tn.hip)
constexpr int TNQ_WAVES = 8;
constexpr int TNQ_DEPTH = 3;
constexpr int TNQ_STAGE = 8 * 4096 + 8192;     // 40 KiB
constexpr int TNQ_LDS = 147456;                // 144 KiB: three stages (120 KiB) + two shared S-plane images (24 KiB); the end-of-slab sum stages 8 x 16 KiB

// 3 x bf16 split of one fragment read as eight ds_read_b64 (token j -> the lane's two columns): the two exact residual
// subtractions run on the (column 0, column 1) pair of each token -- the registers a b64 read delivers together -- and the
// v_perm_b32 that packs two tokens' high halves into one dword picks its operands freely, so no register moves are needed
// (pairing consecutive tokens of ONE column, as split3v does, costs two v_mov per pair here).
// (Integer-typed registers on purpose: hipcc 7.2 folds element 1 of a FLOAT vector into element 0 when the elements are
// bit-cast -- lds_dma.hpp -- and did so here with f32x2 fragments: both columns of the residual came from column 0.)
__device__ __forceinline__ void tnq_residuals(const u32x2 v, uint64_t& x64, uint64_t& r64, uint64_t& q64) {
  constexpr uint32_t MK = 0xffff0000u;
  const uint32_t u0 = v[0], u1 = v[1];
  x64 = (uint64_t)u0 | ((uint64_t)u1 << 32);
  r64 = pk_sub_f32(x64, (uint64_t)(u0 & MK) | ((uint64_t)(u1 & MK) << 32));
  const uint32_t r0 = (uint32_t)r64, r1 = (uint32_t)(r64 >> 32);
  q64 = pk_sub_f32(r64, (uint64_t)(r0 & MK) | ((uint64_t)(r1 & MK) << 32));
}
// tokens 2 pr, 2 pr + 1 of both columns -> element pr of the six plane vectors
__device__ __forceinline__ void tnq_split_pair(const u32x2 va, const u32x2 vb, u32x4 (&p0)[3], u32x4 (&p1)[3], int pr) {
  constexpr uint32_t HI = 0x07060302u;   // v_perm_b32(b, a, HI) = (a >> 16) | (b & 0xffff0000)
  uint64_t xa, ra, qa, xb, rb, qb;
  tnq_residuals(va, xa, ra, qa);
  tnq_residuals(vb, xb, rb, qb);
  p0[0][pr] = __builtin_amdgcn_perm((uint32_t)xb, (uint32_t)xa, HI);
  p1[0][pr] = __builtin_amdgcn_perm((uint32_t)(xb >> 32), (uint32_t)(xa >> 32), HI);
  p0[1][pr] = __builtin_amdgcn_perm((uint32_t)rb, (uint32_t)ra, HI);
  p1[1][pr] = __builtin_amdgcn_perm((uint32_t)(rb >> 32), (uint32_t)(ra >> 32), HI);
  p0[2][pr] = __builtin_amdgcn_perm((uint32_t)qb, (uint32_t)qa, HI);
  p1[2][pr] = __builtin_amdgcn_perm((uint32_t)(qb >> 32), (uint32_t)(qa >> 32), HI);
}

__global__ __launch_bounds__(64 * TNQ_WAVES, 2) void tn_partial_f32_quad_kernel(const TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cg = w & 3, th = w >> 2;
  // Block -> (operand, range, slab), XCD-aware: block b runs on XCD b % 8 (round-robin dispatch), so the ranges of ONE slab are
  // given consecutive places in the same XCD's sequence -- they read the same S rows, which then come from that XCD's L2 after
  // the first read (PMC at the north-star point: 252 -> 218 MB read per launch = the algorithmic bytes; the launch time did not
  // change -- the kernel is bound by its own instruction issue, not by that traffic).  Slab counts are padded to a multiple of 8 per
  // operand; the surplus blocks leave at once.
  int b = blockIdx.x, jid = 0;
  const int ns8 = (p.ns + 7) / 8 * 8;
  const int nr0 = (p.job[0].ncg + 3) / 4;
  if (p.njobs > 1 && b >= nr0 * ns8) {
    b -= nr0 * ns8;
    jid = 1;
  }
  const float* Mg = (const float*)(jid ? p.job[1].M : p.job[0].M);
  const float* Sg = (const float*)(jid ? p.job[1].S : p.job[0].S);
  float* Pg = jid ? p.job[1].partial : p.job[0].partial;
  const int64_t ldm = jid ? p.job[1].ldm : p.job[0].ldm;
  const int D = jid ? p.job[1].D : p.job[0].D;
  const int ncg = jid ? p.job[1].ncg : p.job[0].ncg;
  const int nrange = (ncg + 3) / 4;
  const int range = (b >> 3) % nrange, slab = ((b >> 3) / nrange) * 8 + (b & 7);
  if (slab >= p.ns) return;                  // padding block (uniform: before any barrier)
  const int g = range * 4 + cg;              // this wave's column group
  const bool active = g < ncg;
  const int d0 = g * TN_BD;
  const int64_t t_begin = (int64_t)slab * p.slab_len;
  int64_t t_end = t_begin + p.slab_len;
  if (t_end > p.T) t_end = p.T;
  const int nstage = t_begin < t_end ? (int)((t_end - t_begin + 31) / 32) : 0;
  const char* zp = zero_page_for(lane);
  const int drow = lane >> 4, dpc = lane & 15;

  // per-lane DMA sources of stage 0: M instruction q = token rows 16 th + 4 q + drow, S instruction = rows 4 w + drow
  // Lanes whose columns lie beyond D (or whose wave has no column group) read the zero page with a zero step, so that a whole
  // stage -- every one but possibly the last of a slab -- is five DMA instructions and five pointer increments, nothing else.
  const bool mcol_ok = active && d0 + dpc * 4 < D;
  const char* mptr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    mptr[q] = mcol_ok ? (const char*)(Mg + (t_begin + 16 * th + 4 * q + drow) * ldm + d0 + dpc * 4) : zp;
  const char* sptr = (const char*)(Sg + (t_begin + 4 * w + drow) * 64 + dpc * 4);
  const int64_t mstep = mcol_ok ? 32 * ldm * 4 : 0, sstep = 32 * 64 * 4;
  // (a wave without a column group -- the tail of an operand's last range -- runs the same code on zeros: its SIMD has no
  // other work, and the instruction stream stays free of wave-dependent branches around the asm reads)
  auto issue = [&](int i) {   // strictly in order: the pointers are at stage i; 5 DMA instructions per wave and stage
    const int64_t tt0 = t_begin + (int64_t)i * 32;
    char* slot = smem + (i % TNQ_DEPTH) * TNQ_STAGE;
    if (tt0 + 32 <= t_end) {   // wave-uniform
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dma16(mptr[q], slot + w * 4096 + q * 1024);
        mptr[q] += mstep;
      }
      dma16(sptr, slot + 8 * 4096 + w * 1024);
      sptr += sstep;
    } else {                   // the ragged last stage: rows beyond the slab read zeros
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const void* src = (tt0 + 16 * th + 4 * q + drow < t_end) ? (const void*)mptr[q] : (const void*)zp;
        dma16(src, slot + w * 4096 + q * 1024);
        mptr[q] += mstep;
      }
      const void* src = (tt0 + 4 * w + drow < t_end) ? (const void*)sptr : (const void*)zp;
      dma16(src, slot + 8 * 4096 + w * 1024);
      sptr += sstep;
    }
  };

  f32x16 acc[2][2];   // [M column parity a][S column parity c]: register reg of lane l = G[2 * acc_row(reg, l) + a][2 * (l & 31) + c]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const uint32_t lds0 = lds_addr(smem);
  // lane-half lh reads tokens 8 lh .. 8 lh + 7 of the wave's 16: one ds_read_b64 per token (columns 2 li, 2 li + 1)
  const uint32_t moff = (uint32_t)(w * 4096 + lh * 2048 + li * 8);
  const uint32_t soff = (uint32_t)(8 * 4096 + th * 4096 + lh * 2048 + li * 8);

  // S planes are shared: the four waves of a token half (cg = 0 .. 3) each split ONE token pair (pr = cg) of the half's S
  // fragment -- 18 VALU instead of 72 per wave and stage -- and leave their six dwords (2 column parities x 3 planes) in an LDS
  // image [th][parity * 3 + plane][lane] of u32x4, element cg; the image of stage i is complete at barrier i + 1, where every
  // wave reads its six vectors back for the MFMAs of stage i (which it runs during iteration i + 1 anyway).  Two images
  // (12 KiB each) behind the three stages.
  constexpr int TNQ_SPL = TNQ_DEPTH * TNQ_STAGE;                  // 120 KiB
  const uint32_t spl_rd = lds0 + (uint32_t)(TNQ_SPL + th * 6144 + lane * 16);
  char* spl_wr = smem + TNQ_SPL + th * 6144 + lane * 16 + cg * 4;
  {   // image 1 is read before it has been written (the zero fragments ahead of the first stage): clear it
    const int o = (int)threadIdx.x * 16;
    *(u32x4*)(smem + TNQ_SPL + 12288 + o) = (u32x4){0, 0, 0, 0};
    if (o + 8192 < 12288) *(u32x4*)(smem + TNQ_SPL + 12288 + 8192 + o) = (u32x4){0, 0, 0, 0};
  }

  u32x4 mpA[2][3], mpB[2][3];   // M^T fragments (column parity a, plane) of the stage being multiplied / being split
  u32x4 sp[2][3];               // S fragments of the stage being multiplied (read back from the shared image)
  u32x2 sv[2], mv[8];           // raw fp32 bit patterns: token j -> (column 2 li, column 2 li + 1); S: the pair pr = cg only
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) mpA[a][pl] = mpB[a][pl] = (u32x4){0, 0, 0, 0};

  // planes of stage i_ - 1 (6 reads, issued FIRST so that a counted wait releases them early), then the raw fragments of stage i_
#define TNQ_READ(i_)                                                                                      \
  do {                                                                                                    \
    const uint32_t pa__ = spl_rd + (uint32_t)((((i_) + 1) & 1) * 12288);                                  \
    DS_READ_B128(sp[0][0], pa__, 0 * 1024); DS_READ_B128(sp[0][1], pa__, 1 * 1024); DS_READ_B128(sp[0][2], pa__, 2 * 1024); \
    DS_READ_B128(sp[1][0], pa__, 3 * 1024); DS_READ_B128(sp[1][1], pa__, 4 * 1024); DS_READ_B128(sp[1][2], pa__, 5 * 1024); \
  } while (0)
#define TNQ_READ_RAW(i_)                                                                                  \
  do {                                                                                                    \
    const uint32_t sa__ = lds0 + (uint32_t)(((i_) % TNQ_DEPTH) * TNQ_STAGE);                              \
    const uint32_t ss__ = sa__ + soff + (uint32_t)(cg * 512);                                             \
    DS_READ_B64(sv[0], ss__, 0); DS_READ_B64(sv[1], ss__, 256);                                           \
    _Pragma("unroll") for (int j__ = 0; j__ < 8; ++j__) DS_READ_B64(mv[j__], sa__ + moff, j__ * 256);     \
  } while (0)
  // the six plane reads of TNQ_READ have landed (the 10 raw reads behind them may still be in flight)
#define TNQ_WAIT_SP()                                                                                     \
  do {                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(10)"                                                                  \
                 : "+v"(sp[0][0]), "+v"(sp[0][1]), "+v"(sp[0][2]), "+v"(sp[1][0]), "+v"(sp[1][1]), "+v"(sp[1][2]) \
                 :                                                                                        \
                 : "memory");                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
#define TNQ_WAIT_RAW()                                                                                    \
  do {                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                   \
                 : "+v"(sv[0]), "+v"(sv[1]), "+v"(mv[0]), "+v"(mv[1]), "+v"(mv[2]), "+v"(mv[3]), "+v"(mv[4]), "+v"(mv[5]), \
                   "+v"(mv[6]), "+v"(mv[7])                                                               \
                 :                                                                                        \
                 : "memory");                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
  // this wave's share of the S split of stage i_: token pair cg of both column parities -> six dwords into image i_ & 1
#define TNQ_SPLIT_S(i_)                                                                                   \
  do {                                                                                                    \
    u32x4 t0__[3], t1__[3];                                                                               \
    tnq_split_pair(sv[0], sv[1], t0__, t1__, 0);                                                          \
    char* d__ = spl_wr + ((i_) & 1) * 12288;                                                              \
    *(uint32_t*)(d__ + 0 * 1024) = t0__[0][0], *(uint32_t*)(d__ + 1 * 1024) = t0__[1][0], *(uint32_t*)(d__ + 2 * 1024) = t0__[2][0]; \
    *(uint32_t*)(d__ + 3 * 1024) = t1__[0][0], *(uint32_t*)(d__ + 4 * 1024) = t1__[1][0], *(uint32_t*)(d__ + 5 * 1024) = t1__[2][0]; \
  } while (0)
#define TNQ_SPLIT_M(NXT, pr_) tnq_split_pair(mv[2 * (pr_)], mv[2 * (pr_) + 1], NXT[0], NXT[1], (pr_))
  // the 24 MFMAs of the stage whose M planes are CUR and whose S planes are sp
#define TNQ_MF(CUR, i_)                                                                                              \
  do {                                                                                                               \
    constexpr int i__ = (i_);                                                                                        \
    constexpr int a__ = i__ / 12, c__ = (i__ / 6) & 1, k__ = i__ % 6;                                                \
    constexpr int ia__[6] = {2, 0, 1, 1, 0, 0}, ib__[6] = {0, 2, 1, 0, 1, 0};                                        \
    acc[a__][c__] = mfma32(as_bf16x8(CUR[a__][ia__[k__]]), as_bf16x8(sp[c__][ib__[k__]]), acc[a__][c__]);            \
  } while (0)
  // with FILL: the split of the fresh raw fragments of stage i_ (S share + four M items) is spread over MFMA slots 3 .. 19
#define TNQ_STEP(CUR, NXT, FILL, i_)                                                                     \
  do {                                                                                                   \
    TNQ_MF(CUR, 0); __builtin_amdgcn_sched_barrier(0);                                                   \
    TNQ_MF(CUR, 1); __builtin_amdgcn_sched_barrier(0);                                                   \
    TNQ_MF(CUR, 2); if (FILL) TNQ_WAIT_RAW(); __builtin_amdgcn_sched_barrier(0);                         \
    TNQ_MF(CUR, 3); if (FILL) TNQ_SPLIT_S(i_); __builtin_amdgcn_sched_barrier(0);                        \
    TNQ_MF(CUR, 4); __builtin_amdgcn_sched_barrier(0);                                                   \
    TNQ_MF(CUR, 5); __builtin_amdgcn_sched_barrier(0);                                                   \
    TNQ_MF(CUR, 6); __builtin_amdgcn_sched_barrier(0);                                                   \
    TNQ_MF(CUR, 7); if (FILL) TNQ_SPLIT_M(NXT, 0); __builtin_amdgcn_sched_barrier(0);                    \
    TNQ_MF(CUR, 8); __builtin_amdgcn_sched_barrier(0);                                                   \
    TNQ_MF(CUR, 9); __builtin_amdgcn_sched_barrier(0);                                                   \
    TNQ_MF(CUR, 10); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 11); if (FILL) TNQ_SPLIT_M(NXT, 1); __builtin_amdgcn_sched_barrier(0);                   \
    TNQ_MF(CUR, 12); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 13); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 14); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 15); if (FILL) TNQ_SPLIT_M(NXT, 2); __builtin_amdgcn_sched_barrier(0);                   \
    TNQ_MF(CUR, 16); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 17); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 18); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 19); if (FILL) TNQ_SPLIT_M(NXT, 3); __builtin_amdgcn_sched_barrier(0);                   \
    TNQ_MF(CUR, 20); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 21); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 22); __builtin_amdgcn_sched_barrier(0);                                                  \
    TNQ_MF(CUR, 23); __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
  // stage i: wait for this wave's DMA, barrier (which also publishes the S planes of stage i - 1), the plane reads, refill the
  // ring, the raw reads of stage i, multiply stage i - 1 while stage i is split
#define TNQ_ITER(i_, CUR, NXT)                                                                           \
  do {                                                                                                   \
    const int newer__ = (nstage - 1 - (i_)) < (TNQ_DEPTH - 2) ? (nstage - 1 - (i_)) : (TNQ_DEPTH - 2);   \
    if (newer__ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  \
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    raw_barrier();                                                                                       \
    TNQ_READ(i_);                                                                                        \
    if ((i_) + TNQ_DEPTH - 1 < nstage) issue((i_) + TNQ_DEPTH - 1);                                      \
    TNQ_READ_RAW(i_);                                                                                    \
    TNQ_WAIT_SP();                                                                                       \
    TNQ_STEP(CUR, NXT, true, i_);                                                                        \
  } while (0)

  const int pre = nstage < TNQ_DEPTH - 1 ? nstage : TNQ_DEPTH - 1;
  for (int i = 0; i < pre; ++i) issue(i);
  int i = 0;
#pragma unroll 1
  for (; i + 1 < nstage; i += 2) {
    TNQ_ITER(i, mpA, mpB);       // mpA holds stage i - 1 (zeros before the first stage), mpB receives stage i
    TNQ_ITER(i + 1, mpB, mpA);
  }
  if (i < nstage) {
    TNQ_ITER(i, mpA, mpB);
    ++i;
  }
  // the last stage (i - 1): its S planes are complete after one more barrier
  raw_barrier();
  TNQ_READ(i);
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(sp[0][0]), "+v"(sp[0][1]), "+v"(sp[0][2]), "+v"(sp[1][0]), "+v"(sp[1][1]), "+v"(sp[1][2])
               :
               : "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (i & 1) TNQ_STEP(mpB, mpA, false, 0);   // an odd number of stages left the last one in mpB
  else TNQ_STEP(mpA, mpB, false, 0);
#undef TNQ_ITER
#undef TNQ_STEP
#undef TNQ_MF
#undef TNQ_SPLIT_M
#undef TNQ_SPLIT_S
#undef TNQ_WAIT_RAW
#undef TNQ_WAIT_SP
#undef TNQ_READ_RAW
#undef TNQ_READ

  // ------------------------------------------------------------------ end of slab: sum the two token halves, store
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();   // every wave is done with the rings
  float* red = (float*)smem + w * 4096;   // [64][64] fp32 per wave
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) red[(2 * acc_row(reg, lane) + a) * 64 + 2 * li + c] = acc[a][c][reg];
  __syncthreads();
  const float* redall = (const float*)smem;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int v = t + 64 * TNQ_WAVES * k;   // 4 tiles x 1024 float4
    const int tile = v >> 10, e = v & 1023;
    const int gg = range * 4 + tile;
    if (gg < ncg) {
      const f32x4 s0 = *(const f32x4*)(redall + tile * 4096 + e * 4);
      const f32x4 s1 = *(const f32x4*)(redall + (tile + 4) * 4096 + e * 4);
      float* P = Pg + ((int64_t)slab * ncg * TN_BD + (int64_t)gg * TN_BD) * 64;
      *(f32x4*)(P + e * 4) = s0 + s1;
    }
  }
}

// shapes the quad kernel is planned for (tn_pick_slabs) -- alignment is checked at launch
bool tn_f32q_shape_ok(int64_t T, int d_in, int d_out) {
  if (sw_on(SW_F32_EXACT) || sw_on(SW_TN_NARROW) || sw_on(SW_NO_TN_F32Q)) return false;
  return T >= 4096 && (d_in + 63) / 64 >= 3 && (d_out + 63) / 64 >= 3;
}

bool tn_f32q_ok(const TnParams& p) {
  if (p.njobs != 2 || p.slab_len % 32 || p.ns < 1) return false;
  if (!tn_f32q_shape_ok(p.T, p.job[0].D, p.job[1].D)) return false;
  for (int j = 0; j < p.njobs; ++j) {
    const TnJob& J = p.job[j];
    if (J.D % 4 || J.ldm % 4 || (reinterpret_cast<uintptr_t>(J.M) & 15) || (reinterpret_cast<uintptr_t>(J.S) & 15) || !J.ones_col_in_s)
      return false;
  }
  return true;
}

int launch_tn_f32q(const TnParams& p, hipStream_t stream) {
  int blocks = 0;
  for (int j = 0; j < p.njobs; ++j) blocks += (p.job[j].ncg + 3) / 4 * ((p.ns + 7) / 8 * 8);
  if (blocks <= 0) return SOW_OK;
  SOW_SET_MAX_LDS_ONCE(TNQ_LDS, tn_partial_f32_quad_kernel);
  hipLaunchKernelGGL(tn_partial_f32_quad_kernel, dim3(blocks), dim3(64 * TNQ_WAVES), TNQ_LDS, stream, p);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
