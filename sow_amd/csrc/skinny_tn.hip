// Tall-skinny transposed products  G[D, 64] = sum_t M[t, D]^T . S[t, 64]   (K = tokens)
//
// These are the weight gradients of SoWLinear (autograd of tn_gradient/layer/sow.py:117):
//   dA   = x^T  . dh        M = x  [T, d_in],  S = dh = s*dY.B^T  [T, 64] (saved by the chain kernel)
//   dB^T = dY^T . h         M = dY [T, d_out], S = h  = x.A       [T, 64]
//   dbias = colsum(dY)      obtained for free as an extra all-ones column of S (column 63).
// The reduction runs over T (tens of thousands) into a tiny [D, r] output, so the grid is
// (column groups of 64) x (token slabs); every workgroup keeps its [64, 64] fp32 partial in MFMA
// accumulators over its whole slab and writes it once to the workspace; tn_reduce sums the slabs in a
// fixed order (deterministic, no float atomics), applies alpha/beta, crops 64 -> r and writes
// [D, r] or its transpose [r, D].
// Column groups of one slab are NS blocks apart (NS % 8 == 0) so they share an XCD and S is served
// from that XCD's L2 after the first read.
// Kernels in this file: tn_partial_kernel (generic, any shape / dtype), tn_partial_dma_kernel (bf16, one column group per
// block), tn_partial_dma_wide_kernel (bf16, two column groups per block; grouped + persistent: single layers and small
// groups), tn_partial_rows_kernel (bf16, a workgroup owns ALL columns of a token slab: whole decoder blocks, slab counts
// planned over the group), the fp32 forms (exact and 3 x bf16), and tn_reduce (fixed-order slab sum).
#include "kernels.hpp"
#include "lds_dma.hpp"
#include <cstdlib>

namespace sow {


constexpr int TN_BD = 64;

template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> {
  static constexpr int BT = 64;  // tokens per staged chunk
};
template <> struct TnCfg<float> {
  static constexpr int BT = 32;
};

template <typename T> __global__ __launch_bounds__(256, 2) void tn_partial_kernel(const TnParams p) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int BT = TnCfg<T>::BT;
  // images: bf16 -> Mi[d][t], Si[r][t] (k = t contiguous, swizzled); f32 -> Mi[t][d], Si[t][r]
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 64 * (F32 ? 2 : 2)];
  char* Mi = smem;
  char* Si = smem + 8192;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int dm = w >> 1, rn = w & 1, li = lane & 31, lh = lane >> 5;

  // block -> (job, column group, slab); same-slab blocks are ns apart
  int b = blockIdx.x;
  int jid = 0;
  if (p.njobs > 1 && b >= p.job[0].ncg * p.ns) {
    b -= p.job[0].ncg * p.ns;
    jid = 1;
  }
  const TnJob& J = p.job[jid];
  const int cg = b / p.ns, slab = b % p.ns;
  const int d0 = cg * TN_BD;
  const int64_t t_begin = (int64_t)slab * p.slab_len;
  int64_t t_end = t_begin + p.slab_len;
  if (t_end > p.T) t_end = p.T;
  const T* Mg = (const T*)J.M;
  const T* Sg = (const T*)J.S;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // staging registers
  uint32_t md[8], sd[8];       // bf16 dword path
  u32x4 mv[2], sv[2];          // f32 vector path
  T me[16];                    // generic M path (64*BT/256 elements: 16 bf16 / 8 f32)

  auto load = [&](int64_t tt0) {
    if constexpr (!F32) {
      const int dp = t & 31, to = t >> 5;  // (column pair, token octet)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t tt = tt0 + to * 8 + j;
        const bool tv = tt < t_end;
        sd[j] = tv ? *(const uint32_t*)(Sg + tt * 64 + 2 * dp) : 0u;
        if (J.ones_col >= 0 && (J.ones_col >> 1) == dp && tv) {
          // replace element ones_col of the pair by bf16(1.0) = 0x3F80
          sd[j] = (J.ones_col & 1) ? ((sd[j] & 0xffffu) | 0x3F800000u) : ((sd[j] & 0xffff0000u) | 0x3F80u);
        }
      }
      if (J.vec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t tt = tt0 + to * 8 + j;
          const int d = d0 + 2 * dp;
          md[j] = (tt < t_end && d < J.D) ? *(const uint32_t*)(Mg + tt * J.ldm + d) : 0u;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          const int64_t tt = tt0 + tl;
          me[i] = (tt < t_end && d0 + d < J.D) ? Mg[tt * J.ldm + d0 + d] : (bf16_t)0.f;
        }
      }
    } else {
      // f32: chunk [32 t][64 cols]; 512 vectors of 16 B, two per thread
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = t + 256 * i, tl = v >> 4, c = v & 15;
        const int64_t tt = tt0 + tl;
        sv[i] = tt < t_end ? *(const u32x4*)(Sg + tt * 64 + c * 4) : u32x4{0, 0, 0, 0};
        if (J.ones_col >= 0 && (J.ones_col >> 2) == c && tt < t_end)
          sv[i][J.ones_col & 3] = __builtin_bit_cast(uint32_t, 1.0f);
      }
      if (J.vec) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int v = t + 256 * i, tl = v >> 4, c = v & 15;
          const int64_t tt = tt0 + tl;
          const int d = d0 + c * 4;
          mv[i] = (tt < t_end && d < J.D) ? *(const u32x4*)(Mg + tt * J.ldm + d) : u32x4{0, 0, 0, 0};
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          const int64_t tt = tt0 + tl;
          me[i] = (tt < t_end && d0 + d < J.D) ? Mg[tt * J.ldm + d0 + d] : 0.f;
        }
      }
    }
  };
  auto store = [&]() {
    if constexpr (!F32) {
      const int dp = t & 31, to = t >> 5;
      u32x4 c0, c1;
      transpose_8x2(sd, c0, c1);
      *(u32x4*)(Si + bf16_img_off<BT>(2 * dp, to)) = c0;
      *(u32x4*)(Si + bf16_img_off<BT>(2 * dp + 1, to)) = c1;
      if (J.vec) {
        transpose_8x2(md, c0, c1);
        *(u32x4*)(Mi + bf16_img_off<BT>(2 * dp, to)) = c0;
        *(u32x4*)(Mi + bf16_img_off<BT>(2 * dp + 1, to)) = c1;
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          *(bf16_t*)(Mi + bf16_img_off<BT>(d, tl >> 3) + (tl & 7) * 2) = me[i];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = t + 256 * i, tl = v >> 4, c = v & 15;
        *(u32x4*)(Si + (tl * 64 + c * 4) * 4) = sv[i];
      }
      if (J.vec) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int v = t + 256 * i, tl = v >> 4, c = v & 15;
          *(u32x4*)(Mi + (tl * 64 + c * 4) * 4) = mv[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = t + 256 * i, d = e & 63, tl = e >> 6;
          ((float*)Mi)[tl * 64 + d] = me[i];
        }
      }
    }
  };

  if (t_begin < t_end) load(t_begin);
  for (int64_t tt0 = t_begin; tt0 < t_end; tt0 += BT) {
    store();
    __syncthreads();
    if (tt0 + BT < t_end) load(tt0 + BT);
    if constexpr (F32) {
      const float* ms = (const float*)Mi + lh * 64 + dm * 32 + li;
      const float* ss = (const float*)Si + lh * 64 + rn * 32 + li;
#pragma unroll
      for (int ks = 0; ks < BT / 2; ++ks) acc = mfma32(ms[2 * ks * 64], ss[2 * ks * 64], acc);
    } else {
#pragma unroll
      for (int ks = 0; ks < BT / 16; ++ks) {
        const bf16x8 a = *(const bf16x8*)(Mi + bf16_img_off<BT>(dm * 32 + li, 2 * ks + lh));
        const bf16x8 bb = *(const bf16x8*)(Si + bf16_img_off<BT>(rn * 32 + li, 2 * ks + lh));
        acc = mfma32(a, bb, acc);
      }
    }
    __syncthreads();
  }

  // partial[slab][d0 + row][col]  (Dpad = ncg * 64 rows)
  float* P = J.partial + ((int64_t)slab * J.ncg * TN_BD + d0) * 64;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) P[(dm * 32 + acc_row(reg, lane)) * 64 + rn * 32 + li] = acc[reg];
}

// =================================================================================================
// bf16 fast path: wave-private LDS-DMA rings + transposed LDS reads (gfx950).
//
// Each of the 4 waves streams its own 16-token groups (group g of the slab goes to wave g % 4, so the
// workgroup as a whole reads contiguous memory): per group two 1-KiB `global_load_lds_dwordx4`
// instructions for the M rows (16 tokens x 64 columns) and two for the S rows, straight into the
// wave's ring slot -- no VGPR staging, no workgroup barrier, DEPTH groups in flight per wave
// (counted s_waitcnt vmcnt).  The tiles stay in their natural [token][column] layout; the MFMA
// operands need the contraction index (token) down the register, which `ds_read_b64_tr_b16`
// delivers (4 tokens x 16 columns per 16-lane group, verified by tools/probe.hip).  A 16-byte-chunk
// XOR (chunk ^ 4*((row>>1)&1)), applied on the DMA source address and again on the read address,
// spreads the four rows of a transposed read over all 64 banks.
// Every wave accumulates the full 64x64 tile over its groups; the 4 wave partials are summed
// through LDS once per slab.
// =================================================================================================
typedef __attribute__((ext_vector_type(4))) short s16x4;


constexpr int TN_DEPTH = 4;                 // 16-token groups in flight per wave
constexpr int TN_STAGE_BYTES = 4096;        // [16][64] bf16 M tile + [16][64] bf16 S tile

__device__ __forceinline__ void tn_wait_stages(int newer) {
  // wait until all but the `newer` most recent stages (4 DMA instructions each) have landed
  switch (newer) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

__global__ __launch_bounds__(256, 2) void tn_partial_dma_kernel(const TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // wave id as a scalar: keeps the pipeline control flow uniform
  int b = blockIdx.x, jid = 0;
  if (p.njobs > 1 && b >= p.job[0].ncg * p.ns) {
    b -= p.job[0].ncg * p.ns;
    jid = 1;
  }
  // copy the job into scalars once (a reference into the kernel-argument block with a runtime index
  // makes the compiler re-load it inside the loop)
  const bf16_t* Mg = (const bf16_t*)(jid ? p.job[1].M : p.job[0].M);
  const bf16_t* Sg = (const bf16_t*)(jid ? p.job[1].S : p.job[0].S);
  float* Pg = jid ? p.job[1].partial : p.job[0].partial;
  const int64_t ldm = jid ? p.job[1].ldm : p.job[0].ldm;
  const int D = jid ? p.job[1].D : p.job[0].D;
  const int ncg = jid ? p.job[1].ncg : p.job[0].ncg;
  const int cg = b / p.ns, slab = b % p.ns;
  const int d0 = cg * TN_BD;
  const int64_t t_begin = (int64_t)slab * p.slab_len;
  int64_t t_end = t_begin + p.slab_len;
  if (t_end > p.T) t_end = p.T;
  char* ring = smem + w * (TN_DEPTH * TN_STAGE_BYTES);

  const int ngroups = t_begin < t_end ? (int)((t_end - t_begin + 15) / 16) : 0;
  const int nw = ngroups > w ? (ngroups - w + 3) / 4 : 0;  // groups of this wave

  // DMA lane geometry: lane -> (row in 8-row block, physical 16-byte chunk); logical chunk un-swizzled.
  // Per-lane running source pointers, advanced by a per-lane constant in issue order (2 VALU per DMA); only a
  // group that crosses t_end takes the checked path.
  const int drow = lane >> 3, dpc = lane & 7;
  const char* zp = zero_page_for(lane);
  const char* pm[2];
  const char* ps[2];
  int64_t m_step[2];
  const int64_t s_step = (int64_t)64 * 64 * 2;   // 4 waves x 16 tokens per step of this wave
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = half * 8 + drow;
    const int lc = dpc ^ (((row >> 1) & 1) << 2);
    const int64_t tt = t_begin + (int64_t)w * 16 + row;
    const bool ok = d0 + lc * 8 < D;
    pm[half] = ok ? (const char*)(Mg + tt * ldm + d0 + lc * 8) : zp;
    m_step[half] = ok ? (int64_t)64 * ldm * 2 : 0;
    ps[half] = (const char*)(Sg + tt * 64 + lc * 8);
  }
  auto issue = [&](int i) {  // i-th group of this wave -> ring slot i % DEPTH (groups are issued in order)
    const int64_t tt0 = t_begin + (int64_t)(w + 4 * i) * 16;
    char* slot = ring + (i % TN_DEPTH) * TN_STAGE_BYTES;
    const bool whole = tt0 + 16 <= t_end;   // wave-uniform
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const void* srcM = pm[half];
      const void* srcS = ps[half];
      if (!whole && tt0 + half * 8 + drow >= t_end) srcM = zp, srcS = zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcM,
                                       (__attribute__((address_space(3))) void*)(slot + half * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcS,
                                       (__attribute__((address_space(3))) void*)(slot + 2048 + half * 1024), 16, 0, 0);
      pm[half] += m_step[half];
      ps[half] += s_step;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.f;

  // transposed-read lane geometry (see header): group g = lane>>4, q = row inside the 4-row block,
  // pp = 4-column piece; h = k half of the MFMA operand
  const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3, h = g >> 1;
  int roff[2][2];  // [tile (0,1)][read (rows 8h+q, 8h+4+q)] byte offset inside a [16][64] bf16 tile
#pragma unroll
  for (int tile = 0; tile < 2; ++tile)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int row = 8 * h + 4 * rd + q;
      const int col = tile * 32 + 16 * (g & 1) + 4 * pp;
      const int pc = (col >> 3) ^ (((row >> 1) & 1) << 2);
      roff[tile][rd] = row * 128 + pc * 16 + (col & 7) * 2;
    }

  // LDS byte addresses for the inline-asm transposed reads.  The reads are inline asm on purpose:
  // for a compiler-visible LDS read hipcc inserts `s_waitcnt vmcnt(0)` (LDS-DMA may alias), which
  // would drain the ring every step; here the counted vmcnt above is the only DMA wait.
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  const int pre = nw < TN_DEPTH ? nw : TN_DEPTH;
  for (int i = 0; i < pre; ++i) issue(i);
  for (int i = 0; i < nw; ++i) {
    const int newer = (nw - 1 - i) < (TN_DEPTH - 1) ? (nw - 1 - i) : (TN_DEPTH - 1);
    tn_wait_stages(newer);
    const uint32_t slot_addr = ring_addr + (uint32_t)((i % TN_DEPTH) * TN_STAGE_BYTES);
    const uint32_t ad0 = slot_addr + (uint32_t)roff[0][0], ad1 = slot_addr + (uint32_t)roff[1][0];
    u32x2 a00, a01, a10, a11, b00, b01, b10, b11;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:512\n\t"
        "ds_read_b64_tr_b16 %2, %9\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:512\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:2048\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:2560\n\t"
        "ds_read_b64_tr_b16 %6, %9 offset:2048\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:2560\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a00), "=&v"(a01), "=&v"(a10), "=&v"(a11), "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11)
        : "v"(ad0), "v"(ad1)
        : "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 af[2], bfr[2];
    af[0] = __builtin_bit_cast(bf16x8, (u32x4){a00[0], a00[1], a01[0], a01[1]});
    af[1] = __builtin_bit_cast(bf16x8, (u32x4){a10[0], a10[1], a11[0], a11[1]});
    bfr[0] = __builtin_bit_cast(bf16x8, (u32x4){b00[0], b00[1], b01[0], b01[1]});
    bfr[1] = __builtin_bit_cast(bf16x8, (u32x4){b10[0], b10[1], b11[0], b11[1]});
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][c] = mfma32(af[a], bfr[c], acc[a][c]);
    // the slot's reads have returned (lgkmcnt(0) inside the asm), so it may be refilled now
    __builtin_amdgcn_sched_barrier(0);
    if (i + TN_DEPTH < nw) issue(i + TN_DEPTH);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // cross-wave sum of the four [64][64] fp32 partials through LDS (aliases the rings), then one
  // coalesced 16-byte store per thread-quad row to partial[slab][d0 + row][col]
  float* red = (float*)smem;  // [4 waves][64][64]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        red[w * 4096 + (a * 32 + acc_row(reg, lane)) * 64 + c * 32 + (lane & 31)] = acc[a][c][reg];
  __syncthreads();
  float* P = Pg + ((int64_t)slab * ncg * TN_BD + d0) * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = t + 256 * it;  // 1024 float4 of the [64][64] tile
    f32x4 s0 = *(const f32x4*)(red + v * 4);
    const f32x4 s1 = *(const f32x4*)(red + 4096 + v * 4);
    const f32x4 s2 = *(const f32x4*)(red + 8192 + v * 4);
    const f32x4 s3 = *(const f32x4*)(red + 12288 + v * 4);
    s0 = (s0 + s1) + (s2 + s3);
    *(f32x4*)(P + v * 4) = s0;
  }
}

// =================================================================================================
// bf16 wide variant: 8 waves, TWO 64-column groups of M per workgroup.
//
// With one column group per block the S stream (h / dh rows, served by L2) is as large as the M stream (HBM):
// 134 MB of LDS-DMA for 67 MB of HBM on a 512/512 layer, and the kernel is bound by the DMA path as a whole.
// Here every wave streams [16 tok x 128 col] of M and [16 tok x 64] of S per group (6 KiB, 4 + 2 DMA
// instructions, 3 ring slots), i.e. half the S bytes per M byte; 8 waves (group g -> wave g % 8) keep the slab count
// -- and with it the fp32 partial traffic -- where it was.  144 KiB of LDS, one workgroup per CU.
// =================================================================================================
constexpr int TNW_DEPTH = 3;
constexpr int TNW_STAGE_BYTES = 6144;       // [16][64] bf16 M tile x 2 + [16][64] bf16 S tile
constexpr int TNW_WAVES = 8;

__device__ __forceinline__ void tnw_wait_stages(int newer) {   // 6 DMA instructions per stage
  switch (newer) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

__device__ __forceinline__ void tnw_block(const TnParams& p, int b, char* smem, const int t, const int lane, const int w) {
  int jid = 0;
  const int ncg2_0 = (p.job[0].ncg + 1) / 2;
  if (p.njobs > 1 && b >= ncg2_0 * p.ns) {
    b -= ncg2_0 * p.ns;
    jid = 1;
  }
  const bf16_t* Mg = (const bf16_t*)(jid ? p.job[1].M : p.job[0].M);
  const bf16_t* Sg = (const bf16_t*)(jid ? p.job[1].S : p.job[0].S);
  float* Pg = jid ? p.job[1].partial : p.job[0].partial;
  const int64_t ldm = jid ? p.job[1].ldm : p.job[0].ldm;
  const int D = jid ? p.job[1].D : p.job[0].D;
  const int ncg = jid ? p.job[1].ncg : p.job[0].ncg;
  const int dcg = b / p.ns, slab = b % p.ns;
  const int d0 = dcg * 2 * TN_BD;
  const int64_t t_begin = (int64_t)slab * p.slab_len;
  int64_t t_end = t_begin + p.slab_len;
  if (t_end > p.T) t_end = p.T;
  char* ring = smem + w * (TNW_DEPTH * TNW_STAGE_BYTES);

  const int ngroups = t_begin < t_end ? (int)((t_end - t_begin + 15) / 16) : 0;
  const int nw = ngroups > w ? (ngroups - w + TNW_WAVES - 1) / TNW_WAVES : 0;  // groups of this wave

  // DMA sources: instruction q = 2 * tile + half (tile 0, 1 = the two M column groups, tile 2 = S); running per-lane
  // pointers as in the narrow kernel
  const int drow = lane >> 3, dpc = lane & 7;
  const char* zp = zero_page_for(lane);
  const char* ptr[6];
  int64_t step[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int tile = q >> 1, half = q & 1;
    const int row = half * 8 + drow;
    const int lc = dpc ^ (((row >> 1) & 1) << 2);
    const int64_t tt = t_begin + (int64_t)w * 16 + row;
    if (tile < 2) {
      const bool ok = d0 + tile * TN_BD + lc * 8 < D;
      ptr[q] = ok ? (const char*)(Mg + tt * ldm + d0 + tile * TN_BD + lc * 8) : zp;
      step[q] = ok ? (int64_t)TNW_WAVES * 16 * ldm * 2 : 0;
    } else {
      ptr[q] = (const char*)(Sg + tt * 64 + lc * 8);
      step[q] = (int64_t)TNW_WAVES * 16 * 64 * 2;
    }
  }
  auto issue = [&](int i) {
    const int64_t tt0 = t_begin + (int64_t)(w + TNW_WAVES * i) * 16;
    char* slot = ring + (i % TNW_DEPTH) * TNW_STAGE_BYTES;
    const bool whole = tt0 + 16 <= t_end;   // wave-uniform
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const void* src = ptr[q];
      if (!whole && tt0 + (q & 1) * 8 + drow >= t_end) src = zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
      ptr[q] += step[q];
    }
  };

  f32x16 acc[4][2];   // [2 * column group + row tile][S tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.f;

  const int g = lane >> 4, jj = lane & 15, q4 = jj >> 2, pp = jj & 3, h = g >> 1;
  uint32_t roff[2];   // [tile within a [16][64] image] byte offset of the read of rows 8h + q (+512: rows 8h + 4 + q)
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    const int row = 8 * h + q4;
    const int col = tile * 32 + 16 * (g & 1) + 4 * pp;
    const int pc = (col >> 3) ^ (((row >> 1) & 1) << 2);
    roff[tile] = (uint32_t)(row * 128 + pc * 16 + (col & 7) * 2);
  }

  const uint32_t ring_addr = lds_addr(ring);
  const int pre = nw < TNW_DEPTH ? nw : TNW_DEPTH;
  for (int i = 0; i < pre; ++i) issue(i);
  for (int i = 0; i < nw; ++i) {
    const int newer = (nw - 1 - i) < (TNW_DEPTH - 1) ? (nw - 1 - i) : (TNW_DEPTH - 1);
    tnw_wait_stages(newer);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t sa = ring_addr + (uint32_t)((i % TNW_DEPTH) * TNW_STAGE_BYTES);
    u32x2 ml[4], mh[4], sl[2], sh[2];   // M: [2 * column group + row tile]; S: [tile]; low / high k half
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint32_t ad = sa + (uint32_t)((a >> 1) * 2048) + roff[a & 1];
      DS_READ_TR(ml[a], ad, 0);
      DS_READ_TR(mh[a], ad, 512);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t ad = sa + 4096u + roff[c];
      DS_READ_TR(sl[c], ad, 0);
      DS_READ_TR(sh[c], ad, 512);
    }
    LGKM_WAIT0();
    bf16x8 bfr[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) bfr[c] = as_bf16x8(join2(sl[c], sh[c]));
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const bf16x8 af = as_bf16x8(join2(ml[a], mh[a]));
      acc[a][0] = mfma32(af, bfr[0], acc[a][0]);
      acc[a][1] = mfma32(af, bfr[1], acc[a][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (i + TNW_DEPTH < nw) issue(i + TNW_DEPTH);   // the slot's reads have returned: refill it
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // cross-wave sum of the eight [64][64] fp32 partials of each column group through LDS (aliases the rings)
  float* red = (float*)smem;  // [8 waves][64][64]
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (2 * dcg + half >= ncg) break;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          red[w * 4096 + (a * 32 + acc_row(reg, lane)) * 64 + c * 32 + (lane & 31)] = acc[2 * half + a][c][reg];
    __syncthreads();
    float* P = Pg + ((int64_t)slab * ncg * TN_BD + d0 + half * TN_BD) * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int v = t + 512 * it;  // 1024 float4 of the [64][64] tile
      f32x4 s0 = *(const f32x4*)(red + v * 4);
#pragma unroll
      for (int k = 1; k < TNW_WAVES; ++k) s0 += *(const f32x4*)(red + k * 4096 + v * 4);
      *(f32x4*)(P + v * 4) = s0;
    }
    __syncthreads();
  }
}

// Grouped, persistent launch (TnGroup): the block lists of up to TN_MAXG layers back to back, run by min(total, 256)
// resident workgroups (one per CU: 144 KiB of LDS) that loop over them -- workgroup launch, ring prologue and the drain of
// the partial stores are paid once per resident workgroup instead of once per block; weight gradients have no consumer
// before the optimizer, so a whole decoder block's layers (7 for llama) go into ONE launch.  Every block runs the
// single-layer code on its layer's parameters: bit-identical partials.
__global__ __launch_bounds__(64 * TNW_WAVES, 1) void tn_partial_dma_wide_kernel(const TnGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int total = grp.start[TN_MAXG];
  for (int blk = (int)blockIdx.x; blk < total; blk += (int)gridDim.x) {
    int layer = 0;
#pragma unroll
    for (int i = 1; i < TN_MAXG; ++i)
      if (i < grp.n && blk >= grp.start[i]) layer = i;
    int tt = t;
    asm volatile("" : "+v"(tt));   // keeps per-lane address arithmetic inside the iteration (see chain2.hip)
    tnw_block(grp.p[layer], blk - grp.start[layer], smem, tt, tt & 63, w);
  }
}

// =================================================================================================
// bf16 row-owner variant (grouped launches of whole decoder blocks): a workgroup owns ALL columns of its token slab.
//
// The column-owner kernels above re-read S (h / dh: 128 bytes per token) once per 128 columns of M -- on a 512-wide
// operand that is 4 x 4 MB from L2 beside 33.5 MB from HBM, and M arrives as 256-byte pieces of 1-KiB rows
// (tools/probe8.hip: 8.5 us for that pattern, 7.7 us when a workgroup reads whole rows and S once).  Here the eight waves
// of a workgroup split the COLUMNS of a range of up to 16 column groups (1024 columns) and walk the same tokens in lock
// step: every wave streams its own one or two [16 tok x 64 col] M images per k-step through a wave-private ring, the S
// images of a stage are DMA'd once (by waves 7, 6, ...) into a shared ring, one raw s_barrier per stage.  No cross-wave
// sum: a wave's accumulators are final for its columns and go straight to the partial buffer.
// A stage is 32 tokens (one group per wave: two k-steps) or 16 tokens (two groups per wave): 4 KiB of M per wave either
// way, 4 ring slots (3 stages = 100 KiB in flight per CU), 144 KiB of LDS, one workgroup per CU.
// With every column in one block the grid cannot be filled from a single layer (26 blocks for 512 -> 512); the slab
// counts are therefore planned over the whole grouped launch (tn_rows_plan: equal work per block, one resident round),
// which also cuts the fp32 partial traffic (13 slabs instead of 32 for a 512-wide operand of llama_60m).
// =================================================================================================
constexpr int TNR_WAVES = 8;
constexpr int TNR_DEPTH = 4;
constexpr int TNR_MSLOT = 4096;   // per wave and stage: two [16][64] bf16 images
constexpr int TNR_SSLOT = 4096;   // per stage: up to two [16][64] bf16 images of S
constexpr int TNR_LDS = TNR_WAVES * TNR_DEPTH * TNR_MSLOT + TNR_DEPTH * TNR_SSLOT;   // 144 KiB

__device__ __forceinline__ void tnr_wait(int n) {   // all but the n most recent DMA instructions of this wave have landed
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;   // 2 stages x 5 instructions
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int CGW>
__device__ __forceinline__ void tnr_block(const TnRowsItem& J, const int b, char* smem, const int lane, const int w, const bool nt) {
  constexpr int KS = 2 / CGW;      // 16-token k-steps per stage
  constexpr int TOKS = 16 * KS;
  const int range = b % J.nr, slab = b / J.nr;
  const int g_lo = range * J.gpr + w * CGW;
  int g_hi = (range + 1) * J.gpr;
  if (g_hi > J.ncg) g_hi = J.ncg;
  const int nmy = g_lo >= g_hi ? 0 : (g_hi - g_lo < CGW ? g_hi - g_lo : CGW);   // column groups of this wave
  const int64_t t_begin = (int64_t)slab * J.slab_len;
  int64_t t_end = t_begin + J.slab_len;
  if (t_end > J.T) t_end = J.T;
  const int nstage = t_begin < t_end ? (int)((t_end - t_begin + TOKS - 1) / TOKS) : 0;
  const int sq = TNR_WAVES - 1 - w;        // S instruction of this wave (image sq >> 1, row half sq & 1)
  const bool s_own = sq < 2 * KS;
  const int ni = 2 * (CGW == 2 ? nmy : 2 * nmy) + (s_own ? 1 : 0);   // DMA instructions of this wave per stage

  const bf16_t* Mg = (const bf16_t*)J.M;
  const bf16_t* Sg = (const bf16_t*)J.S;
  char* mring = smem + w * (TNR_DEPTH * TNR_MSLOT);
  char* sring = smem + TNR_WAVES * TNR_DEPTH * TNR_MSLOT;
  const char* zp = zero_page_for(lane);
  const int drow = lane >> 3, dpc = lane & 7;

  // running per-lane source pointers: instruction (image im, row half hf) of every stage
  const char* mp[2][2];
  int64_t mstep[2][2];
  int mtok[2][2];
#pragma unroll
  for (int im = 0; im < 2; ++im)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int row = hf * 8 + drow;
      const int lc = dpc ^ (((row >> 1) & 1) << 2);
      const int grp = CGW == 2 ? g_lo + im : g_lo;
      const int tok = (CGW == 2 ? 0 : im * 16) + row;
      const bool ok = (CGW == 2 ? im < nmy : nmy > 0) && grp * TN_BD + lc * 8 < J.D;
      mp[im][hf] = ok ? (const char*)(Mg + (t_begin + tok) * J.ldm + grp * TN_BD + lc * 8) : zp;
      mstep[im][hf] = ok ? (int64_t)TOKS * J.ldm * 2 : 0;
      mtok[im][hf] = tok;
    }
  const char* sp = zp;
  int64_t sstep = 0;
  int stok = 0;
  if (s_own) {
    const int row = (sq & 1) * 8 + drow;
    const int lc = dpc ^ (((row >> 1) & 1) << 2);
    stok = (sq >> 1) * 16 + row;
    sp = (const char*)(Sg + (t_begin + stok) * 64 + lc * 8);
    sstep = (int64_t)TOKS * 64 * 2;
  }
  auto issue = [&](int i) {   // strictly in order: the pointers are at stage i
    const int64_t tt0 = t_begin + (int64_t)i * TOKS;
    const bool whole = tt0 + TOKS <= t_end;   // wave-uniform
    char* ms = mring + (i % TNR_DEPTH) * TNR_MSLOT;
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      if (CGW == 2 ? im < nmy : nmy > 0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const void* src = mp[im][hf];
          if (!whole && tt0 + mtok[im][hf] >= t_end) src = zp;
          if (nt) dma16_nt(src, ms + (im * 2 + hf) * 1024);
          else dma16(src, ms + (im * 2 + hf) * 1024);
          mp[im][hf] += mstep[im][hf];
        }
      }
    }
    if (s_own) {
      const void* src = sp;
      if (!whole && tt0 + stok >= t_end) src = zp;
      dma16(src, sring + (i % TNR_DEPTH) * TNR_SSLOT + sq * 1024);
      sp += sstep;
    }
  };

  f32x16 acc[CGW][2][2];   // [group][32-row tile of the group][S tile]
#pragma unroll
  for (int gi = 0; gi < CGW; ++gi)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[gi][a][c][i] = 0.f;

  const int g = lane >> 4, jj = lane & 15, q4 = jj >> 2, pp = jj & 3, h = g >> 1;
  uint32_t roff[2];   // transposed read of rows 8h + q (+512: rows 8h + 4 + q) of a [16][64] image, 32-column tile
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    const int row = 8 * h + q4;
    const int col = tile * 32 + 16 * (g & 1) + 4 * pp;
    const int pc = (col >> 3) ^ (((row >> 1) & 1) << 2);
    roff[tile] = (uint32_t)(row * 128 + pc * 16 + (col & 7) * 2);
  }
  const uint32_t mring_a = lds_addr(mring), sring_a = lds_addr(sring);

  const int pre = nstage < TNR_DEPTH - 1 ? nstage : TNR_DEPTH - 1;
  for (int i = 0; i < pre; ++i) issue(i);
#pragma unroll 1
  for (int i = 0; i < nstage; ++i) {
    const int newer = (nstage - 1 - i) < (TNR_DEPTH - 2) ? (nstage - 1 - i) : (TNR_DEPTH - 2);
    tnr_wait(newer * ni);    // this wave's part of stage i has landed ...
    raw_barrier();           // ... and so has everybody else's (S); all waves have finished stage i - 1
    if (i + TNR_DEPTH - 1 < nstage) issue(i + TNR_DEPTH - 1);   // into the slots of stage i - 1
    if (nmy == 0) continue;
    const uint32_t ma = mring_a + (uint32_t)((i % TNR_DEPTH) * TNR_MSLOT);
    const uint32_t sa = sring_a + (uint32_t)((i % TNR_DEPTH) * TNR_SSLOT);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x2 sl[2], sh[2], ml[CGW][2], mh[CGW][2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t ad = sa + (uint32_t)(ks * 2048) + roff[c];
        DS_READ_TR(sl[c], ad, 0);
        DS_READ_TR(sh[c], ad, 512);
      }
#pragma unroll
      for (int gi = 0; gi < CGW; ++gi)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const uint32_t ad = ma + (uint32_t)((CGW == 2 ? gi : ks) * 2048) + roff[a];
          DS_READ_TR(ml[gi][a], ad, 0);
          DS_READ_TR(mh[gi][a], ad, 512);
        }
      LGKM_WAIT0();
      bf16x8 bfr[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) bfr[c] = as_bf16x8(join2(sl[c], sh[c]));
#pragma unroll
      for (int gi = 0; gi < CGW; ++gi) {
        if (gi < nmy) {   // wave-uniform; an image that is not this wave's holds stale data
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const bf16x8 af = as_bf16x8(join2(ml[gi][a], mh[gi][a]));
            acc[gi][a][0] = mfma32(af, bfr[0], acc[gi][a][0]);
            acc[gi][a][1] = mfma32(af, bfr[1], acc[gi][a][1]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // partial[slab][group * 64 + row][col]: accumulator (a, c) register reg of lane l = row a*32 + acc_row(reg, l), col c*32 + (l & 31)
#pragma unroll
  for (int gi = 0; gi < CGW; ++gi) {
    if (gi < nmy) {
      float* P = J.partial + ((int64_t)slab * J.ncg * TN_BD + (int64_t)(g_lo + gi) * TN_BD) * 64;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            P[(a * 32 + acc_row(reg, lane)) * 64 + c * 32 + (lane & 31)] = acc[gi][a][c][reg];
    }
  }
  raw_barrier();   // end of block: every wave is done with the shared S ring before the next block's DMA refills it
}

__global__ __launch_bounds__(64 * TNR_WAVES, 1) void tn_partial_rows_kernel(const TnRowsGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  for (int blk = (int)blockIdx.x; blk < grp.total; blk += (int)gridDim.x) {
    int item = 0;
#pragma unroll
    for (int i = 1; i < TNR_MAXI; ++i)
      if (i < grp.n && blk >= grp.it[i].start) item = i;
    int tt = (int)threadIdx.x;
    asm volatile("" : "+v"(tt));   // keeps per-lane address arithmetic inside the iteration (see chain2.hip)
    if (grp.it[item].cgw == 2) tnr_block<2>(grp.it[item], blk - grp.it[item].start, smem, tt & 63, w, grp.nt_load != 0);
    else tnr_block<1>(grp.it[item], blk - grp.it[item].start, smem, tt & 63, w, grp.nt_load != 0);
  }
}

// =================================================================================================
// fp32 fast path: the same wave-private LDS-DMA rings with exact-fp32 MFMA (32x32x2).
//
// fp32 needs no transposed reads: the MFMA takes ONE token per lane-half, and with the tiles in their
// natural [token][column] layout lane li reads its column of token 2kp + lh with a plain row-wise read.
// To fetch both 32-row tiles of an operand in one instruction, lane li owns columns (2 li, 2 li + 1):
// tile a of the M operand is "columns == a (mod 2)", so one ds_read_b64 along the row (32 lanes = 256
// contiguous bytes, conflict-free) feeds two MFMAs, and accumulator (a, c) register `reg` of lane
// (li, lh) is G[2 * acc_row + a][2 * li + c].  Groups are 8 tokens (4 KiB: 2 + 2 DMA instructions), 4 in
// flight per wave, 64 KiB of LDS, two workgroups per CU.  Per group: 8 ds_read_b64, 16 MFMAs (1024
// cycles): the kernel is MFMA-bound at ~the HBM rate (AI = 32 flop/B).
// =================================================================================================
constexpr int TNF_DEPTH = 4;
constexpr int TNF_STAGE_BYTES = 4096;       // [8][64] fp32 M tile + [8][64] fp32 S tile

// X3 = true: the 3 x bf16 form above -- a k-step is 16 tokens = TWO consecutive 8-token stages of this wave (lane-half lh
// takes stage 2 q + lh; any 16 tokens will do, M and S use the same ones), accumulator layout unchanged.
// X3 = false: the exact-fp32 MFMA form (kept behind the F32_EXACT switch: the reference for the A/B test).
template <bool X3> __global__ __launch_bounds__(256, 2) void tn_partial_dma_f32_kernel(const TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int b = blockIdx.x, jid = 0;
  if (p.njobs > 1 && b >= p.job[0].ncg * p.ns) {
    b -= p.job[0].ncg * p.ns;
    jid = 1;
  }
  const float* Mg = (const float*)(jid ? p.job[1].M : p.job[0].M);
  const float* Sg = (const float*)(jid ? p.job[1].S : p.job[0].S);
  float* Pg = jid ? p.job[1].partial : p.job[0].partial;
  const int64_t ldm = jid ? p.job[1].ldm : p.job[0].ldm;
  const int D = jid ? p.job[1].D : p.job[0].D;
  const int ncg = jid ? p.job[1].ncg : p.job[0].ncg;
  const int cg = b / p.ns, slab = b % p.ns;
  const int d0 = cg * TN_BD;
  const int64_t t_begin = (int64_t)slab * p.slab_len;
  int64_t t_end = t_begin + p.slab_len;
  if (t_end > p.T) t_end = p.T;
  char* ring = smem + w * (TNF_DEPTH * TNF_STAGE_BYTES);

  const int ngroups = t_begin < t_end ? (int)((t_end - t_begin + 7) / 8) : 0;
  const int nw = ngroups > w ? (ngroups - w + 3) / 4 : 0;  // groups of this wave (group g -> wave g % 4)

  // DMA sources: per-lane running pointers advanced by a per-lane constant in issue order (2 VALU per DMA) -- on
  // this path every non-MFMA instruction is paid for in matrix-pipe time.  Only a group that
  // crosses t_end takes the checked path.
  const int drow = lane >> 4, dpc = lane & 15;   // DMA lane -> (row in a 4-row block, 16-byte chunk)
  const char* zp = zero_page_for(lane);
  const bool mcol_ok = d0 + dpc * 4 < D;
  const int64_t m_step = mcol_ok ? (int64_t)32 * ldm * 4 : 0;   // 4 waves x 8 tokens per step of this wave
  const int64_t s_step = (int64_t)32 * 64 * 4;
  const char* pm[2];
  const char* ps[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int64_t tt = t_begin + (int64_t)w * 8 + hf * 4 + drow;
    pm[hf] = mcol_ok ? (const char*)(Mg + tt * ldm + d0 + dpc * 4) : zp;
    ps[hf] = (const char*)(Sg + tt * 64 + dpc * 4);
  }
  auto issue = [&](int i) {   // groups are issued strictly in order: pm / ps point at group i
    const int64_t tt0 = t_begin + (int64_t)(w + 4 * i) * 8;
    char* slot = ring + (i % TNF_DEPTH) * TNF_STAGE_BYTES;
    const bool whole = tt0 + 8 <= t_end;   // wave-uniform
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const void* srcM = pm[half];
      const void* srcS = ps[half];
      if (!whole && tt0 + half * 4 + drow >= t_end) srcM = zp, srcS = zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcM,
                                       (__attribute__((address_space(3))) void*)(slot + half * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcS,
                                       (__attribute__((address_space(3))) void*)(slot + 2048 + half * 1024), 16, 0, 0);
      pm[half] += m_step;
      ps[half] += s_step;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  const uint32_t lane_off = (uint32_t)(lh * 256 + li * 8);   // token row lh of a pair, columns 2 li, 2 li + 1
  const int pre = nw < TNF_DEPTH ? nw : TNF_DEPTH;
  for (int i = 0; i < pre; ++i) issue(i);
  if constexpr (X3) {
    const int npair = (nw + 1) / 2;
    for (int q = 0; q < npair; ++q) {
      const bool has1 = 2 * q + 1 < nw;   // wave-uniform: the second stage of the pair exists
      const int last = has1 ? 2 * q + 1 : 2 * q;
      const int issued = nw < 2 * q + TNF_DEPTH ? nw : 2 * q + TNF_DEPTH;
      tn_wait_stages(issued - 1 - last);
      __builtin_amdgcn_sched_barrier(0);
      const int st = 2 * q + (has1 ? lh : 0);
      const uint32_t ad = ring_addr + (uint32_t)((st % TNF_DEPTH) * TNF_STAGE_BYTES) + (uint32_t)(li * 8);
      f32x2 mv[8], sv[8];
      DS_READ_B64(mv[0], ad, 0 * 256);
      DS_READ_B64(mv[1], ad, 1 * 256);
      DS_READ_B64(mv[2], ad, 2 * 256);
      DS_READ_B64(mv[3], ad, 3 * 256);
      DS_READ_B64(mv[4], ad, 4 * 256);
      DS_READ_B64(mv[5], ad, 5 * 256);
      DS_READ_B64(mv[6], ad, 6 * 256);
      DS_READ_B64(mv[7], ad, 7 * 256);
      DS_READ_B64(sv[0], ad, 2048 + 0 * 256);
      DS_READ_B64(sv[1], ad, 2048 + 1 * 256);
      DS_READ_B64(sv[2], ad, 2048 + 2 * 256);
      DS_READ_B64(sv[3], ad, 2048 + 3 * 256);
      DS_READ_B64(sv[4], ad, 2048 + 4 * 256);
      DS_READ_B64(sv[5], ad, 2048 + 5 * 256);
      DS_READ_B64(sv[6], ad, 2048 + 6 * 256);
      DS_READ_B64(sv[7], ad, 2048 + 7 * 256);
      LGKM_WAIT0();
      const bool dead = !has1 && lh == 1;   // odd tail: the upper lane-half has no tokens
      u32x4 mp[2][3], sp[2][3];            // [column parity][plane]
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const float m0 = dead ? 0.f : mv[2 * pr][a], m1 = dead ? 0.f : mv[2 * pr + 1][a];
          const float s0 = dead ? 0.f : sv[2 * pr][a], s1 = dead ? 0.f : sv[2 * pr + 1][a];
          uint32_t h, m, l;
          split3(m0, m1, h, m, l);
          mp[a][0][pr] = h, mp[a][1][pr] = m, mp[a][2][pr] = l;
          split3(s0, s1, h, m, l);
          sp[a][0][pr] = h, sp[a][1][pr] = m, sp[a][2][pr] = l;
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = mfma_x3(mp[a], sp[c], acc[a][c]);
      __builtin_amdgcn_sched_barrier(0);
      if (2 * q + TNF_DEPTH < nw) issue(2 * q + TNF_DEPTH);
      if (2 * q + 1 + TNF_DEPTH < nw) issue(2 * q + 1 + TNF_DEPTH);
    }
  } else {
  for (int i = 0; i < nw; ++i) {
    const int newer = (nw - 1 - i) < (TNF_DEPTH - 1) ? (nw - 1 - i) : (TNF_DEPTH - 1);
    tn_wait_stages(newer);
    const uint32_t ad = ring_addr + (uint32_t)((i % TNF_DEPTH) * TNF_STAGE_BYTES) + lane_off;
    // NB: float-typed vectors on purpose -- with uint32 vectors + bit_cast hipcc (ROCm 7.2) folds element 1
    // of the pair into element 0 when the value feeds an MFMA intrinsic
    f32x2 m0, m1, m2, m3, s0, s1, s2, s3;
    DS_READ_B64(m0, ad, 0);
    DS_READ_B64(s0, ad, 2048);
    DS_READ_B64(m1, ad, 512);
    DS_READ_B64(s1, ad, 2560);
    DS_READ_B64(m2, ad, 1024);
    DS_READ_B64(s2, ad, 3072);
    DS_READ_B64(m3, ad, 1536);
    DS_READ_B64(s3, ad, 3584);
    LGKM_WAIT0();
#define TNF_PAIR(MV, SV)                                                                       \
  {                                                                                            \
    const float ma0 = MV[0], ma1 = MV[1];                                                      \
    const float sc0 = SV[0], sc1 = SV[1];                                                      \
    acc[0][0] = mfma32(ma0, sc0, acc[0][0]);                                                   \
    acc[0][1] = mfma32(ma0, sc1, acc[0][1]);                                                   \
    acc[1][0] = mfma32(ma1, sc0, acc[1][0]);                                                   \
    acc[1][1] = mfma32(ma1, sc1, acc[1][1]);                                                   \
  }
    TNF_PAIR(m0, s0)
    TNF_PAIR(m1, s1)
    TNF_PAIR(m2, s2)
    TNF_PAIR(m3, s3)
#undef TNF_PAIR
    __builtin_amdgcn_sched_barrier(0);
    if (i + TNF_DEPTH < nw) issue(i + TNF_DEPTH);
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // cross-wave sum through LDS (aliases the rings), un-interleaving rows and columns on the way
  float* red = (float*)smem;  // [4 waves][64][64]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        red[w * 4096 + (2 * acc_row(reg, lane) + a) * 64 + 2 * li + c] = acc[a][c][reg];
  __syncthreads();
  float* P = Pg + ((int64_t)slab * ncg * TN_BD + d0) * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = t + 256 * it;
    f32x4 q0 = *(const f32x4*)(red + v * 4);
    const f32x4 q1 = *(const f32x4*)(red + 4096 + v * 4);
    const f32x4 q2 = *(const f32x4*)(red + 8192 + v * 4);
    const f32x4 q3 = *(const f32x4*)(red + 12288 + v * 4);
    q0 = (q0 + q1) + (q2 + q3);
    *(f32x4*)(P + v * 4) = q0;
  }
}

// =================================================================================================
// fp32 wide variant (3 x bf16 products): TWO 64-column groups of M per workgroup, as the bf16 wide kernel -- with the
// matrix pipe out of the way (3 x bf16) the narrow fp32 kernel is bound by the DMA path as a whole, and half of what it
// moves is the S stream (h / dh rows from L2, once per column group).  6 waves, wave-private rings of 4 stages of
// [8 tok] x ([128] M + [64] S) fp32 = 6 KiB (6 DMA instructions of 4 rows x 256 B), 144 KiB of LDS, one workgroup per CU.
// A k-step is 16 tokens = two consecutive stages of the wave (lane-half lh takes stage 2 q + lh).  Lane li owns columns
// (2 li, 2 li + 1) of every 64-column image, so accumulator (g, a, c) register `reg` is G_g[2 * acc_row + a][2 * li + c],
// the layout of the narrow kernel; the column groups are processed one after the other to keep the registers under 256.
// =================================================================================================
constexpr int TNFW_DEPTH = 4;
constexpr int TNFW_STAGE_BYTES = 6144;
constexpr int TNFW_WAVES = 6;

__global__ __launch_bounds__(64 * TNFW_WAVES, 1) void tn_partial_dma_f32_wide_kernel(const TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int b = blockIdx.x, jid = 0;
  const int ncg2_0 = (p.job[0].ncg + 1) / 2;
  if (p.njobs > 1 && b >= ncg2_0 * p.ns) {
    b -= ncg2_0 * p.ns;
    jid = 1;
  }
  const float* Mg = (const float*)(jid ? p.job[1].M : p.job[0].M);
  const float* Sg = (const float*)(jid ? p.job[1].S : p.job[0].S);
  float* Pg = jid ? p.job[1].partial : p.job[0].partial;
  const int64_t ldm = jid ? p.job[1].ldm : p.job[0].ldm;
  const int D = jid ? p.job[1].D : p.job[0].D;
  const int ncg = jid ? p.job[1].ncg : p.job[0].ncg;
  const int dcg = b / p.ns, slab = b % p.ns;
  const int d0 = dcg * 2 * TN_BD;
  const int64_t t_begin = (int64_t)slab * p.slab_len;
  int64_t t_end = t_begin + p.slab_len;
  if (t_end > p.T) t_end = p.T;
  char* ring = smem + w * (TNFW_DEPTH * TNFW_STAGE_BYTES);
  const int ngroups = t_begin < t_end ? (int)((t_end - t_begin + 7) / 8) : 0;
  const int nw = ngroups > w ? (ngroups - w + TNFW_WAVES - 1) / TNFW_WAVES : 0;   // 8-token groups of this wave

  // DMA instruction q = 2 * tile + half: tile 0, 1 = the two M column groups, 2 = S; half = token rows 4 half .. 4 half + 3
  const int drow = lane >> 4, dpc = lane & 15;
  const char* zp = zero_page_for(lane);
  const char* ptr[6];
  int64_t step[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int tile = q >> 1, half = q & 1;
    const int64_t tt = t_begin + (int64_t)w * 8 + half * 4 + drow;
    if (tile < 2) {
      const bool ok = d0 + tile * TN_BD + dpc * 4 < D;
      ptr[q] = ok ? (const char*)(Mg + tt * ldm + d0 + tile * TN_BD + dpc * 4) : zp;
      step[q] = ok ? (int64_t)TNFW_WAVES * 8 * ldm * 4 : 0;
    } else {
      ptr[q] = (const char*)(Sg + tt * 64 + dpc * 4);
      step[q] = (int64_t)TNFW_WAVES * 8 * 64 * 4;
    }
  }
  auto issue = [&](int i) {   // strictly in order: ptr[] points at group i
    const int64_t tt0 = t_begin + (int64_t)(w + TNFW_WAVES * i) * 8;
    char* slot = ring + (i % TNFW_DEPTH) * TNFW_STAGE_BYTES;
    const bool whole = tt0 + 8 <= t_end;   // wave-uniform
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const void* src = ptr[q];
      if (!whole && tt0 + (q & 1) * 4 + drow >= t_end) src = zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
      ptr[q] += step[q];
    }
  };

  f32x16 acc[2][2][2];   // [column group][row parity a][S column parity c]
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][a][c][i] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const uint32_t ring_addr = lds_addr(ring);
  const int pre = nw < TNFW_DEPTH ? nw : TNFW_DEPTH;
  for (int i = 0; i < pre; ++i) issue(i);
  const int npair = (nw + 1) / 2;
  for (int q = 0; q < npair; ++q) {
    const bool has1 = 2 * q + 1 < nw;
    const int last = has1 ? 2 * q + 1 : 2 * q;
    const int issued = nw < 2 * q + TNFW_DEPTH ? nw : 2 * q + TNFW_DEPTH;
    tnw_wait_stages(issued - 1 - last);   // 6 DMA instructions per stage, at most two stages newer
    __builtin_amdgcn_sched_barrier(0);
    const int st = 2 * q + (has1 ? lh : 0);
    const uint32_t ad = ring_addr + (uint32_t)((st % TNFW_DEPTH) * TNFW_STAGE_BYTES) + (uint32_t)(li * 8);
    const bool dead = !has1 && lh == 1;
    f32x2 sv[8];
    DS_READ_B64(sv[0], ad, 4096 + 0 * 256);
    DS_READ_B64(sv[1], ad, 4096 + 1 * 256);
    DS_READ_B64(sv[2], ad, 4096 + 2 * 256);
    DS_READ_B64(sv[3], ad, 4096 + 3 * 256);
    DS_READ_B64(sv[4], ad, 4096 + 4 * 256);
    DS_READ_B64(sv[5], ad, 4096 + 5 * 256);
    DS_READ_B64(sv[6], ad, 4096 + 6 * 256);
    DS_READ_B64(sv[7], ad, 4096 + 7 * 256);
    f32x2 mv[8];
    DS_READ_B64(mv[0], ad, 0 * 256);
    DS_READ_B64(mv[1], ad, 1 * 256);
    DS_READ_B64(mv[2], ad, 2 * 256);
    DS_READ_B64(mv[3], ad, 3 * 256);
    DS_READ_B64(mv[4], ad, 4 * 256);
    DS_READ_B64(mv[5], ad, 5 * 256);
    DS_READ_B64(mv[6], ad, 6 * 256);
    DS_READ_B64(mv[7], ad, 7 * 256);
    LGKM_WAIT0();
    u32x4 sp[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) split3v(dead ? 0.f : sv[2 * pr][c], dead ? 0.f : sv[2 * pr + 1][c], sp[c], pr);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      u32x4 mp[2][3];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) split3v(dead ? 0.f : mv[2 * pr][a], dead ? 0.f : mv[2 * pr + 1][a], mp[a], pr);
      if (g == 0) {   // fetch the second column group while the first multiplies
        DS_READ_B64(mv[0], ad, 2048 + 0 * 256);
        DS_READ_B64(mv[1], ad, 2048 + 1 * 256);
        DS_READ_B64(mv[2], ad, 2048 + 2 * 256);
        DS_READ_B64(mv[3], ad, 2048 + 3 * 256);
        DS_READ_B64(mv[4], ad, 2048 + 4 * 256);
        DS_READ_B64(mv[5], ad, 2048 + 5 * 256);
        DS_READ_B64(mv[6], ad, 2048 + 6 * 256);
        DS_READ_B64(mv[7], ad, 2048 + 7 * 256);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[g][a][c] = mfma_x3(mp[a], sp[c], acc[g][a][c]);
      if (g == 0) LGKM_WAIT0();
    }
    __builtin_amdgcn_sched_barrier(0);
    if (2 * q + TNFW_DEPTH < nw) issue(2 * q + TNFW_DEPTH);
    if (2 * q + 1 + TNFW_DEPTH < nw) issue(2 * q + 1 + TNFW_DEPTH);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // cross-wave sum of the six [64][64] fp32 partials of each column group through LDS (aliases the rings)
  float* red = (float*)smem;  // [6 waves][64][64]
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (2 * dcg + g >= ncg) break;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          red[w * 4096 + (2 * acc_row(reg, lane) + a) * 64 + 2 * li + c] = acc[g][a][c][reg];
    __syncthreads();
    float* P = Pg + ((int64_t)slab * ncg * TN_BD + d0 + g * TN_BD) * 64;
    for (int v = t; v < 1024; v += 64 * TNFW_WAVES) {   // 1024 float4 of the [64][64] tile
      f32x4 s0 = *(const f32x4*)(red + v * 4);
#pragma unroll
      for (int k = 1; k < TNFW_WAVES; ++k) s0 += *(const f32x4*)(red + k * 4096 + v * 4);
      *(f32x4*)(P + v * 4) = s0;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Deterministic slab reduction + crop + optional transpose + cast.
//   out[d][r]   (transpose = 0, ld = out_ld)   or   out[r][d]   (transpose = 1)
//   colsum[d] = sum_s partial[s][d][ones_col]
// ---------------------------------------------------------------------------------------------

template <typename T> __device__ __forceinline__ void tn_reduce_body(const ReduceParams& p, int b) {
  int jid = 0;
  if (p.njobs > 1 && b >= p.blocks0) {
    b -= p.blocks0;
    jid = 1;
  }
  const ReduceJob& J = p.job[jid];
  const int ns = J.ns > 0 ? J.ns : p.ns;
  // 256 threads = 4 slab splits x 4 rows x 16 column quads; splits are summed through LDS in a fixed
  // order, so the result does not depend on scheduling
  __shared__ f32x4 red[4][64];
  const int split = threadIdx.x >> 6, rl = (threadIdx.x >> 4) & 3, cq = threadIdx.x & 15;
  const int d = b * 4 + rl;
  const int c4 = cq * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (d < J.D) {
    const int64_t stride = (int64_t)J.Dpad * 64;
    const float* src = J.partial + (int64_t)d * 64 + c4;
    // four loads in flight per thread (the kernel is a chain of dependent round trips otherwise); the summation
    // order per thread stays i = split, split + 4, ... so the result is unchanged
    int i = split;
    for (; i + 12 < ns; i += 16) {
      const f32x4 v0 = *(const f32x4*)(src + (int64_t)i * stride);
      const f32x4 v1 = *(const f32x4*)(src + (int64_t)(i + 4) * stride);
      const f32x4 v2 = *(const f32x4*)(src + (int64_t)(i + 8) * stride);
      const f32x4 v3 = *(const f32x4*)(src + (int64_t)(i + 12) * stride);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = (((s[e] + v0[e]) + v1[e]) + v2[e]) + v3[e];
    }
    for (; i < ns; i += 4) {
      const f32x4 v = *(const f32x4*)(src + (int64_t)i * stride);
      s[0] += v[0], s[1] += v[1], s[2] += v[2], s[3] += v[3];
    }
  }
  red[split][threadIdx.x & 63] = s;
  __syncthreads();
  if (split != 0 || d >= J.D) return;
  {
    const f32x4 s1 = red[1][threadIdx.x], s2 = red[2][threadIdx.x], s3 = red[3][threadIdx.x];
    s = (s + s1) + (s2 + s3);
  }
  T* out = (T*)J.out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c4 + j;
    if (c < J.r) {
      T* dst = J.transpose ? out + (int64_t)c * J.out_ld + d : out + (int64_t)d * J.out_ld + c;
      float v = J.alpha * s[j];
      if (J.beta != 0.f) v += J.beta * to_f32(*dst);
      *dst = from_f32<T>(v);
    }
    if (J.colsum && c == J.ones_col) {
      T* cs = (T*)J.colsum + d;
      float v = s[j];
      if (J.beta != 0.f) v += J.beta * to_f32(*cs);
      *cs = from_f32<T>(v);
    }
  }
}

template <typename T> __global__ __launch_bounds__(256) void tn_reduce_kernel(const ReduceParams p) {
  tn_reduce_body<T>(p, (int)blockIdx.x);
}

// The reductions of many layers in one launch (deferred weight-gradient reduction): block -> layer by binary search in
// the prefix sums of the per-layer block counts; same per-element summation order as the per-layer kernel.
template <typename T>
__global__ __launch_bounds__(256) void tn_reduce_batch_kernel(const ReduceParams* __restrict__ descs, const int* __restrict__ starts,
                                                              int n) {
  int lo = 0, hi = n - 1;
  const int b = (int)blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= b) lo = mid;
    else hi = mid - 1;
  }
  tn_reduce_body<T>(descs[lo], b - starts[lo]);
}

// ---------------------------------------------------------------------------------------------
int tn_pick_slabs(int64_t T, int total_colgroups, int total_colgroup_pairs, int total_colgroup_quads, int dtype, int* slab_len) {
  const int bt = dtype == SOW_F32 ? TnCfg<float>::BT : TnCfg<bf16_t>::BT;
  // Two workgroups per CU are resident (64 KiB of LDS each), 512 in all; slabs stay >= 512 tokens so the
  // partial traffic (ns * D * 64 * 4 bytes) is a small fraction of the streamed operand.
  //   bf16 (HBM-bound): see below; a multiple of 8 keeps the column groups of one slab on one XCD (S from that L2).
  //   fp32 (MFMA-bound): AT MOST 512 blocks -- a 13 % overshoot is a second, nearly empty round on a saturated
  //     matrix pipe (measured: 576 blocks 93 us vs 504 blocks 74 us).
  int ns;
  int64_t max_ns = (T + 511) / 512;
  if (max_ns < 1) max_ns = 1;
  if (dtype == SOW_F32 && (sw_on(SW_F32_EXACT) || sw_on(SW_TN_NARROW))) {
    ns = 512 / total_colgroups;
    if (ns > max_ns) ns = (int)max_ns;
    if (ns < 1) ns = 1;
    if (ns > 8 && (ns & ~7) * 10 >= ns * 9) ns &= ~7;
  } else {   // bf16, and fp32 on the wide 3 x bf16 kernel
    // the wide kernel owns two column groups per block and one 8-wave workgroup per CU: the largest multiple of 8
    // that keeps the grid within ONE round of 256 (measured at d = 768, 12 double groups: 16 slabs / 192 blocks
    // 22.2 us, 21 / 252 26.9 us, 24 / 288 30.5 us; llama_60m: 32 slabs for 512/512, 16 for the 1376-wide layers)
    const int cg2 = total_colgroup_pairs > 0 ? total_colgroup_pairs : 1;
    ns = (256 / cg2) & ~7;
    if (ns < 8) ns = 256 / cg2;
    // fp32 wide kernel: latency-bound on the bytes one CU keeps in flight (144 KiB of rings), so every CU counts:
    // no rounding to a multiple of 8 (d = 768: 21 slabs / 252 blocks instead of 16 / 192)
    if (dtype == SOW_F32) ns = 256 / cg2;
    // fp32 quad kernel (skinny_tn_f32q.hip): four column groups per block, one 8-wave workgroup per CU, one round
    // (slab counts are padded to a multiple of 8 per operand there -- same-slab ranges share an XCD -- so a multiple of 8 it is,
    // where that still leaves at least 8 slabs: the padded grid must stay within the one round)
    if (dtype == SOW_F32 && total_colgroup_quads > 0) {
      ns = 256 / total_colgroup_quads;
      if (ns >= 8) ns &= ~7;
    }
    if (ns > max_ns) ns = (int)max_ns;
    if (ns < 1) ns = 1;
  }
  int64_t len = (T + ns - 1) / ns;
  len = (len + bt - 1) / bt * bt;
  if (len < bt) len = bt;
  ns = (int)((T + len - 1) / len);
  if (ns < 1) ns = 1;
  *slab_len = (int)len;
  return ns;
}

size_t tn_partial_bytes(int ns, int D) { return (size_t)ns * ((D + 63) / 64 * 64) * 64 * sizeof(float); }

static bool tn_wide_ok(const TnParams& p) {
  bool dma = p.slab_len % 16 == 0;
  for (int j = 0; j < p.njobs; ++j) {
    const TnJob& J = p.job[j];
    dma = dma && J.D % 8 == 0 && J.ldm % 8 == 0 && (reinterpret_cast<uintptr_t>(J.M) & 15) == 0 &&
          (reinterpret_cast<uintptr_t>(J.S) & 15) == 0 && J.ones_col_in_s;
  }
  return dma;
}
static int tn_wide_blocks(const TnParams& p) {
  int blocks2 = 0;   // two 64-column groups per block
  for (int j = 0; j < p.njobs; ++j) blocks2 += (p.job[j].ncg + 1) / 2 * p.ns;
  return blocks2;
}

bool tn_group_supported(const TnParams& p, int dtype) {
  return dtype == SOW_BF16 && tn_wide_ok(p) && !sw_on(SW_TN_NARROW) && tn_wide_blocks(p) > 0;
}

int launch_tn_group(const TnParams* ps, int n, hipStream_t stream) {
  if (n <= 0) return SOW_OK;
  if (n > TN_MAXG) return SOW_ERR_SHAPE;
  TnGroup g{};
  g.n = n;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = ps[i];
    g.start[i] = (int)total;
    total += tn_wide_blocks(ps[i]);
  }
  for (int i = n; i <= TN_MAXG; ++i) g.start[i] = (int)total;
  if (total <= 0) return SOW_OK;
  if (total > 0x7fffffff) return SOW_ERR_SHAPE;
  constexpr int LDS = TNW_WAVES * TNW_DEPTH * TNW_STAGE_BYTES;  // 144 KiB (rings; reused by the cross-wave sum)
  SOW_SET_MAX_LDS_ONCE(LDS, tn_partial_dma_wide_kernel);
  const int64_t grid = (sw_on(SW_NO_PERSIST) || total < 256) ? total : 256;   // one resident workgroup per CU
  hipLaunchKernelGGL(tn_partial_dma_wide_kernel, dim3((unsigned)grid), dim3(64 * TNW_WAVES), LDS, stream, g);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

bool tn_rows_plan(const int64_t* T, const int* D, const int* cap, int n, int* ns_out, int* slab_len_out) {
  if (n <= 0 || n > TNR_MAXI || sw_on(SW_NO_TN_ROWS) || sw_on(SW_TN_NARROW) || sw_on(SW_NO_GROUPED)) return false;
  double work = 0.0;
  for (int i = 0; i < n; ++i) {
    if (T[i] <= 0 || D[i] <= 0 || D[i] % 8) return false;
    work += (double)T[i] * D[i];
  }
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const int ncg = (D[i] + 63) / 64, nr = (ncg + 15) / 16;
    int ns = (int)(256.0 * (double)T[i] * D[i] / work) / nr;   // this item's share of one resident round
    const int64_t max_ns = T[i] / 512 > 0 ? T[i] / 512 : 1;   // slabs of at least 512 tokens
    if (ns > max_ns) ns = (int)max_ns;
    if (ns < 1) ns = 1;
    if (ns > TNR_MAX_SLABS) return false;                      // a small group: the column-owner kernel fills the chip better
    int64_t len = (T[i] + ns - 1) / ns;
    len = (len + 31) / 32 * 32;
    ns = (int)((T[i] + len - 1) / len);
    if (ns > cap[i]) return false;
    ns_out[i] = ns, slab_len_out[i] = (int)len;
    total += ns * nr;
  }
  return total >= 160 && total <= 256;
}

int launch_tn_rows(TnRowsItem* items, int n, hipStream_t stream) {
  if (n <= 0) return SOW_OK;
  if (n > TNR_MAXI) return SOW_ERR_SHAPE;
  TnRowsGroup g{};
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    TnRowsItem& it = items[i];
    if ((reinterpret_cast<uintptr_t>(it.M) & 15) || (reinterpret_cast<uintptr_t>(it.S) & 15) || it.ldm % 8 || it.D % 8 ||
        it.slab_len % 32 || it.ns < 1)
      return SOW_ERR_ALIGN;
    it.ncg = (it.D + 63) / 64;
    it.nr = (it.ncg + 15) / 16;
    it.gpr = (it.ncg + it.nr - 1) / it.nr;
    it.cgw = it.gpr > TNR_WAVES ? 2 : 1;
    it.start = total;
    total += it.ns * it.nr;
    g.it[i] = it;
  }
  g.total = total;
  // M (x, dY) is streamed once: non-temporal loads keep it out of the Infinity Cache, where dh, the slab partials and the next
  // kernels' operands live (the launch itself 135 -> 138 us, the step 3.82 -> 3.78 ms; TN_NO_NT_LOAD switch)
  g.nt_load = sw_on(SW_TN_NO_NT_LOAD) ? 0 : 1;
  SOW_SET_MAX_LDS_ONCE(TNR_LDS, tn_partial_rows_kernel);
  hipLaunchKernelGGL(tn_partial_rows_kernel, dim3((unsigned)(total < 256 ? total : 256)), dim3(64 * TNR_WAVES), TNR_LDS, stream, g);
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

int launch_tn(const TnParams& p, int dtype, hipStream_t stream) {
  int blocks = 0;
  for (int j = 0; j < p.njobs; ++j) blocks += p.job[j].ncg * p.ns;
  if (blocks == 0) return SOW_OK;
  if (dtype == SOW_BF16) {
    const bool dma = tn_wide_ok(p);
    if (dma && !sw_on(SW_TN_NARROW)) {
      return launch_tn_group(&p, 1, stream);
    } else if (dma) {
      constexpr int LDS = 4 * TN_DEPTH * TN_STAGE_BYTES;  // 64 KiB (rings; reused by the cross-wave sum)
      SOW_SET_MAX_LDS_ONCE(LDS, tn_partial_dma_kernel);
      hipLaunchKernelGGL(tn_partial_dma_kernel, dim3(blocks), dim3(256), LDS, stream, p);
    } else {
      hipLaunchKernelGGL(tn_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, p);
    }
  } else if (dtype == SOW_F32) {
    bool dma = p.slab_len % 8 == 0;
    for (int j = 0; j < p.njobs; ++j) {
      const TnJob& J = p.job[j];
      dma = dma && J.D % 4 == 0 && J.ldm % 4 == 0 && (reinterpret_cast<uintptr_t>(J.M) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(J.S) & 15) == 0 && J.ones_col_in_s;
    }
    if (dma && tn_f32q_ok(p)) return launch_tn_f32q(p, stream);
    if (dma && !sw_on(SW_F32_EXACT) && !sw_on(SW_TN_NARROW)) {
      constexpr int LDSW = TNFW_WAVES * TNFW_DEPTH * TNFW_STAGE_BYTES;  // 144 KiB
      int blocks2 = 0;   // two 64-column groups per block
      for (int j = 0; j < p.njobs; ++j) blocks2 += (p.job[j].ncg + 1) / 2 * p.ns;
      SOW_SET_MAX_LDS_ONCE(LDSW, tn_partial_dma_f32_wide_kernel);
      hipLaunchKernelGGL(tn_partial_dma_f32_wide_kernel, dim3(blocks2), dim3(64 * TNFW_WAVES), LDSW, stream, p);
    } else if (dma) {
      constexpr int LDS = 4 * TNF_DEPTH * TNF_STAGE_BYTES;  // 64 KiB
      if (sw_on(SW_F32_EXACT)) {
        SOW_SET_MAX_LDS_ONCE(LDS, tn_partial_dma_f32_kernel<false>);
        hipLaunchKernelGGL(tn_partial_dma_f32_kernel<false>, dim3(blocks), dim3(256), LDS, stream, p);
      } else {
        SOW_SET_MAX_LDS_ONCE(LDS, tn_partial_dma_f32_kernel<true>);
        hipLaunchKernelGGL(tn_partial_dma_f32_kernel<true>, dim3(blocks), dim3(256), LDS, stream, p);
      }
    } else {
      hipLaunchKernelGGL(tn_partial_kernel<float>, dim3(blocks), dim3(256), 0, stream, p);
    }
  } else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

int launch_tn_reduce_batch(const ReduceParams* descs, const int* starts, int n, int total_blocks, int dtype, hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0) return SOW_OK;
  if (dtype == SOW_BF16)
    hipLaunchKernelGGL(tn_reduce_batch_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, stream, descs, starts, n);
  else if (dtype == SOW_F32)
    hipLaunchKernelGGL(tn_reduce_batch_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, descs, starts, n);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

int launch_tn_reduce(ReduceParams p, int dtype, hipStream_t stream) {
  p.blocks0 = (p.job[0].D + 3) / 4;
  int blocks = p.blocks0;
  if (p.njobs > 1) blocks += (p.job[1].D + 3) / 4;
  if (blocks == 0) return SOW_OK;
  if (dtype == SOW_BF16)
    hipLaunchKernelGGL(tn_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, p);
  else if (dtype == SOW_F32)
    hipLaunchKernelGGL(tn_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, p);
  else
    return SOW_ERR_DTYPE;
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

}  // namespace sow
