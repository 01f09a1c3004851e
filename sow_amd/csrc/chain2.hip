// Fused low-rank chain, bf16 streaming version (gfx950):  Y = beta*Y + (scale * X.F1).F2 + bias
//
// Same contract as chain.hip (reference tn_gradient/layer/sow.py:107-126 forward, and its autograd
// backward with F1 = B^T, F2 = A^T), rebuilt around the two CDNA4 features that matter for an
// HBM-bound kernel:
//   * LDS-DMA (`global_load_lds_dwordx4`): every compute wave streams its own 32 token rows of X through
//     a private 4-deep ring of [32 x 64] stages -- no VGPR staging, no workgroup barrier on the X path,
//     16 KiB in flight per wave behind a COUNTED s_waitcnt vmcnt;
//   * `ds_read_b64_tr_b16`: the factors stay in their storage layout in LDS (A is [d_in, r], B is
//     [r, d_out]); whichever of them has the contraction index as its row index is read transposed.
// Workgroup = 5 waves: 4 compute waves (32 tokens each, 128 per workgroup) + 1 loader wave that
// double-buffers the factor chunks (256 rows of F1, then 256 columns of F2) in LDS, one raw
// s_barrier per chunk.  Per compute wave: phase 1 accumulates H[32,64] over K, H is scaled, rounded
// and parked in the wave's own LDS (it never leaves the CU except as the saved copy for backward),
// phase 2 produces Y in 64-column slices written as 16-byte row segments.
// All LDS reads of the compute waves are inline asm: for a compiler-visible LDS read hipcc emits
// `s_waitcnt vmcnt(0)` while LDS-DMA is outstanding, which would drain the ring every step.
//
// LDS image conventions (16-byte chunk c of a row -> physical chunk):
//   X stage / H / bwd-F2   rows of 128 B, ds_read_b128:      c ^ ((row >> 1) & 7)
//   fwd-F1 (A, [k][64])    rows of 128 B, transposed read:   c ^ (((row >> 1) & 1) << 2)
//   fwd-F2 (B, [r][256])   rows of 512 B, transposed read:   c ^ ((row & 3) << 2)
//   bwd-F1 (B, [r][256])   rows of 512 B, ds_read_b128:      c ^ (row & 15)
// DMA writes LDS lane-linearly, so the XOR is applied to the per-lane SOURCE address.
#include "kernels.hpp"

namespace sow {

constexpr int C2_BM = 128;            // tokens per workgroup
constexpr int C2_KC = 256;            // factor chunk: rows of F1 / columns of F2
constexpr int C2_DEPTH = 4;           // X stages in flight per compute wave
constexpr int C2_STAGE = 4096;        // [32 tok][64 k] bf16
constexpr int C2_FSLOT = 32768;       // [256][64] or [64][256] bf16
constexpr int C2_RING0 = 2 * C2_FSLOT;
constexpr int C2_RING = C2_DEPTH * C2_STAGE;  // 16 KiB per compute wave
constexpr int C2_LDS = C2_RING0 + 4 * C2_RING;
constexpr int C2_THREADS = 320;

__device__ __attribute__((aligned(256))) uint32_t g_zero_page2[64];

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
#define DS_READ_B128(dst, addr, off) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
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
__device__ __forceinline__ void wait_x_stages(int newer) {
  switch (newer) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 join_tr(u32x2 lo, u32x2 hi) {
  return __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
}

// ------------------------------------------------------------------------------------------------
// Loader wave: factor chunk `ci` of the unified sequence [F1 chunk 0.. | F2 chunk 0..] -> slot
// ------------------------------------------------------------------------------------------------
// A-derived chunk: source rows of `rb` bf16 (dword = 2 elements), image rows of 128 B (64 elements),
// `tr` selects the transposed-read swizzle (fwd F1) or the b128 swizzle (bwd F2).
template <bool TR>
__device__ __forceinline__ void load_a_chunk(const bf16_t* A, int64_t ld, int rows_total, int row0, int rb, char* slot,
                                             int lane) {
  const int dw_per_row = rb >> 1;  // rb is even on this path
  // 256 rows x 32 dwords (25 valid for r = 50, zero padding written explicitly); lanes walk the dword
  // columns fastest.  32 loads are kept in flight per lane so a chunk costs ~4 L2 round trips.
#pragma unroll 1
  for (int b = 0; b < (C2_KC * 32) / (64 * 32); ++b) {
    uint32_t v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int idx = lane + 64 * (b * 32 + u);
      const int row = idx >> 5, j = idx & 31;
      v[u] = (j < dw_per_row && row0 + row < rows_total) ? *(const uint32_t*)(A + (int64_t)(row0 + row) * ld + 2 * j) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int idx = lane + 64 * (b * 32 + u);
      const int row = idx >> 5, j = idx & 31;
      const int c = j >> 2;
      const int pc = TR ? (c ^ (((row >> 1) & 1) << 2)) : (c ^ ((row >> 1) & 7));
      *(uint32_t*)(slot + row * 128 + pc * 16 + (j & 3) * 4) = v[u];
    }
  }
}
// B-derived chunk: rows r (64, zero beyond rb), 256 columns starting at col0; DMA, 2 rows per instruction.
template <bool TR>
__device__ __forceinline__ void load_b_chunk(const bf16_t* B, int64_t ld, int cols_total, int col0, int rb, char* slot,
                                             int lane) {
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    const int row = 2 * i + (lane >> 5), pc = lane & 31;
    const int lc = TR ? (pc ^ ((row & 3) << 2)) : (pc ^ (row & 15));
    const int col = col0 + lc * 8;
    const void* src = (row < rb && col < cols_total) ? (const void*)(B + (int64_t)row * ld + col)
                                                      : (const void*)(g_zero_page2 + (lane & 7) * 4);
    dma16(src, slot + i * 1024);
  }
}

// =================================================================================================
template <bool BWD> __global__ __launch_bounds__(C2_THREADS, 1) void chain2_kernel(const ChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * C2_BM;
  const int D1 = p.D1, D2 = p.D2, rb = p.rb;
  const int nkc = (D1 + C2_KC - 1) / C2_KC;  // F1 chunks
  const int nnc = (D2 + C2_KC - 1) / C2_KC;  // F2 chunks
  const bf16_t* F1 = (const bf16_t*)p.F1b;
  const bf16_t* F2 = (const bf16_t*)p.F2b;

  if (w == 4) {
    // ------------------------------------------------------------------ loader wave
    auto load_chunk = [&](int ci) {
      char* slot = smem + (ci & 1) * C2_FSLOT;
      if (ci < nkc) {
        if constexpr (!BWD)
          load_a_chunk<true>(F1, p.ldf1b, D1, ci * C2_KC, rb, slot, lane);   // A [d_in, r] -> [k][64], tr reads
        else
          load_b_chunk<false>(F1, p.ldf1b, D1, ci * C2_KC, rb, slot, lane);  // B [r, d_out] -> [r][256 k], b128
      } else {
        const int nc = ci - nkc;
        if constexpr (!BWD)
          load_b_chunk<true>(F2, p.ldf2b, D2, nc * C2_KC, rb, slot, lane);   // B [r, d_out] -> [r][256 n], tr reads
        else
          load_a_chunk<false>(F2, p.ldf2b, D2, nc * C2_KC, rb, slot, lane);  // A [d_in, r] -> [n][64], b128
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    load_chunk(0);
    for (int ci = 0; ci < nkc + nnc; ++ci) {
      raw_barrier();                                 // chunk ci is complete; consumers finished chunk ci-1
      if (ci + 1 < nkc + nnc) load_chunk(ci + 1);    // into the slot chunk ci-1 just vacated
    }
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
  const int nst = (D1 + 63) / 64;  // X stages

  const int drow = lane >> 3, dpc = lane & 7;
  auto issue_x = [&](int st) {
    char* dst = ring + (st % C2_DEPTH) * C2_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + drow;
      const int lc = dpc ^ ((row >> 1) & 7);
      const int k = st * 64 + lc * 8;
      const int64_t tok = tok0 + row;
      const void* src = (tok < p.M && k < D1) ? (const void*)(X + tok * p.ldx + k) : (const void*)(g_zero_page2 + (lane & 7) * 4);
      dma16(src, dst + i * 1024);
    }
  };

  f32x16 hacc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[a][i] = 0.f;

  // per-lane LDS offsets
  const uint32_t xoff = (uint32_t)(li * 128);            // X stage / H row
  const int xsw = (li >> 1) & 7;                         // row swizzle of the b128 images with 128-B rows
  // fwd F1 (tr): row = 16*kk + 8*(g>>1) + 4*rd + q, col = rt*32 + 16*(g&1) + 4*pp
  // bwd F1 (b128): row r = rt*32 + li, chunk = 2*kk + lh
  uint32_t f1off[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    if constexpr (!BWD) {
      const int row = 8 * (g >> 1) + q, col = rt * 32 + 16 * (g & 1) + 4 * pp;
      const int pc = (col >> 3) ^ (((row >> 1) & 1) << 2);
      f1off[rt] = (uint32_t)(row * 128 + pc * 16 + (col & 7) * 2);
    } else {
      f1off[rt] = (uint32_t)((rt * 32 + li) * 512);
    }
  }

  const int pre = nst < C2_DEPTH ? nst : C2_DEPTH;
  for (int st = 0; st < pre; ++st) issue_x(st);

  // ================================================================== phase 1: H = X . F1
  for (int kc = 0; kc < nkc; ++kc) {
    raw_barrier();  // F1 chunk kc is in slot kc & 1
    const uint32_t fs = slot_a + (uint32_t)((kc & 1) * C2_FSLOT);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      const int st = kc * 4 + j;
      if (st >= nst) break;
      const int newer = (nst - 1 - st) < (C2_DEPTH - 1) ? (nst - 1 - st) : (C2_DEPTH - 1);
      wait_x_stages(newer);
      const uint32_t xs = ring_a + (uint32_t)((st % C2_DEPTH) * C2_STAGE) + xoff;
      u32x4 af[4];
      // A fragments: chunk (2*ks + lh) ^ xsw of this lane's row
      {
        const uint32_t a0 = xs + (uint32_t)(((0 + lh) ^ xsw) * 16), a1 = xs + (uint32_t)(((2 + lh) ^ xsw) * 16);
        const uint32_t a2 = xs + (uint32_t)(((4 + lh) ^ xsw) * 16), a3 = xs + (uint32_t)(((6 + lh) ^ xsw) * 16);
        DS_READ_B128(af[0], a0, 0);
        DS_READ_B128(af[1], a1, 0);
        DS_READ_B128(af[2], a2, 0);
        DS_READ_B128(af[3], a3, 0);
      }
      if constexpr (!BWD) {
        u32x2 bl[4][2], bh[4][2];  // [ks][rt] low / high 4 k
        const uint32_t b0 = fs + f1off[0] + (uint32_t)(j * 4 * 2048), b1 = fs + f1off[1] + (uint32_t)(j * 4 * 2048);
#define F1_TR(ks)                          \
  DS_READ_TR(bl[ks][0], b0, ks * 2048);    \
  DS_READ_TR(bh[ks][0], b0, ks * 2048 + 512); \
  DS_READ_TR(bl[ks][1], b1, ks * 2048);    \
  DS_READ_TR(bh[ks][1], b1, ks * 2048 + 512);
        F1_TR(0) F1_TR(1) F1_TR(2) F1_TR(3)
#undef F1_TR
        LGKM_WAIT0();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          hacc[0] = mfma32(as_bf16x8(af[ks]), join_tr(bl[ks][0], bh[ks][0]), hacc[0]);
          hacc[1] = mfma32(as_bf16x8(af[ks]), join_tr(bl[ks][1], bh[ks][1]), hacc[1]);
        }
      } else {
        u32x4 bf_[4][2];
        const int m = li & 15;  // (rt*32 + li) & 15
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int chunk = 8 * j + 2 * ks + lh;  // 0..31 inside the 256-k chunk
          const uint32_t o = (uint32_t)(((chunk & ~15) | ((chunk & 15) ^ m)) * 16);
          const uint32_t a0 = fs + f1off[0] + o, a1 = fs + f1off[1] + o;
          DS_READ_B128(bf_[ks][0], a0, 0);
          DS_READ_B128(bf_[ks][1], a1, 0);
        }
        LGKM_WAIT0();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          hacc[0] = mfma32(as_bf16x8(af[ks]), as_bf16x8(bf_[ks][0]), hacc[0]);
          hacc[1] = mfma32(as_bf16x8(af[ks]), as_bf16x8(bf_[ks][1]), hacc[1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (st + C2_DEPTH < nst) issue_x(st + C2_DEPTH);  // the stage's reads have returned (lgkmcnt(0) above)
    }
  }

  // ================================================================== hand-off: H -> wave-private LDS
  // every X stage of this wave has been consumed, so its ring is free: H image at +0 (4 KiB), fp32
  // epilogue scratch [32][68] at +4096.
  char* Himg = ring;
  float* scratch = (float*)(ring + 4096);
  constexpr int SLD = 68;
  {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int c = rt * 32 + li;
      const bool live = c < rb;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lane);
        float hv = live ? hacc[rt][reg] * p.scale : 0.f;
        // column 63 of the SAVED copy carries 1.0 (dbias trick of the skinny-TN kernel); the image used
        // by phase 2 must keep 0 there, so the 1.0 is patched in when the saved copy is written below
        *(bf16_t*)(Himg + row * 128 + (((c >> 3) ^ ((row >> 1) & 7)) * 16) + (c & 7) * 2) = (bf16_t)hv;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint32_t h_a = ring_a + xoff;
  u32x4 hf[4];
  DS_READ_B128(hf[0], h_a + (uint32_t)(((0 + lh) ^ xsw) * 16), 0);
  DS_READ_B128(hf[1], h_a + (uint32_t)(((2 + lh) ^ xsw) * 16), 0);
  DS_READ_B128(hf[2], h_a + (uint32_t)(((4 + lh) ^ xsw) * 16), 0);
  DS_READ_B128(hf[3], h_a + (uint32_t)(((6 + lh) ^ xsw) * 16), 0);
  if (p.Hsave) {
    bf16_t* Hs = (bf16_t*)p.Hsave;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), c = lane & 7;
      u32x4 v;
      DS_READ_B128(v, ring_a + (uint32_t)(row * 128 + ((c ^ ((row >> 1) & 7)) * 16)), 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c == 7 && rb < 64) v[3] = (v[3] & 0xffffu) | 0x3F800000u;  // element 63 <- bf16(1.0)
      const int64_t tok = tok0 + row;
      if (tok < p.M) *(u32x4*)(Hs + tok * 64 + c * 8) = v;
    }
  }
  LGKM_WAIT0();

  // ================================================================== phase 2: Y = H . F2
  bf16_t* Y = (bf16_t*)p.Y;
  const bf16_t* bias = (const bf16_t*)p.bias;
  const int ksteps = (rb + 15) / 16;
  for (int nc = 0; nc < nnc; ++nc) {
    raw_barrier();  // F2 chunk nc is in slot (nkc + nc) & 1
    const uint32_t fs = slot_a + (uint32_t)(((nkc + nc) & 1) * C2_FSLOT);
#pragma unroll 1
    for (int sub = 0; sub < 4; ++sub) {
      const int ncol0 = nc * C2_KC + sub * 64;  // first output column of this slice
      if (ncol0 >= D2) break;
      f32x16 yacc[2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[a][i] = 0.f;
      if constexpr (!BWD) {
        // F2 image [64 r][256 n] rows of 512 B, transposed reads: row r = 16*ks + 8*(g>>1) + 4*rd + q,
        // col n = sub*64 + nt*32 + 16*(g&1) + 4*pp ; physical chunk = (n>>3) ^ (q << 2)
        u32x2 bl[4][2], bh[4][2];
        uint32_t nb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int n = sub * 64 + nt * 32 + 16 * (g & 1) + 4 * pp;
          const int pc = (n >> 3) ^ (q << 2);
          nb[nt] = fs + (uint32_t)((8 * (g >> 1) + q) * 512 + pc * 16 + (n & 7) * 2);
        }
#define F2_TR(ks)                                  \
  DS_READ_TR(bl[ks][0], nb[0], ks * 8192);         \
  DS_READ_TR(bh[ks][0], nb[0], ks * 8192 + 2048);  \
  DS_READ_TR(bl[ks][1], nb[1], ks * 8192);         \
  DS_READ_TR(bh[ks][1], nb[1], ks * 8192 + 2048);
        F2_TR(0) F2_TR(1) F2_TR(2) F2_TR(3)
#undef F2_TR
        LGKM_WAIT0();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < ksteps) {
            yacc[0] = mfma32(as_bf16x8(hf[ks]), join_tr(bl[ks][0], bh[ks][0]), yacc[0]);
            yacc[1] = mfma32(as_bf16x8(hf[ks]), join_tr(bl[ks][1], bh[ks][1]), yacc[1]);
          }
        }
      } else {
        // F2 image [256 n][64 k] rows of 128 B, b128 reads: row n = sub*64 + nt*32 + li, chunk (2*ks+lh) ^ xsw
        u32x4 bf_[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint32_t o = (uint32_t)(((2 * ks + lh) ^ xsw) * 16);
          const uint32_t a0 = fs + (uint32_t)((sub * 64 + li) * 128) + o, a1 = a0 + 32 * 128;
          DS_READ_B128(bf_[ks][0], a0, 0);
          DS_READ_B128(bf_[ks][1], a1, 0);
        }
        LGKM_WAIT0();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks < ksteps) {
            yacc[0] = mfma32(as_bf16x8(hf[ks]), as_bf16x8(bf_[ks][0]), yacc[0]);
            yacc[1] = mfma32(as_bf16x8(hf[ks]), as_bf16x8(bf_[ks][1]), yacc[1]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- epilogue: accumulators -> wave-private fp32 scratch -> 16-byte row segments
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) scratch[acc_row(reg, lane) * SLD + nt * 32 + li] = yacc[nt][reg];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t sc_a = ring_a + 4096u;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 8 + (lane >> 3), c = (lane & 7) * 8;
        u32x4 v0, v1;
        const uint32_t sa = sc_a + (uint32_t)((r * SLD + c) * 4);
        DS_READ_B128(v0, sa, 0);
        DS_READ_B128(v1, sa, 16);
        LGKM_WAIT0();
        const int64_t tok = tok0 + r;
        const int col = ncol0 + c;
        if (tok < p.M && col < D2) {
          float v[8];
          const float* f0 = (const float*)&v0;
          const float* f1 = (const float*)&v1;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = f0[e], v[4 + e] = f1[e];
          bf16_t* dst = Y + tok * p.ldy + col;
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
          u32x4 pk;
          bf16_t* pe = (bf16_t*)&pk;
#pragma unroll
          for (int e = 0; e < 8; ++e) pe[e] = (bf16_t)v[e];
          *(u32x4*)dst = pk;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// =================================================================================================
bool chain2_supported(const ChainParams& p, int dtype) {
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  auto a4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; };
  if (dtype != SOW_BF16 || p.ra != 0 || p.rb <= 0 || p.rb > 64 || (p.rb & 1)) return false;
  if (p.D1 % 8 || p.D2 % 8 || p.ldx % 8 || p.ldy % 8) return false;
  if (!a16(p.X) || !a16(p.Y) || (p.bias && !a16(p.bias)) || (p.Hsave && !a16(p.Hsave))) return false;
  if (p.M < 4096) return false;  // short inputs: the 64-row generic kernel fills the chip better
  return true;
}

int launch_chain2(const ChainParams& p, bool bwd, hipStream_t stream) {
  // which factor is DMA-loaded (B, rows of D elements: 16-byte aligned rows) and which is dword-loaded (A)
  const void* Bp = bwd ? p.F1b : p.F2b;
  const int64_t ldB = bwd ? p.ldf1b : p.ldf2b;
  const void* Ap = bwd ? p.F2b : p.F1b;
  const int64_t ldA = bwd ? p.ldf2b : p.ldf1b;
  if ((reinterpret_cast<uintptr_t>(Bp) & 15) || ldB % 8 || (reinterpret_cast<uintptr_t>(Ap) & 3) || ldA % 2)
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
