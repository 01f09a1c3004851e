// bf16 streaming GEMM that computes its own K-extension operand (gfx950):
//     h = hscale * X . op(F)                    [M, 64]  (saved to H by the workgroups of the first column tile)
//     C = [X, h] . [op(W); op(G)] + bias        [M, N]
//
// The dense-accumulator form of the SoW layer (tn_gradient/layer/sow.py:109-121) in ONE launch:
//     forward  (NN): y  = x  . W_acc   + (scale * x  . A  ) . B      F = A [K, r],   G = B [r, N]
//     backward (NT): dX = dY . W_acc^T + (scale * dY . B^T) . A^T    F = B [r, K],   G = A [N, r]
// gemm2.hip needs h from a separate H-only chain launch that streams X a second time (13 us of a 40-us layer at
// 512 -> 512); here every workgroup projects its own 256 rows while X passes through LDS anyway: the factor's
// k-slice rides along as a 4-KiB piece of every stage (+12 % LDS-DMA bytes), each wave adds 4 MFMAs to its 16
// (H^T tile of 32 of the workgroup's rows, accumulator registers = rank rows, so four consecutive ranks pack into
// one 8-byte LDS write), and after the last main stage the bf16 h tile is written into the X-piece regions of the
// two extension stages, which then run as ordinary stages.  The projection is recomputed by every column tile, so
// the host uses this kernel only when N spans at most two of them.
//
// The [rows, r] factor (A in both directions) has 2r-byte rows that no 16-byte DMA piece can address exactly:
// pieces are read from the row start in 16-byte steps (gfx950 needs 4-byte alignment only, tools/probe2.hip), the
// piece that straddles column r carries the head of the next row in its tail -- those columns meet an explicit
// zero (masked ranks of h in the forward, zero ranks of the LDS dh tile in the backward) -- pieces that would cross
// the end of the buffer read the zero page, and the valid head of the last row's straddling piece is rewritten
// from a guarded load by the wave that issued that DMA (after its own counted wait, before the stage barrier).
#include "kernels.hpp"
#include "epilogue.hpp"
#include "lds_dma.hpp"
#include <cstdlib>
#include <type_traits>

namespace sow {

constexpr int GH_BM = 256, GH_BN = 256, GH_BK = 32;
constexpr int GH_THREADS = 512;
constexpr int GH_NSLOT = 4;
constexpr int GH_PIECE = 256 * GH_BK * 2;          // 16 KiB: X or W piece of a stage
constexpr int GH_FPIECE = 64 * GH_BK * 2;          // 4 KiB: projection-factor piece
constexpr int GH_STAGE = 2 * GH_PIECE + GH_FPIECE; // 36 KiB
constexpr int GH_LDS = GH_NSLOT * GH_STAGE;        // 144 KiB

struct Gemm2hParams {
  const bf16_t* X;     // [M, K]
  const bf16_t* W;     // NT: [N, K]; NN: [K, N]
  const bf16_t* F;     // NN: [K, r] contiguous; NT: [r, K], ld = ldf
  const bf16_t* G;     // NN: [r, N], ld = ldg;  NT: [N, r] contiguous
  bf16_t* C;
  const bf16_t* bias;
  bf16_t* H;           // [M, 64]: scaled h, zeros, 1.0 in column 63 when r <= 63 (skinny-TN dbias trick)
  int64_t M, ldx, ldw, ldf, ldg, ldc;
  int N, K, r;
  float hscale;
};

#define DS_WRITE_B64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define DS_WRITE_B32(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")

// chunk swizzle of a k-major [k][64 n] image with 128-byte rows read by ds_read_b64_tr_b16 (as chain2.hip)
__device__ __forceinline__ int gh_img8(int row, int c) { return c ^ (((row >> 1) & 1) << 2); }

__device__ __forceinline__ void vm_wait_le(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Grouped launch: the tile lists of up to GH_MAXG independent layers (q / k / v; gate / up) back to back in one grid --
// every workgroup runs its own layer's tile unchanged (bit-identical results); prologue, epilogue and the write drain
// of different workgroups overlap across the group instead of every layer paying its own launch ramp.
constexpr int GH_MAXG = 4;
struct Gemm2hGroup {
  Gemm2hParams p[GH_MAXG];
  int start[GH_MAXG + 1];
  int n;
};

template <bool NT> __global__ __launch_bounds__(GH_THREADS, 1) void gemm2h_kernel(const Gemm2hGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 2, wn = w & 3, li = lane & 31, lh = lane >> 5;
  int layer = 0;
#pragma unroll
  for (int i = 1; i < GH_MAXG; ++i)
    if (i < grp.n && (int)blockIdx.x >= grp.start[i]) layer = i;
  const Gemm2hParams& p = grp.p[layer];
  const int tiles_n = (p.N + GH_BN - 1) / GH_BN;
  const int lid = xcd_remap((int)blockIdx.x - grp.start[layer], grp.start[layer + 1] - grp.start[layer]);
  const int64_t m0 = (int64_t)(lid / tiles_n) * GH_BM;
  const int n0 = (lid % tiles_n) * GH_BN;
  const int K = p.K, N = p.N, r = p.r;
  const int64_t M = p.M;
  const int s_main = (K + GH_BK - 1) / GH_BK;
  const int S = s_main + 2;
  const char* zp = zero_page_for(lane);

  // ---------------------------------------------------------------- the ragged [rows, r] factor and its fix-up
  const bf16_t* R = NT ? p.G : p.F;
  const int r_rows = NT ? N : K;
  const char* r_end = (const char*)(R + (int64_t)r_rows * r);
  const int gc = r >> 3;             // 16-byte piece of a row that holds column r (partial when r % 8 != 0)
  const int nfix = (r & 7) >> 1;     // its valid dwords
  bool own_fix = false;
  int fix_stage = -1;
  uint32_t fix_off = 0u, fix_dw = 0u;
  if (nfix > 0) {
    if constexpr (NT) {
      const int lr = N - 1 - n0;     // G's last row inside this column tile?
      if (lr >= 0 && lr < GH_BN && w == (lr >> 5)) {
        own_fix = true;
        fix_stage = s_main + (gc >> 2);
        fix_off = (uint32_t)(GH_PIECE + lr * 64 + (((gc & 3) ^ ((lr >> 2) & 3)) * 16) + 4 * lane);
      }
    } else {
      const int lr = K - 1 - GH_BK * (s_main - 1);
      if (w == (lr >> 3)) {
        own_fix = true;
        fix_stage = s_main - 1;
        fix_off = (uint32_t)(2 * GH_PIECE + lr * 128 + gh_img8(lr, gc) * 16 + 4 * lane);
      }
    }
    if (own_fix && lane < nfix) fix_dw = *((const uint32_t*)(R + (int64_t)(r_rows - 1) * r + 8 * gc) + lane);
    asm volatile("" : "+v"(fix_dw));   // consume now: the wait for this load lands here, not inside the pipeline
  }

  // ---------------------------------------------------------------- DMA sources (per lane), as gemm2.hip
  const int crow = 16 * (2 * w) + (lane >> 2);
  const int cpc = lane & 3;
  const int krow = 2 * (2 * w) + (lane >> 5);
  const int kpc = lane & 31;
  const int per_main = 4 + (w < 4 ? 1 : 0);   // this wave's DMA instructions per main stage; 2 per extension stage
  auto issue = [&](int s) {
    char* slot = smem + (s % GH_NSLOT) * GH_STAGE;
    const bool ext = s >= s_main;
    const int k0 = (ext ? s - s_main : s) * GH_BK;
    if (!ext) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int row = crow + 16 * ii;
        const int lc = cpc ^ ((row >> 2) & 3);
        const int64_t gr = m0 + row;
        const void* src = (gr < M && k0 + 8 * lc < K) ? (const void*)(p.X + gr * p.ldx + k0 + 8 * lc) : (const void*)zp;
        dma16(src, slot + (2 * w + ii) * 1024);
      }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const void* src;
      if constexpr (NT) {
        const int row = crow + 16 * ii;
        const int lc = cpc ^ ((row >> 2) & 3);
        const int gn = n0 + row;
        if (!ext) {
          src = (gn < N && k0 + 8 * lc < K) ? (const void*)(p.W + (int64_t)gn * p.ldw + k0 + 8 * lc) : (const void*)zp;
        } else {
          const char* q = (const char*)(p.G + (int64_t)gn * r + k0 + 8 * lc);
          src = (gn < N && k0 + 8 * lc < r && q + 16 <= r_end) ? (const void*)q : (const void*)zp;
        }
      } else {
        const int row = krow + 2 * ii;
        const int lc = kpc ^ ((row & 3) << 2);
        const int gk = k0 + row, gn = n0 + 8 * lc;
        if (!ext)
          src = (gk < K && gn < N) ? (const void*)(p.W + (int64_t)gk * p.ldw + gn) : (const void*)zp;
        else
          src = (gk < r && gn < N) ? (const void*)(p.G + (int64_t)gk * p.ldg + gn) : (const void*)zp;
      }
      dma16(src, slot + GH_PIECE + (2 * w + ii) * 1024);
    }
  };
  // projection-factor piece of main stage s (waves 0..3, one instruction each).  Issued at the END of a step, when the
  // CU's LDS fill path is idle: in the burst after the barrier a fifth instruction keeps these four waves ~190 cycles
  // longer in the issue and everyone waits for them at the next barrier (measured 0.1-0.16 us per stage).
  auto issue_f = [&](int s) {
    char* slot = smem + (s % GH_NSLOT) * GH_STAGE;
    const int k0 = s * GH_BK;
    if (s < s_main && w < 4) {
      const void* src;
      if constexpr (NT) {   // [64 rank rows][32 k], 64-byte rows
        const int row = 16 * w + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        src = (row < r && k0 + 8 * lc < K) ? (const void*)(p.F + (int64_t)row * p.ldf + k0 + 8 * lc) : (const void*)zp;
      } else {              // [32 k][64 ranks], 128-byte rows read from 2r-byte rows
        const int row = 8 * w + (lane >> 3);
        const int lc = gh_img8(row, lane & 7);
        const int gk = k0 + row;
        const char* q = (const char*)(p.F + (int64_t)gk * r) + 16 * lc;
        src = (gk < K && 8 * lc < r && q + 16 <= r_end) ? (const void*)q : (const void*)zp;
      }
      dma16(src, slot + 2 * GH_PIECE + w * 1024);
    }
  };
  auto cnt = [&](int s) { return s < s_main ? per_main : 2; };

  // ---------------------------------------------------------------- fragment addresses (per lane)
  const uint32_t base = lds_addr(smem);
  const int fsw = (li >> 2) & 3;
  // accumulator mi of wave (wm, wn) is row tile mi ^ wn of its 128-row half, so that the four waves of a half project
  // four different row tiles from their af[.][0] without a register select
  const uint32_t a_off = (uint32_t)((wm * 128 + li) * 64);   // + ((mi ^ wn) * 2048)
  uint32_t b_off[2], f_off[2];
  if constexpr (NT) {
    b_off[0] = (uint32_t)(GH_PIECE + (wn * 64 + li) * 64);
    b_off[1] = b_off[0] + 2048;
    f_off[0] = (uint32_t)(2 * GH_PIECE + li * 64);
    f_off[1] = f_off[0] + 2048;
  } else {
    const int g = lane >> 4, jj = lane & 15, q = jj >> 2, pp = jj & 3;
    const int r1 = 8 * (g >> 1) + q;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = wn * 64 + ni * 32 + 16 * (g & 1) + 4 * pp;
      b_off[ni] = (uint32_t)(GH_PIECE + r1 * 512 + (((col >> 3) ^ ((r1 & 3) << 2)) * 16) + (col & 7) * 2);
      const int fc = ni * 32 + 16 * (g & 1) + 4 * pp;
      f_off[ni] = (uint32_t)(2 * GH_PIECE + r1 * 128 + gh_img8(r1, fc >> 3) * 16 + (fc & 7) * 2);
    }
  }
  const uint32_t ch0 = (uint32_t)(((0 + lh) ^ fsw) * 16), ch1 = (uint32_t)(((2 + lh) ^ fsw) * 16);

  f32x16 acc[4][2];
  f32x16 hT[2];   // H^T tiles of rows wm*128 + wn*32 .. +31: lane = row, registers = rank rows of tile rt
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) hT[a][i] = 0.f;

  for (int s = 0; s < GH_NSLOT - 1; ++s) issue(s), issue_f(s);   // S >= 6 (host: K >= 128)

  // one pipeline step: wait for stage s, barrier, refill the ring, multiply.  MAIN stages also project.
  auto step = [&](int s, auto main_tag) {
    constexpr bool MAIN = decltype(main_tag)::value;
    // stages issued after s: s+1, s+2
    vm_wait_le((s + 1 < S ? cnt(s + 1) : 0) + (s + 2 < S ? cnt(s + 2) : 0));
    if (own_fix && s == fix_stage && lane < nfix) {
      DS_WRITE_B32(base + (uint32_t)((s % GH_NSLOT) * GH_STAGE) + fix_off, fix_dw, 0);
    }
    raw_barrier();   // stage s complete for everyone (DMA pieces, fix-up, h tile); everyone is done with stage s-1
    if (s + GH_NSLOT - 1 < S) issue(s + GH_NSLOT - 1);
    const uint32_t sb = base + (uint32_t)((s % GH_NSLOT) * GH_STAGE);
    u32x4 af[2][4], bf[2][2], ff[2];   // ff: factor fragments of ONE k-step (the second is fetched under the MFMAs)
    u32x2 fl[2], fh[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t aa = sb + a_off + (ks ? ch1 : ch0);
      DS_READ_B128(af[ks][0], aa + (uint32_t)(((0 ^ wn) * 2048)), 0);
      DS_READ_B128(af[ks][1], aa + (uint32_t)(((1 ^ wn) * 2048)), 0);
      DS_READ_B128(af[ks][2], aa + (uint32_t)(((2 ^ wn) * 2048)), 0);
      DS_READ_B128(af[ks][3], aa + (uint32_t)(((3 ^ wn) * 2048)), 0);
      if constexpr (NT) {
        const uint32_t bb = sb + (ks ? ch1 : ch0);
        DS_READ_B128(bf[ks][0], bb + b_off[0], 0);
        DS_READ_B128(bf[ks][1], bb + b_off[1], 0);
      }
    }
    if constexpr (MAIN) {
      if constexpr (NT) {
        DS_READ_B128(ff[0], sb + ch0 + f_off[0], 0);
        DS_READ_B128(ff[1], sb + ch0 + f_off[1], 0);
      } else {
        DS_READ_TR(fl[0], sb + f_off[0], 0);
        DS_READ_TR(fh[0], sb + f_off[0], 512);
        DS_READ_TR(fl[1], sb + f_off[1], 0);
        DS_READ_TR(fh[1], sb + f_off[1], 512);
      }
    }
    if constexpr (!NT) {
      u32x2 bl[2][2], bh[2][2];
      DS_READ_TR(bl[0][0], sb + b_off[0], 0);
      DS_READ_TR(bh[0][0], sb + b_off[0], 2048);
      DS_READ_TR(bl[0][1], sb + b_off[1], 0);
      DS_READ_TR(bh[0][1], sb + b_off[1], 2048);
      DS_READ_TR(bl[1][0], sb + b_off[0], 8192);
      DS_READ_TR(bh[1][0], sb + b_off[0], 8192 + 2048);
      DS_READ_TR(bl[1][1], sb + b_off[1], 8192);
      DS_READ_TR(bh[1][1], sb + b_off[1], 8192 + 2048);
      LGKM_WAIT0();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[ks][ni] = join2(bl[ks][ni], bh[ks][ni]);
      if constexpr (MAIN) ff[0] = join2(fl[0], fh[0]), ff[1] = join2(fl[1], fh[1]);
    } else {
      LGKM_WAIT0();
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma32(as_bf16x8(af[0][mi]), as_bf16x8(bf[0][ni]), acc[mi][ni]);
    if constexpr (MAIN) {
      // projection: H^T[rt] += F^T[rt] . X^T for the wave's first row tile (= row tile wn of its half, see the reads)
      hT[0] = mfma32(as_bf16x8(ff[0]), as_bf16x8(af[0][0]), hT[0]);
      hT[1] = mfma32(as_bf16x8(ff[1]), as_bf16x8(af[0][0]), hT[1]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NT) {   // lands under the eight MFMAs of the second k-step
        DS_READ_B128(ff[0], sb + ch1 + f_off[0], 0);
        DS_READ_B128(ff[1], sb + ch1 + f_off[1], 0);
      } else {
        DS_READ_TR(fl[0], sb + f_off[0], 2048);
        DS_READ_TR(fh[0], sb + f_off[0], 2048 + 512);
        DS_READ_TR(fl[1], sb + f_off[1], 2048);
        DS_READ_TR(fh[1], sb + f_off[1], 2048 + 512);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma32(as_bf16x8(af[1][mi]), as_bf16x8(bf[1][ni]), acc[mi][ni]);
    if constexpr (MAIN) {
      LGKM_WAIT0();
      if constexpr (!NT) ff[0] = join2(fl[0], fh[0]), ff[1] = join2(fl[1], fh[1]);
      hT[0] = mfma32(as_bf16x8(ff[0]), as_bf16x8(af[1][0]), hT[0]);
      hT[1] = mfma32(as_bf16x8(ff[1]), as_bf16x8(af[1][0]), hT[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MAIN) issue_f(s + GH_NSLOT - 1);   // same stage as the issue() after this step's barrier
  };

#pragma unroll 1
  for (int s = 0; s < s_main; ++s) step(s, std::true_type{});

  {
    // h tile -> X-piece regions of the two extension stages (rank tile rt = k range of extension stage rt).
    // Their slots held stages s_main-4 / s_main-3, drained before earlier barriers; the extension DMAs only
    // touch the W-piece regions.  Visible to the other waves after the barrier of step s_main.
    const int row_l = wm * 128 + wn * 32 + li;
    const int sw = (row_l >> 2) & 3;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const uint32_t areg = base + (uint32_t)(((s_main + rt) % GH_NSLOT) * GH_STAGE) + (uint32_t)(row_l * 64 + 8 * lh);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rank0 = rt * 32 + 8 * j + 4 * lh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (rank0 + e < r) ? hT[rt][4 * j + e] * p.hscale : 0.f;
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        DS_WRITE_B64(areg + (uint32_t)((j ^ sw) * 16), pk, 0);
      }
    }
  }
  step(s_main, std::false_type{});
  step(s_main + 1, std::false_type{});

  // ---------------------------------------------------------------- epilogue
  // Transpose scratch = the slots of stages s_main-2 / s_main-1 (four waves each): every wave finished with them
  // before barrier s_main+1, which this wave has passed, and the h tile lives in the other two slots -- no barrier.
  if ((m0 + wm * 128 < M) && (n0 + wn * 64 < N)) {
    float* scratch = (float*)(smem + ((s_main + 2 + (w >> 2)) % GH_NSLOT) * GH_STAGE + (w & 3) * (EpiScratch<2>::FLOATS * 4));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      wave_store_tiles<bf16_t, 2, true>(acc[mi], scratch, p.C, p.ldc, m0 + wm * 128 + (mi ^ wn) * 32, n0 + wn * 64, M, N, 1.f,
                                        0.f, p.bias, lane, SOW_GEMM_NT(M));
  }
  if (tiles_n == 2 || n0 == 0) {
    // saved copy of h for the weight-gradient kernels, AFTER the C stores (hipcc guards the scratch reads above with
    // vmcnt(0), which would expose the latency of these stores).  One column tile: thread -> (row, extension stage),
    // four 16-byte pieces; two column tiles: tile j saves extension stage j (ranks 32j .. 32j+31), two pieces a thread.
    const int row = t >> 1;
    const bool split = tiles_n == 2;
    const int e = split ? (lid % tiles_n) : (t & 1);
    const int c0 = split ? 2 * (t & 1) : 0;
    const int64_t grow = m0 + row;
    const uint32_t hreg = base + (uint32_t)(((s_main + e) % GH_NSLOT) * GH_STAGE) + (uint32_t)(row * 64);
    const int sw = (row >> 2) & 3;
    u32x4 hv[4];
    DS_READ_B128(hv[0], hreg + (uint32_t)(((c0 + 0) ^ sw) * 16), 0);
    DS_READ_B128(hv[1], hreg + (uint32_t)(((c0 + 1) ^ sw) * 16), 0);
    if (!split) {
      DS_READ_B128(hv[2], hreg + (uint32_t)((2 ^ sw) * 16), 0);
      DS_READ_B128(hv[3], hreg + (uint32_t)((3 ^ sw) * 16), 0);
    }
    LGKM_WAIT0();
    if (e == 1 && r < 64) {   // column 63 <- 1.0
      if (split) {
        if (c0 == 2) hv[1][3] = (hv[1][3] & 0xffffu) | 0x3F800000u;
      } else {
        hv[3][3] = (hv[3][3] & 0xffffu) | 0x3F800000u;
      }
    }
    if (grow < M) {
      u32x4* dst = (u32x4*)(p.H + grow * 64 + e * 32) + c0;
      dst[0] = hv[0];
      dst[1] = hv[1];
      if (!split) dst[2] = hv[2], dst[3] = hv[3];
    }
  }
}

// ------------------------------------------------------------------------------------------------
static bool gh_al(const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; }

bool gemm2h_supported(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                      const void* G, int64_t ldg, const void* C, int64_t ldc, const void* bias, const void* H, int64_t M,
                      int N, int K, int r, int dtype) {
  if (dtype != SOW_BF16 || !X || !W || !F || !G || !C || !H) return false;
  if (sw_on(SW_NO_FUSED_H) || sw_on(SW_FORCE_GEMM_V1)) return false;   // A/B switches
  // r >= 4: a 16-byte piece spans at most two 2r-byte rows, so only the LAST row has pieces crossing the end of the buffer
  if (r < 4 || r > 64 || (r & 1)) return false;
  if (N < 64 || K < 128) return false;
  const int tiles_n = ceil_div(N, GH_BN);
  // every column tile recomputes the projection (+25 % MFMA, +12 % DMA per tile): pays for one or two of them
  if (tiles_n > 2 || (int64_t)ceil_div(M, GH_BM) * tiles_n < 160) return false;
  if (K % 8 || N % 8 || ldx % 8 || ldw % 8 || ldc % 8) return false;
  if (!gh_al(X, 16) || !gh_al(W, 16) || !gh_al(C, 16) || !gh_al(H, 16) || (bias && !gh_al(bias, 16))) return false;
  if (nt) {
    if (ldf % 8 || !gh_al(F, 16) || !gh_al(G, 4)) return false;
  } else {
    if (ldg % 8 || !gh_al(G, 16) || !gh_al(F, 4)) return false;
  }
  return true;
}

int launch_gemm2h_group(const Gemm2hArgs* a, int n, bool nt, hipStream_t stream) {
  if (n <= 0) return SOW_OK;
  if (n > GH_MAXG) return SOW_ERR_SHAPE;
  Gemm2hGroup g{};
  g.n = n;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    Gemm2hParams& p = g.p[i];
    p.X = (const bf16_t*)a[i].X, p.W = (const bf16_t*)a[i].W, p.F = (const bf16_t*)a[i].F, p.G = (const bf16_t*)a[i].G;
    p.C = (bf16_t*)a[i].C, p.bias = (const bf16_t*)a[i].bias, p.H = (bf16_t*)a[i].H;
    p.M = a[i].M, p.ldx = a[i].ldx, p.ldw = a[i].ldw, p.ldf = a[i].ldf, p.ldg = a[i].ldg, p.ldc = a[i].ldc;
    p.N = a[i].N, p.K = a[i].K, p.r = a[i].r, p.hscale = a[i].hscale;
    g.start[i] = (int)total;
    total += (int64_t)ceil_div(p.M, GH_BM) * ceil_div(p.N, GH_BN);
  }
  for (int i = n; i <= GH_MAXG; ++i) g.start[i] = (int)total;
  if (total <= 0) return SOW_OK;
  if (total > 0x7fffffff) return SOW_ERR_SHAPE;
  if (nt) {
    SOW_SET_MAX_LDS_ONCE(GH_LDS, gemm2h_kernel<true>);
    hipLaunchKernelGGL(gemm2h_kernel<true>, dim3((unsigned)total), dim3(GH_THREADS), GH_LDS, stream, g);
  } else {
    SOW_SET_MAX_LDS_ONCE(GH_LDS, gemm2h_kernel<false>);
    hipLaunchKernelGGL(gemm2h_kernel<false>, dim3((unsigned)total), dim3(GH_THREADS), GH_LDS, stream, g);
  }
  SOW_CHECK_LAUNCH();
  return SOW_OK;
}

int launch_gemm2h(const void* X, int64_t ldx, const void* W, int64_t ldw, bool nt, const void* F, int64_t ldf,
                  const void* G, int64_t ldg, void* C, int64_t ldc, const void* bias, void* H, int64_t M, int N, int K,
                  int r, float hscale, hipStream_t stream) {
  const Gemm2hArgs a{X, W, F, G, C, bias, H, M, ldx, ldw, ldf, ldg, ldc, N, K, r, hscale};
  return launch_gemm2h_group(&a, 1, nt, stream);
}

}  // namespace sow
