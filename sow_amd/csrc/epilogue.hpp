// Wave-private accumulator -> global epilogue shared by the chain and GEMM kernels.
//
// The 32x32 MFMA C/D layout keeps one output COLUMN per lane, so a direct store writes 2-byte
// (bf16) or 4-byte pieces.  Streaming kernels live or die by their store efficiency, so the tile
// is transposed through a wave-private fp32 LDS scratch and written as 16-byte-per-lane row
// segments (full 128-byte lines for a 64-column bf16 / 32-column f32 tile).
#pragma once
#include "common.hpp"

namespace sow {

// scratch row stride in floats for NT side-by-side 32x32 tiles (+4 keeps rows 16-byte aligned and
// staggers banks between rows)
template <int NT> struct EpiScratch {
  static constexpr int W = NT * 32;
  static constexpr int LD = W + 4;
  static constexpr int FLOATS = 32 * LD;
};

// out[row0 + r][col0 + c] = alpha * acc + beta * out_old + bias[col]   (r < 32, c < NT*32)
// VEC: ld % VE == 0, col0 % VE == 0, pointers 16-byte aligned, N % VE == 0.
template <typename T, int NT, bool VEC>
__device__ __forceinline__ void wave_store_tiles(const f32x16* acc, float* scratch, T* out, int64_t ld, int64_t row0,
                                                 int col0, int64_t M, int N, float alpha, float beta,
                                                 const T* bias, int lane, bool nt = false) {
  constexpr int W = EpiScratch<NT>::W, LD = EpiScratch<NT>::LD, VE = DT<T>::VE;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) scratch[acc_row(reg, lane) * LD + nt * 32 + (lane & 31)] = acc[nt][reg];
  }
  __builtin_amdgcn_wave_barrier();
  if constexpr (VEC) {
    constexpr int VPR = W / VE;            // vectors per row
    constexpr int RPP = 64 / VPR;          // rows per pass
#pragma unroll
    for (int pass = 0; pass < 32 / RPP; ++pass) {
      const int r = pass * RPP + lane / VPR;
      const int c = (lane % VPR) * VE;
      const int64_t grow = row0 + r;
      const int gcol = col0 + c;
      if (grow < M && gcol < N) {
        float v[VE];
#pragma unroll
        for (int q = 0; q < VE / 4; ++q) {
          f32x4 t = *(const f32x4*)(scratch + r * LD + c + 4 * q);
          v[4 * q + 0] = t[0] * alpha;
          v[4 * q + 1] = t[1] * alpha;
          v[4 * q + 2] = t[2] * alpha;
          v[4 * q + 3] = t[3] * alpha;
        }
        T* dst = out + grow * ld + gcol;
        if (beta != 0.f) {
          u32x4 old = *(const u32x4*)dst;
          const T* o = (const T*)&old;
#pragma unroll
          for (int j = 0; j < VE; ++j) v[j] += beta * to_f32(o[j]);
        }
        if (bias) {
          u32x4 bv = *(const u32x4*)(bias + gcol);
          const T* b = (const T*)&bv;
#pragma unroll
          for (int j = 0; j < VE; ++j) v[j] += to_f32(b[j]);
        }
        u32x4 pk;
        T* pe = (T*)&pk;
#pragma unroll
        for (int j = 0; j < VE; ++j) pe[j] = from_f32<T>(v[j]);
        // nt: streaming (non-temporal) store for outputs the kernel never re-reads -- keeps the L2s clean, so the
        // write-back at the end of the kernel has nothing left to do (see chain2.hip)
        if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(pk) : "memory");
        else *(u32x4*)dst = pk;
      }
    }
  } else {
#pragma unroll 4
    for (int it = 0; it < 32 * W / 64; ++it) {
      const int idx = it * 64 + lane;
      const int r = idx / W, c = idx % W;
      const int64_t grow = row0 + r;
      const int gcol = col0 + c;
      if (grow < M && gcol < N) {
        float v = scratch[r * LD + c] * alpha;
        T* dst = out + grow * ld + gcol;
        if (beta != 0.f) v += beta * to_f32(*dst);
        if (bias) v += to_f32(bias[gcol]);
        *dst = from_f32<T>(v);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

}  // namespace sow
