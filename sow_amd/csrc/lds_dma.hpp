// LDS-DMA / inline-asm LDS read helpers shared by the streaming kernels (chain2.hip, gemm2.hip).
//
// Why inline asm: while an LDS-DMA (`global_load_lds_dwordx4`) is outstanding hipcc guards every
// compiler-visible LDS read with `s_waitcnt vmcnt(0)`, which would drain the rings at every step.
// Reads issued through these macros are invisible to that logic; the kernels order them by hand with
// counted `s_waitcnt vmcnt(N)`, raw `s_barrier` and `s_waitcnt lgkmcnt(0)`.
#pragma once
#include "common.hpp"

namespace sow {

// Source of out-of-range DMA lanes: exact zeros (one copy per translation unit).  LDS-DMA has no per-lane
// predication that still writes the destination, so a lane that must contribute zeros reads from here.
static __device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];
__device__ __forceinline__ const char* zero_page_for(int lane) { return (const char*)(g_zero_page + (lane & 7) * 4); }

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
#define DS_READ_B128(dst, addr, off) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DS_READ_B64(dst, addr, off) \
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DS_READ_B32(dst, addr, off) \
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
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
// the same with the non-temporal cache policy (aux bit 1 = nt on gfx94x/gfx950): streamed-once operands
__device__ __forceinline__ void dma16_nt(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 join2(u32x2 lo, u32x2 hi) { return (u32x4){lo[0], lo[1], hi[0], hi[1]}; }
// ---- fp32 on the bf16 matrix pipe ("3 x bf16"): x = hi + mid + lo with three bf16 values that hold the 24 mantissa bits
// EXACTLY (truncation split: the residuals x - hi and x - hi - mid are exact in fp32), and
//     a * b  ~=  ah bh + ah bm + am bh + am bm + ah bl + al bh          (fp32 accumulation in the MFMA)
// drops only am bl + al bm + al bl <= 2^-23 |a b| -- below the rounding of one fp32 multiply-add.  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each, 16 k) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each, 2 k): 2.7 x less
// matrix-pipe time per product, and -- unlike the fp32 MFMA, which blocks every other instruction of its SIMD while
// it runs (tools/probe4.hip) -- the bf16 MFMA lets LDS reads, VALU and DMA issue of the partner wave pass.
// The price is the split: 11 VALU instructions per pair of operand elements.
// two fp32 subtractions in one instruction, on a 64-bit register pair.  Integer-typed on purpose: hipcc (ROCm 7.2) folds
// element 1 of a float vector into element 0 when the element is bit-cast (seen in the ISA of an f32x2 version of split3:
// both halves of the mid plane came from element 0 -- the lo-plane accuracy was silently lost, caught by the 2e-6 loss bar of
// the config-4 protocol trace).
__device__ __forceinline__ uint64_t pk_sub_f32(uint64_t a, uint64_t b) {
  uint64_t r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  // 9 VALU instructions per pair: 2 masks + 1 packed subtract per level, one v_perm_b32 per plane to pack the high halves
  constexpr uint32_t HI = 0x07060302u;   // v_perm_b32(b, a, HI) = (a >> 16) | (b & 0xffff0000)
  constexpr uint32_t MK = 0xffff0000u;
  const uint32_t u0 = __builtin_bit_cast(uint32_t, x0), u1 = __builtin_bit_cast(uint32_t, x1);
  h = __builtin_amdgcn_perm(u1, u0, HI);
  const uint64_t x64 = (uint64_t)u0 | ((uint64_t)u1 << 32);
  const uint64_t r64 = pk_sub_f32(x64, (uint64_t)(u0 & MK) | ((uint64_t)(u1 & MK) << 32));          // exact residuals
  const uint32_t r0 = (uint32_t)r64, r1 = (uint32_t)(r64 >> 32);
  m = __builtin_amdgcn_perm(r1, r0, HI);
  const uint64_t q64 = pk_sub_f32(r64, (uint64_t)(r0 & MK) | ((uint64_t)(r1 & MK) << 32));           // exact, <= 8 bits
  l = __builtin_amdgcn_perm((uint32_t)(q64 >> 32), (uint32_t)q64, HI);
}
// element pair e (bf16 elements 2e, 2e + 1) of the three plane vectors of a fragment
__device__ __forceinline__ void split3v(float x0, float x1, u32x4 (&pl)[3], int e) {
  uint32_t h, m, l;
  split3(x0, x1, h, m, l);
  pl[0][e] = h, pl[1][e] = m, pl[2][e] = l;
}
// acc += a * b for fragments split into (hi, mid, lo) planes
__device__ __forceinline__ f32x16 mfma_x3(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
  acc = mfma32(as_bf16x8(a[2]), as_bf16x8(b[0]), acc);   // small terms first
  acc = mfma32(as_bf16x8(a[0]), as_bf16x8(b[2]), acc);
  acc = mfma32(as_bf16x8(a[1]), as_bf16x8(b[1]), acc);
  acc = mfma32(as_bf16x8(a[1]), as_bf16x8(b[0]), acc);
  acc = mfma32(as_bf16x8(a[0]), as_bf16x8(b[1]), acc);
  acc = mfma32(as_bf16x8(a[0]), as_bf16x8(b[0]), acc);
  return acc;
}

}  // namespace sow
