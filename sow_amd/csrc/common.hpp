// Shared device/host helpers for the sow_amd HIP library (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, workgroups of 256 threads (4 waves) unless stated;
//   * MFMA shapes: v_mfma_f32_32x32x16_bf16 (bf16 in, f32 acc) and
//     v_mfma_f32_32x32x2_f32 (exact f32).  Both share the C/D map
//       col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), reg in [0,16);
//     A operand: lane holds A[row = lane & 31][k-group lane >> 5]; B operand: B[k-group][col = lane & 31];
//     bf16: 8 consecutive k per lane (k = 8 * (lane >> 5) + j), f32: one k per lane (k = lane >> 5);
//   * LDS tiles for bf16 operands are "k-contiguous" images Img[row][k] with a 16-byte chunk XOR
//     swizzle so that ds_read_b128 fragment reads are bank-conflict free;
//   * no function here allocates, frees or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "../../include/sow_amd.h"

namespace sow {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

enum : int { SOW_F32 = 0, SOW_BF16 = 1 };


template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int id = SOW_F32;
  static constexpr int VE = 4;  // elements per 16-byte vector
};
template <> struct DT<bf16_t> {
  static constexpr int id = SOW_BF16;
  static constexpr int VE = 8;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN-preserving

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of accumulator register `reg` for this lane inside a 32x32 C/D tile
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------
// bf16 k-contiguous LDS image: rows of KT elements (KT = 32, 64 or 128), 16-byte chunks XOR-swizzled.
// chunk index c in [0, KT/8); physical chunk = c ^ swz(row).  The swizzle spreads the 16 rows a
// ds_read_b128 lane group touches over all 16 slots of the 256-byte bank row.
// ---------------------------------------------------------------------------------------------
template <int KT> __device__ __forceinline__ int bf16_img_chunk(int row, int c) {
  constexpr int NC = KT / 8;        // chunks per row
  constexpr int RPB = 16 / NC;      // rows per 256-byte bank row (NC = 4 -> 4, 8 -> 2, 16 -> 1)
  static_assert(NC == 4 || NC == 8 || NC == 16, "KT must be 32, 64 or 128");
  return c ^ ((row / RPB) & (NC - 1));
}
// byte offset of chunk c of `row`
template <int KT> __device__ __forceinline__ int bf16_img_off(int row, int c) {
  return row * (KT * 2) + bf16_img_chunk<KT>(row, c) * 16;
}

// 8x(2 columns) dword block -> two 8-element k-vectors (register transpose used when the stored
// matrix has the contraction index as its ROW index).  d[j] holds (col0, col1) of k-row j.
__device__ __forceinline__ void transpose_8x2(const uint32_t (&d)[8], u32x4& col0, u32x4& col1) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    uint32_t a = d[2 * m], b = d[2 * m + 1];
    col0[m] = (a & 0xffffu) | (b << 16);
    col1[m] = (a >> 16) | (b & 0xffff0000u);
  }
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// XCD-aware bijective block remap (cdna guide T1): consecutive logical ids land on the same XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device; thread-safe: forward runs on the
// caller's thread, backward on the autograd thread -- two threads racing here both set the same value).  The kernel
// expression goes last because template-ids contain commas.
#define SOW_SET_MAX_LDS_ONCE(bytes, ...)                                                                            \
  do {                                                                                                              \
    static std::atomic<uint64_t> done__{0};                                                                         \
    int dev__ = 0;                                                                                                  \
    (void)hipGetDevice(&dev__);                                                                                     \
    const uint64_t bit__ = 1ull << (dev__ & 63);                                                                    \
    if (!(done__.load(std::memory_order_acquire) & bit__)) {                                                        \
      (void)hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));    \
      done__.fetch_or(bit__, std::memory_order_release);                                                            \
    }                                                                                                               \
  } while (0)

// Kernel-selection switches (A/B measurements and the tests that pin every kernel variant).  Read from the
// environment ONCE, when the library is first used (SOW_AMD_<NAME>), and changed afterwards only through
// sow_set_switch() (include/sow_amd.h); launches read an atomic, never getenv.  -1 = unset (automatic choice).
enum Switch : int {
  SW_FORCE_CHAIN_V1 = 0,  // generic chain kernels instead of the streaming ones
  SW_NO_SHORT_SPLIT,      // no K / column split of the chain for short inputs
  SW_NO_FUSED_H,          // dense-accumulator layer as H-only chain + K-extended GEMM (no gemm2h)
  SW_FORCE_GEMM_V1,       // generic GEMM kernel everywhere
  SW_TN_NARROW,           // one column group per skinny-TN workgroup
  SW_NO_GEMM3S,           // never the 128x128-tile streaming GEMM
  SW_GEMM3S,              // 1 / 0: force / forbid gemm3s
  SW_GEMM3,               // 1 / 0: force / forbid gemm3
  SW_NO_GROUPED,          // grouped (multi-layer) entry points launch layer by layer
  SW_NO_PERSIST,          // grouped grids launch one workgroup per token block instead of resident workgroups that loop
  SW_NO_NT_STORE,         // plain (cached) Y stores in the chain kernel instead of non-temporal ones
  SW_NT_LOAD,             // experiment: non-temporal X loads in the chain kernel
  SW_NO_PAIR_FLUSH,       // chain kernel stores one 64-column slice per flush (128-byte pieces) instead of two
  SW_F32_EXACT,           // fp32 kernels on the exact v_mfma_f32_32x32x2_f32 instead of the 3 x bf16 split
  SW_NO_PARK16,           // chain kernel parks output slices as fp32 even when bf16 would be exact
  SW_TN_NO_NT_LOAD,       // row-owner weight-gradient kernel streams x / dY with plain (cached) loads
  SW_NO_TN_ROWS,          // grouped weight-gradient launches never use the row-owner kernel (group-planned slabs)
  SW_GEMM4,               // 1 / 0: force / forbid gemm4 (anti-phase wave groups, 64-wide K-tiles)
  SW_NO_GEMM4H,           // dense-accumulator layer with the projection inside the kernel: gemm2h instead of gemm4h
  SW_NO_CHAIN3F,          // fp32 chain: chain2f (per-use factor split, 64-token workgroups) instead of chain3f
  SW_NO_TN_F32Q,          // fp32 weight gradients: the wide (two column groups, private S stream) kernel instead of the quad one
  SW_NO_SPLITK,           // bf16 GEMM with few output tiles (short T): never split K over workgroups
  SW_COUNT
};
int sw(int which);
inline bool sw_on(int which) { return sw(which) > 0; }

// streaming-store policy of the GEMM epilogues: outputs of tall products (activations, activation gradients) are not
// re-read by the kernel that writes them
#ifdef SOW_GEMM_NO_NT
#define SOW_GEMM_NT(M) false
#else
#define SOW_GEMM_NT(M) ((M) >= 8192)
#endif

#define SOW_CHECK_LAUNCH()                     \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

}  // namespace sow
