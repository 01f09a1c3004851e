// Hardware probes for gfx950 instruction semantics used by the sow_amd kernels:
//   1. ds_read_b64_tr_b16 (transposed LDS read) lane/element mapping
//   2. global_load_lds dwordx4 (LDS-DMA) destination layout
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe.hip -o tools/probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void probe_tr(uint16_t* out) {
  // LDS tile [16 rows][64 cols] of u16, value = row*256 + col ; row stride 128 B
  __shared__ __attribute__((aligned(16))) uint16_t tile[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) tile[i] = (uint16_t)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, j = l & 15, q = j >> 2, p = j & 3;
  // group g reads the 4x16 block at rows 4*g..4*g+3, cols 16*g .. 16*g+15 (distinct per group)
  const uint16_t* addr = &tile[(4 * g + q) * 64 + 16 * g + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (uint16_t)v[e];
}

__global__ void probe_glds(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[2 * 256];  // 2 KiB: two wave-instructions
  for (int i = threadIdx.x; i < 512; i += 64) buf[i] = 0xdeadbeefu;
  __syncthreads();
  const int l = threadIdx.x;
  // lane l loads 16 bytes from a PERMUTED source position (l ^ 5), dest base = buf (wave uniform)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (l ^ 5)),
                                   (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
  // second instruction into buf + 1 KiB, identity source order, offset immediate 0
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 256 + 4 * l),
                                   (__attribute__((address_space(3))) void*)(buf + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}

int main() {
  uint16_t* d_out;
  hipMalloc(&d_out, 64 * 4 * 2);
  hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d_out);
  std::vector<uint16_t> h(256);
  hipMemcpy(h.data(), d_out, 512, hipMemcpyDeviceToHost);
  printf("ds_read_tr16_b64: lane -> 4 elements as (row,col)\n");
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      printf(" (%d,%d)", h[l * 4 + e] >> 8, h[l * 4 + e] & 255);
      const int g = l >> 4, i = l & 15;
      if ((h[l * 4 + e] >> 8) != 4 * g + e || (h[l * 4 + e] & 255) != 16 * g + i) ok = 0;
    }
    printf("\n");
  }
  printf("TR_EXPECTED_MAPPING %s  (lane i of group g gets rows 4g..4g+3 of column 16g+i)\n", ok ? "YES" : "NO");

  uint32_t *d_src, *d_o2;
  hipMalloc(&d_src, 2048);
  hipMalloc(&d_o2, 2048);
  std::vector<uint32_t> s(512);
  for (int i = 0; i < 512; ++i) s[i] = i;
  hipMemcpy(d_src, s.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, 0, d_src, d_o2);
  std::vector<uint32_t> o(512);
  hipMemcpy(o.data(), d_o2, 2048, hipMemcpyDeviceToHost);
  int ok2 = 1;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      if (o[l * 4 + e] != (uint32_t)(4 * (l ^ 5) + e)) ok2 = 0;
      if (o[256 + l * 4 + e] != (uint32_t)(256 + 4 * l + e)) ok2 = 0;
    }
  printf("GLDS lane-linear dest with per-lane source: %s\n", ok2 ? "YES" : "NO");
  if (!ok2) {
    for (int i = 0; i < 32; ++i) printf("%u ", o[i]);
    printf("\n");
    for (int i = 256; i < 288; ++i) printf("%u ", o[i]);
    printf("\n");
  }
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  return 0;
}
