// Probe: 16-byte global loads / LDS-DMA from 4-byte-aligned (not 16-byte-aligned) source addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void probe(const uint32_t* src, uint32_t* out_vec, uint32_t* out_dma) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[256];
  const int l = threadIdx.x;
  // lane l reads 16 bytes starting at dword (25*l + 1): 4-byte aligned, rows of 100 bytes
  const uint32_t* p = src + 25 * l + 1;
  u32x4 v = *(const u32x4*)p;
  for (int e = 0; e < 4; ++e) out_vec[l * 4 + e] = v[e];
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                   (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) out_dma[l * 4 + e] = buf[l * 4 + e];
}
int main() {
  uint32_t *d_src, *d_a, *d_b;
  hipMalloc(&d_src, 8192); hipMalloc(&d_a, 1024); hipMalloc(&d_b, 1024);
  std::vector<uint32_t> s(2048);
  for (int i = 0; i < 2048; ++i) s[i] = i;
  hipMemcpy(d_src, s.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, d_a, d_b);
  std::vector<uint32_t> a(256), b(256);
  hipMemcpy(a.data(), d_a, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d_b, 1024, hipMemcpyDeviceToHost);
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    if (a[l*4+e] != (uint32_t)(25*l+1+e)) ok1 = 0;
    if (b[l*4+e] != (uint32_t)(25*l+1+e)) ok2 = 0;
  }
  printf("MISALIGNED global_load_dwordx4: %s\nMISALIGNED global_load_lds_dwordx4: %s\n", ok1?"OK":"WRONG", ok2?"OK":"WRONG");
  if (!ok2) { for (int i = 0; i < 16; ++i) printf("%u ", b[i]); printf("\n"); }
  printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
