#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr_bytes, unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr_bytes[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_addr[l] = l * 8;                                  // lane l -> elements 4l..4l+3
      if (pat == 1) h_addr[l] = (l & 15) * 32 + (l >> 4) * 8;           // row = l&15 (32-B rows), 8-B column = l>>4
      if (pat == 2) h_addr[l] = ((l & 3) * 4 + ((l >> 2) & 3)) * 32 + (l >> 4) * 8 + 1024;
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
  }
  return 0;
}
