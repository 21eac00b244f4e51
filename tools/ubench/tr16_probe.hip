#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr_bytes, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr_bytes[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int h[64]; unsigned short o[256];
  int *d; unsigned short* dout;
  hipMalloc(&d, 256); hipMalloc(&dout, 512);
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h[l] = l * 8;                                  // linear: lane l -> elements 4l..4l+3
      if (pat == 1) h[l] = ((l & 15) / 4) * 200 + (l & 3) * 8 + (l >> 4) * 1000;   // row pitch 100 elements: rows p/4, col group p%4
      if (pat == 2) h[l] = (l & 15) * 96 + (l >> 4) * 8;          // each lane its own row (pitch 48 elem), col group g
    }
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d(el %4d): %4d %4d %4d %4d\n", l, h[l], h[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  return 0;
}
