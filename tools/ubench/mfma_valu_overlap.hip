// Micro-benchmark: does v_mfma_f32_16x16x4_f32 (f32-in MFMA) overlap with f32 VALU work issued by
// another wave of the same SIMD?  512-thread blocks, one per CU: waves 0-3 and 4-7 share SIMDs 0-3.
//   mode 0: waves 0-3 run MFMAs, waves 4-7 idle      mode 1: waves 4-7 run v_fma, waves 0-3 idle
//   mode 2: both                                      mode 3: waves 4-7 run v_pk_fma only
//   mode 4: MFMA + pk_fma                             mode 5: all 8 waves MFMA    mode 6: all 8 waves v_fma
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool first_half = wave < 4;
  float r = 0.f;
  const bool do_mfma = (mode == 0 || mode == 2 || mode == 4) ? first_half : (mode == 5);
  const bool do_fma = (mode == 1 || mode == 2) ? !first_half : (mode == 6);
  const bool do_pk = (mode == 3 || mode == 4) ? !first_half : false;
  if (do_mfma) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (do_fma) {
    float c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    const float m = 1.0001f, b = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // 64 v_fma per iteration
        c0 = fmaf(c0, m, b); c1 = fmaf(c1, m, b); c2 = fmaf(c2, m, b); c3 = fmaf(c3, m, b);
        c4 = fmaf(c4, m, b); c5 = fmaf(c5, m, b); c6 = fmaf(c6, m, b); c7 = fmaf(c7, m, b);
      }
    }
    r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  } else if (do_pk) {
    f32x2 c0 = {(float)threadIdx.x, 1}, c1 = {2, 3}, c2 = {4, 5}, c3 = {6, 7}, c4 = {8, 9}, c5 = {1, 2}, c6 = {3, 4}, c7 = {5, 6};
    const f32x2 m = {1.0001f, 1.0002f}, b = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // 64 v_pk_fma per iteration
        c0 = __builtin_elementwise_fma(c0, m, b); c1 = __builtin_elementwise_fma(c1, m, b);
        c2 = __builtin_elementwise_fma(c2, m, b); c3 = __builtin_elementwise_fma(c3, m, b);
        c4 = __builtin_elementwise_fma(c4, m, b); c5 = __builtin_elementwise_fma(c5, m, b);
        c6 = __builtin_elementwise_fma(c6, m, b); c7 = __builtin_elementwise_fma(c7, m, b);
      }
    }
    r = c0[0] + c1[1] + c2[0] + c3[1] + c4[0] + c5[1] + c6[0] + c7[1];
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int mode = 0; mode <= 6; ++mode) {
    k<<<256, 512>>>(d, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<<<256, 512>>>(d, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    // per SIMD: mfma = 4*iters instr (half waves) ; fma = 64*iters
    printf("mode %d: %.3f ms  (cycles/iter per wave at 2.1GHz: %.1f)\n", mode, ms, ms * 1e-3 * 2.1e9 / iters);
  }
  return 0;
}
