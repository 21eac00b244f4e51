// Micro-benchmark: issue rate of v_mfma_f32_16x16x16_bf16 / v_mfma_f32_32x32x8_bf16 and whether the bf16 matrix pipe
// overlaps with f32 VALU work (v_fma_f32, and the conversion mix of a split-precision GEMM) issued by another wave of
// the same SIMD.  512-thread blocks, one per CU: waves 0-3 and 4-7 share SIMDs 0-3.
//   mode 0: waves 0-3 bf16 16x16x16 MFMA only       mode 1: waves 4-7 v_fma only
//   mode 2: both                                     mode 3: all 8 waves bf16 MFMA
//   mode 4: one wave interleaves 6 bf16 MFMA with 12 v_fma (same wave, independent)   mode 5: f32 16x16x4 MFMA only (waves 0-3)
//   mode 6: f32 MFMA (waves 0-3) + bf16 MFMA (waves 4-7)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_overlap.hip -o /tmp/ub2 && /tmp/ub2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool first_half = wave < 4;
  float r = 0.f;
  const bool do_bf = (mode == 0 || mode == 2) ? first_half : (mode == 3 || (mode == 6 && !first_half));
  const bool do_fma = (mode == 1 || mode == 2) ? !first_half : false;
  const bool do_mix = mode == 4 && first_half;
  const bool do_f32 = (mode == 5 || mode == 6) && first_half;
  if (do_bf) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    s16x4 x = {(short)threadIdx.x, 1, 2, 3}, y = {4, 5, 6, (short)threadIdx.x};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (do_f32) {
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
  } else if (do_mix) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
    s16x4 x = {(short)threadIdx.x, 1, 2, 3}, y = {4, 5, 6, (short)threadIdx.x};
    float c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3;
    const float m = 1.0001f, b = 0.5f;
    for (int i = 0; i < iters; ++i) {   // 6 bf16 MFMA + 12 v_fma per iteration, independent of each other
      a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a0, 0, 0, 0);
      c0 = fmaf(c0, m, b); c1 = fmaf(c1, m, b);
      a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a1, 0, 0, 0);
      c2 = fmaf(c2, m, b); c3 = fmaf(c3, m, b);
      a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a2, 0, 0, 0);
      c0 = fmaf(c0, m, b); c1 = fmaf(c1, m, b);
      a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a3, 0, 0, 0);
      c2 = fmaf(c2, m, b); c3 = fmaf(c3, m, b);
      a4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a4, 0, 0, 0);
      c0 = fmaf(c0, m, b); c1 = fmaf(c1, m, b);
      a5 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a5, 0, 0, 0);
      c2 = fmaf(c2, m, b); c3 = fmaf(c3, m, b);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + c0 + c1 + c2 + c3;
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
  const char* what[] = {"bf16 16x16x16 MFMA x4/iter, waves 0-3", "v_fma x64/iter, waves 4-7", "both", "bf16 MFMA x4/iter, all 8 waves",
                        "6 bf16 MFMA + 12 v_fma interleaved in one wave (waves 0-3)", "f32 16x16x4 MFMA x4/iter, waves 0-3", "f32 MFMA (0-3) + bf16 MFMA (4-7)"};
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
    printf("mode %d: %.3f ms  cycles/iter at 2.4 GHz: %.1f   (%s)\n", mode, ms, ms * 1e-3 * 2.4e9 / iters, what[mode]);
  }
  return 0;
}
