// Micro-benchmark: fp32 x fp32 products on the bf16 matrix pipe with COMPLETE split products.
//   x = x1 + x2 + x3 exactly (three bf16 values: the three 8-bit slices of the fp32 mantissa, by truncation), likewise y;
//   x * y = sum of all nine cross products xi * yj, each exact in fp32, accumulated in fp32 by v_mfma_f32_16x16x32_bf16.
// Part 1 (accuracy): C = A[16 x 128] * B[128 x 16] from random fp32 data, three ways - v_mfma_f32_16x16x4_f32 (32 k-steps),
//   the nine-term split (4 k-steps x 9), a six-term split (terms with i + j <= 4) - against a float64 reference.
// Part 2 (rate): 256-thread blocks, four per CU (one wave of each block on every SIMD, as fwd_first_kernel runs):
//   mode 0: per iteration 2 accumulators x 8 f32 MFMAs (K = 32 of exact-fp32 work)
//   mode 1: per iteration 2 accumulators x 9 bf16 16x16x32 MFMAs (the same K = 32) on one accumulator each
//   mode 2: the same with three accumulators per row tile (hi*hi | the four middle terms | the four small ones)
//   mode 3: mode 1 plus the split of one A fragment per iteration in VALU (8 elements: and / sub / and / sub / pack)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16x9.hip -o tools/ubench/mfma_bf16x9 && tools/ubench/mfma_bf16x9
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// three bf16 slices of eight fp32 values, packed for the MFMA operand (element i of the lane = k index g*8 + i)
struct Split3 { bf16x8 p[3]; };
__device__ __forceinline__ Split3 split8(const float (&v)[8]) {
  unsigned h[3][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned b0 = __float_as_uint(v[i]) & 0xffff0000u;
    const float r1 = v[i] - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b1);
    h[0][i] = b0; h[1][i] = b1; h[2][i] = __float_as_uint(r2) & 0xffff0000u;
  }
  Split3 s;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (h[p][2 * i] >> 16) | h[p][2 * i + 1];
    s.p[p] = __builtin_bit_cast(bf16x8, w);
  }
  return s;
}

__global__ void accuracy(const float* A, const float* B, float* Cf32, float* C9, float* C6, int K) {
  const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
  f32x4 cf = {0, 0, 0, 0}, c9 = cf, c6 = cf;
  for (int k = 0; k < K; k += 4) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r16 * K + k + g], B[(k + g) * 16 + r16], cf, 0, 0, 0);
  for (int k = 0; k < K; k += 32) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[i] = A[r16 * K + k + g * 8 + i];
      b[i] = B[(k + g * 8 + i) * 16 + r16];
    }
    const Split3 sa = split8(a), sb = split8(b);
    // small terms first
#pragma unroll
    for (int s = 4; s >= 0; --s)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int j = s - i;
        if (j < 0 || j > 2) continue;
        c9 = mf(sa.p[i], sb.p[j], c9);
        if (s <= 2) c6 = mf(sa.p[i], sb.p[j], c6);
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Cf32[(g * 4 + r) * 16 + r16] = cf[r];
    C9[(g * 4 + r) * 16 + r16] = c9[r];
    C6[(g * 4 + r) * 16 + r16] = c6[r];
  }
}

template <int mode>
__global__ __launch_bounds__(256, 4) void rate(float* out, int iters) {
  float r = 0.f;
  if (mode == 0) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1];
  } else {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    Split3 sa = split8(v), sb = sa;
    f32x4 c[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 3; ++q) c[t][q] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      if (mode == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * 1.0001f;
        sa = split8(v);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int q = mode == 2 ? (i + j == 0 ? 0 : (i + j <= 2 ? 1 : 2)) : 0;
          c[0][q] = mf(sa.p[i], sb.p[j], c[0][q]);
          c[1][q] = mf(sb.p[j], sa.p[i], c[1][q]);
        }
    }
    r = c[0][0][0] + c[0][1][1] + c[0][2][2] + c[1][0][0] + c[1][1][1] + c[1][2][2];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  const int K = 128;
  std::vector<float> A(16 * K), B(K * 16);
  srand(7);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 26.f;                       // spectrogram-like magnitudes
  for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;              // weight-like
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 3 * 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, dC, dC + 256, dC + 512, K);
  std::vector<float> C(3 * 256);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double e[3] = {0, 0, 0}, scale = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { ref += (double)A[m * K + k] * B[k * 16 + n]; mag += fabs((double)A[m * K + k] * B[k * 16 + n]); }
      scale = fmax(scale, mag);
      for (int v = 0; v < 3; ++v) e[v] = fmax(e[v], fabs(C[v * 256 + m * 16 + n] - ref) / mag);
    }
  printf("accuracy, max |C - C64| / sum|a*b| over a 16x16 tile, K = %d:  f32 MFMA %.3e   nine-term bf16 %.3e   six-term bf16 %.3e\n", K, e[0], e[1], e[2]);

  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 4, iters = 20000;
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"2 x 8 f32 16x16x4 MFMA (K = 32 of exact fp32)", "2 x 9 bf16 16x16x32 MFMA, one accumulator per row tile",
                         "2 x 9 bf16 16x16x32 MFMA, three accumulators per row tile", "mode 1 + one 8-element operand split per iteration"};
  for (int mode = 0; mode < 4; ++mode) {
    auto launch = [&](int n) {
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, out, n);
      if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, out, n);
      if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 0, 0, out, n);
      if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(256), 0, 0, out, n);
    };
    launch(100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // four waves per SIMD, each running `iters` iterations: SIMD cycles per iteration of ONE wave's work = ms * clk / (4 * iters)
    printf("mode %d: %.3f ms  -> %.1f SIMD cycles per wave-iteration at a nominal 2.4 GHz   (%s)\n", mode, ms, ms * 1e-3 * 2.4e9 / (4.0 * iters), names[mode]);
  }
  return 0;
}
