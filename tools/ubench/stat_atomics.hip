// Micro-benchmark: what does a BN-statistics hand-over between two kernels cost?
//   chain A (what the engine does):  producer writes per-workgroup partial rows -> finalize kernel (one
//            workgroup per channel sums the rows) -> consumer reads scale/shift
//   chain B: producer adds its row to one of R replicated fp64 accumulator rows with memory-side atomics
//            (no fence) -> consumer sums the R rows in its prologue
// Both producers / consumers spin for `iters` FMAs per thread so that the launches look like the block kernels
// (1024 workgroups, all finishing at about the same time).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stat_atomics.hip -o /tmp/sa && /tmp/sa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int C2 = 96;   // 2 statistics x 48 channels

__device__ __forceinline__ float spin(int iters, float seed) {
  float c0 = seed, c1 = 1, c2 = 2, c3 = 3;
  for (int i = 0; i < iters; ++i) {
    c0 = fmaf(c0, 1.0001f, 0.5f); c1 = fmaf(c1, 1.0001f, 0.5f); c2 = fmaf(c2, 1.0001f, 0.5f); c3 = fmaf(c3, 1.0001f, 0.5f);
  }
  return c0 + c1 + c2 + c3;
}

__global__ __launch_bounds__(256) void producer_rows(float* part, int iters) {
  const float r = spin(iters, threadIdx.x);
  if (threadIdx.x < C2) part[(size_t)blockIdx.x * C2 + threadIdx.x] = r * 1e-30f + 1.0f;
}
__global__ __launch_bounds__(256) void finalize_rows(const float* part, int G, float* fin) {
  __shared__ double s[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double t = 0;
  for (int j = tid; j < G; j += 256) t += part[(size_t)j * C2 + c];
  s[tid] = t;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (tid < w) s[tid] += s[tid + w]; __syncthreads(); }
  if (tid == 0) fin[c] = (float)s[0];
}
__global__ __launch_bounds__(256) void consumer_fin(const float* fin, float* out, int iters) {
  __shared__ float sc[C2];
  if (threadIdx.x < C2) sc[threadIdx.x] = fin[threadIdx.x];
  __syncthreads();
  const float r = spin(iters, sc[threadIdx.x % C2]);
  if (r == 12345.678f) out[threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void producer_atomic(double* acc, int R, int iters) {
  const float r = spin(iters, threadIdx.x);
  if (threadIdx.x < C2) unsafeAtomicAdd(acc + (size_t)(blockIdx.x % R) * C2 + threadIdx.x, (double)(r * 1e-30f + 1.0f));
}
__global__ __launch_bounds__(256) void consumer_acc(const double* acc, int R, float* out, double* next_acc, int iters) {
  __shared__ float sc[C2];
  if (threadIdx.x < C2) {
    double t = 0;
    for (int j = 0; j < R; ++j) t += acc[(size_t)j * C2 + threadIdx.x];
    sc[threadIdx.x] = (float)t;
    if (blockIdx.x == 0) out[1024 + threadIdx.x] = (float)t;
  }
  // the accumulator rows of the *other* parity are cleared for the step after next
  if (blockIdx.x < R && threadIdx.x < C2) next_acc[(size_t)blockIdx.x * C2 + threadIdx.x] = 0.0;
  __syncthreads();
  const float r = spin(iters, sc[threadIdx.x % C2]);
  if (r == 12345.678f) out[threadIdx.x] = r;
}

int main() {
  const int G = 1024, reps = 200;
  float *part, *fin, *out;
  double* acc;
  hipMalloc(&part, sizeof(float) * G * C2);
  hipMalloc(&fin, sizeof(float) * C2);
  hipMalloc(&out, sizeof(float) * 4096);
  hipMalloc(&acc, sizeof(double) * 2 * 64 * C2);
  hipMemset(acc, 0, sizeof(double) * 2 * 64 * C2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int iters : {0, 2000, 8000}) {
    float ms;
    for (int pass = 0; pass < 2; ++pass) {
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(producer_rows, dim3(G), dim3(256), 0, 0, part, iters);
        hipLaunchKernelGGL(finalize_rows, dim3(C2), dim3(256), 0, 0, part, G, fin);
        hipLaunchKernelGGL(consumer_fin, dim3(G), dim3(256), 0, 0, fin, out, iters);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("iters %5d  rows+finalize            : %7.2f us per producer->consumer pair\n", iters, ms * 1e3f / reps);
    for (int pass = 0; pass < 2; ++pass) {
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(producer_rows, dim3(G), dim3(256), 0, 0, part, iters);
        hipLaunchKernelGGL(consumer_fin, dim3(G), dim3(256), 0, 0, fin, out, iters);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("iters %5d  rows, finalize skipped    : %7.2f us per pair\n", iters, ms * 1e3f / reps);
    for (int which = 0; which < 4; ++which) {
      for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) {
          if (which == 0) hipLaunchKernelGGL(producer_rows, dim3(G), dim3(256), 0, 0, part, iters);
          if (which == 1) hipLaunchKernelGGL(consumer_fin, dim3(G), dim3(256), 0, 0, fin, out, iters);
          if (which == 2) hipLaunchKernelGGL(producer_atomic, dim3(G), dim3(256), 0, 0, acc, 16, iters);
          if (which == 3) hipLaunchKernelGGL(consumer_acc, dim3(G), dim3(256), 0, 0, acc, 16, out, acc + 64 * C2, iters);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("iters %5d  kernel %d alone           : %7.2f us\n", iters, which, ms * 1e3f / reps);
    }
    for (int R : {1, 4, 16, 64}) {
      for (int pass = 0; pass < 2; ++pass) {
        hipMemset(acc, 0, sizeof(double) * 2 * 64 * C2);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) {
          double* cur = acc + (size_t)(i & 1) * 64 * C2;
          double* nxt = acc + (size_t)((i + 1) & 1) * 64 * C2;
          hipLaunchKernelGGL(producer_atomic, dim3(G), dim3(256), 0, 0, cur, R, iters);
          hipLaunchKernelGGL(consumer_acc, dim3(G), dim3(256), 0, 0, cur, R, out, nxt, iters);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<float> h(C2);
      hipMemcpy(h.data(), out + 1024, sizeof(float) * C2, hipMemcpyDeviceToHost);
      printf("iters %5d  fp64 atomics, %2d acc rows : %7.2f us per pair   (check: total %.1f, expect %d)\n", iters, R, ms * 1e3f / reps, h[5], G);
    }
  }
  return 0;
}
