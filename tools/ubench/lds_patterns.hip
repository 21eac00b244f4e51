// Micro-benchmark: the LDS access patterns of the block kernels, one kernel per pattern, for the SQ LDS counters
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel) - calibrates tools/lds_banks.py against the hardware.
// Every kernel issues ITERS x 8 copies of ONE DS instruction (inline asm: the compiler cannot merge or split it), 256
// threads per workgroup, 256 workgroups.  The pattern is the template argument:
//   window<C, P, L>      ds_read_b32   (channel, chunk) register-window read: lane -> (c = tid % C, chunk = tid / C), row chunk*L
//   window2<C, P, L>     ds_read2_b32  the same with the next row in the second slot (what the compiler emits for windows)
//   uwrite<C, P, L>      ds_write_b32  the (channel, chunk) write of the u tile
//   uwrite2<C, P, L>     ds_write2_b32 two rows per instruction
//   cols<P>              ds_read_b32   MFMA column-pattern operand: lane (r16, g) reads row r16, column g
//   cols2<P>             ds_read2_b32  the same with column g + 4 in the second slot (two k-steps per instruction)
//   cols128<P>           ds_read_b128  the same operand as one float4 per lane (columns 4 g ..)
//   rows<P, D>           ds_read_b32   MFMA row-pattern operand: lane (r16, g) reads row D*g, column r16
//   table4               4 x ds_read_b32 of a 48-float table at 4 (tid % 12) + e   (the BN scale / shift reads of a commit)
//   commit128<C, P>      ds_write_b128 float4 i of a [rows][C] tile at (i / (C/4)) * P + 4 (i % (C/4))
// build + run:  hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_patterns.hip -o /tmp/lds_patterns
//               rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- /tmp/lds_patterns
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 64;
__shared__ float s_lds[16384];

#define REP8(x) x x x x x x x x

__device__ __forceinline__ unsigned lds_base() { return (unsigned)(size_t)s_lds; }

template <int C, int P, int L>
__global__ __launch_bounds__(256) void window(float* out) {
  const int tid = threadIdx.x, c = tid % C, ch = tid / C;
  const bool on = ch < 256 / C;
  const unsigned a = lds_base() + 4u * (unsigned)((ch * L) * P + c);
  float v = 0.f, acc = 0.f;
  if (on)
    for (int i = 0; i < ITERS; ++i) {
      REP8(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));)
      acc += v;
    }
  if (acc == 12345.f) out[tid] = acc;
}

template <int C, int P, int L>
__global__ __launch_bounds__(256) void window2(float* out) {
  const int tid = threadIdx.x, c = tid % C, ch = tid / C;
  const bool on = ch < 256 / C;
  const unsigned a = lds_base() + 4u * (unsigned)((ch * L) * P + c);
  f32x2 v = {0.f, 0.f};
  float acc = 0.f;
  if (on)
    for (int i = 0; i < ITERS; ++i) {
      REP8(asm volatile("ds_read2_b32 %0, %1 offset1:%2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(P));)
      acc += v.x;
    }
  if (acc == 12345.f) out[tid] = acc;
}

template <int C, int P, int L>
__global__ __launch_bounds__(256) void uwrite(float* out) {
  const int tid = threadIdx.x, c = tid % C, ch = tid / C;
  const bool on = ch < 256 / C;
  const unsigned a = lds_base() + 4u * (unsigned)((ch * L) * P + c);
  const float v = (float)tid;
  if (on)
    for (int i = 0; i < ITERS; ++i) {
      REP8(asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v));)
    }
  if (v == 12345.f) out[tid] = s_lds[tid];
}

template <int C, int P, int L>
__global__ __launch_bounds__(256) void uwrite2(float* out) {
  const int tid = threadIdx.x, c = tid % C, ch = tid / C;
  const bool on = ch < 256 / C;
  const unsigned a = lds_base() + 4u * (unsigned)((ch * L) * P + c);
  const float v = (float)tid;
  if (on)
    for (int i = 0; i < ITERS; ++i) {
      REP8(asm volatile("ds_write2_b32 %0, %1, %1 offset1:%2\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v), "n"(P));)
    }
  if (v == 12345.f) out[tid] = s_lds[tid];
}

template <int P>
__global__ __launch_bounds__(256) void cols(float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const unsigned a = lds_base() + 4u * (unsigned)((wave * 16 + r16) * P + g);
  float v = 0.f, acc = 0.f;
  for (int i = 0; i < ITERS; ++i) {
    REP8(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));)
    acc += v;
  }
  if (acc == 12345.f) out[tid] = acc;
}

template <int P>
__global__ __launch_bounds__(256) void cols2(float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const unsigned a = lds_base() + 4u * (unsigned)((wave * 16 + r16) * P + g);
  f32x2 v = {0.f, 0.f};
  float acc = 0.f;
  for (int i = 0; i < ITERS; ++i) {
    REP8(asm volatile("ds_read2_b32 %0, %1 offset1:4\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));)
    acc += v.x;
  }
  if (acc == 12345.f) out[tid] = acc;
}

template <int P>
__global__ __launch_bounds__(256) void cols128(float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const unsigned a = lds_base() + 4u * (unsigned)((wave * 16 + r16) * P + 4 * g);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  float acc = 0.f;
  for (int i = 0; i < ITERS; ++i) {
    REP8(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));)
    acc += v.x;
  }
  if (acc == 12345.f) out[tid] = acc;
}

template <int P, int D>
__global__ __launch_bounds__(256) void rows(float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const unsigned a = lds_base() + 4u * (unsigned)((wave * 16 + D * g) * P + r16);
  float v = 0.f, acc = 0.f;
  for (int i = 0; i < ITERS; ++i) {
    REP8(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));)
    acc += v;
  }
  if (acc == 12345.f) out[tid] = acc;
}

__global__ __launch_bounds__(256) void table4(float* out) {
  const int tid = threadIdx.x;
  const unsigned a = lds_base() + 16u * (unsigned)(tid % 12);
  float v = 0.f, acc = 0.f;
  for (int i = 0; i < ITERS; ++i) {
    REP8(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));)
    REP8(asm volatile("ds_read_b32 %0, %1 offset:4\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));)
    acc += v;
  }
  if (acc == 12345.f) out[tid] = acc;
}

template <int C, int P>
__global__ __launch_bounds__(256) void commit128(float* out) {
  const int tid = threadIdx.x, Q = C / 4;
  const unsigned a = lds_base() + 4u * (unsigned)((tid / Q) * P + 4 * (tid % Q));
  const f32x4 v = {(float)tid, 1.f, 2.f, 3.f};
  for (int i = 0; i < ITERS; ++i) {
    REP8(asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v));)
  }
  if (v.x == 12345.f) out[tid] = s_lds[tid];
}

#define RUN(k)                                                  \
  hipLaunchKernelGGL((k), dim3(256), dim3(256), 0, 0, d);      \
  hipDeviceSynchronize();

int main() {
  float* d;
  hipMalloc(&d, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    RUN((window<48, 52, 13>)) RUN((window<48, 48, 13>)) RUN((window2<48, 52, 13>)) RUN((window2<48, 48, 13>))
    RUN((window<32, 36, 8>)) RUN((window2<32, 36, 8>))
    RUN((uwrite<48, 52, 13>)) RUN((uwrite<48, 50, 13>)) RUN((uwrite<48, 48, 13>)) RUN((uwrite2<48, 52, 13>)) RUN((uwrite2<48, 50, 13>))
    RUN((cols<52>)) RUN((cols<50>)) RUN((cols<48>)) RUN((cols<36>)) RUN((cols<34>))
    RUN((cols2<52>)) RUN((cols2<50>)) RUN((cols2<34>))
    RUN((cols128<52>)) RUN((cols128<48>)) RUN((cols128<56>)) RUN((cols128<72>))
    RUN((rows<52, 1>)) RUN((rows<52, 4>)) RUN((rows<48, 1>)) RUN((rows<36, 1>)) RUN((rows<36, 4>)) RUN((rows<72, 2>))
    RUN(table4)
    RUN((commit128<48, 52>)) RUN((commit128<48, 48>)) RUN((commit128<32, 36>))
  }
  printf("done\n");
  return 0;
}
