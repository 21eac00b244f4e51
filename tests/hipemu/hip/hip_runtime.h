// TEST INFRASTRUCTURE ONLY — a host-side stand-in for <hip/hip_runtime.h>.
//
// The build container has hipcc but no GPU.  To debug kernel index logic before spending GPU
// minutes, tests/hipemu compiles the *unchanged* product HIP sources (microwakeword_amd/csrc)
// as host C++ with this header shadowing the real one (-I tests/hipemu comes first).  Every
// workgroup is run as a set of ucontext fibers on one OS thread:
//   * __syncthreads()            -> fiber yields until every live thread of the block arrived
//   * __shfl*/mfma (wave ops)    -> fiber yields until all 64 lanes of its wave arrived
//   * fibers are scheduled in a selectable order (HIPEMU_ORDER=0 forward, 1 reverse,
//     2 waves reversed) so a missing barrier shows up as a deterministic mismatch
//   * device memory = host memory with an inaccessible guard page right after each allocation
//   * v_mfma_f32_16x16x4_f32 follows the lane maps of cdna_hip_programming.md §3
// The product never links this: microwakeword_amd loads only libmww_hip.so (real HIP).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
using std::min;
using std::max;

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
struct ushort2 { unsigned short x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline int2 make_int2(int a, int b) { return {a, b}; }

namespace hipemu {
extern uint3_emu g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
void sync_block();
unsigned wave_exchange(unsigned v, int src_lane);                 // 32-bit shuffle primitive
void wave_exchange2(float a, float b, const float** A, const float** B);  // publish 2 floats, get arrays
const unsigned long long* wave_publish2(unsigned long long a, unsigned long long b);  // publish 2 x 64 bit, get [64][2]
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void spin_yield();                                                // a thread polling memory lets every other fiber run
int lane_id();
void* dyn_shared();                                               // dynamic LDS of the running block
}  // namespace hipemu

#define threadIdx (hipemu::g_threadIdx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
#define warpSize 64

static inline void __syncthreads() { hipemu::sync_block(); }
static inline float __shfl_xor(float v, int mask, int = 64) {
  unsigned u; std::memcpy(&u, &v, 4);
  u = hipemu::wave_exchange(u, hipemu::lane_id() ^ mask);
  std::memcpy(&v, &u, 4); return v;
}
static inline int __shfl_xor(int v, int mask, int = 64) { return (int)hipemu::wave_exchange((unsigned)v, hipemu::lane_id() ^ mask); }
static inline float __shfl(float v, int src, int = 64) {
  unsigned u; std::memcpy(&u, &v, 4);
  u = hipemu::wave_exchange(u, src & 63);
  std::memcpy(&v, &u, 4); return v;
}
static inline int __shfl(int v, int src, int = 64) { return (int)hipemu::wave_exchange((unsigned)v, src & 63); }
static inline float __shfl_down(float v, unsigned d, int = 64) {
  int l = hipemu::lane_id(); int s = l + (int)d; if (s > 63) s = l;
  unsigned u; std::memcpy(&u, &v, 4);
  u = hipemu::wave_exchange(u, s);
  std::memcpy(&v, &u, 4); return v;
}

typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
// D = A(16x4) * B(4x16) + C ; lane l: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; C/D[row=(l>>4)*4+r][col=l&15]
static inline hipemu_f32x4 hipemu_mfma_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  const float *A, *B;
  hipemu::wave_exchange2(a, b, &A, &B);
  int l = hipemu::lane_id();
  int col = l & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = std::fmaf(A[k * 16 + row], B[k * 16 + col], acc);  // k-ordered fma chain
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4f32
// D = A(16x16) * B(16x16) + C with bf16 operands: lane l holds A[i=l&15][k=4*(l>>4)+j], B[k=4*(l>>4)+j][n=l&15]
typedef short hipemu_s16x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4 hipemu_mfma_16x16x16bf16(hipemu_s16x4 a, hipemu_s16x4 b, hipemu_f32x4 c, int, int, int) {
  unsigned long long ua, ub;
  std::memcpy(&ua, &a, 8);
  std::memcpy(&ub, &b, 8);
  const unsigned long long* all = hipemu::wave_publish2(ua, ub);
  auto elem = [&](int lane, int which, int j) {
    unsigned bits = (unsigned)((all[lane * 2 + which] >> (16 * j)) & 0xFFFFull) << 16;
    float f; std::memcpy(&f, &bits, 4); return f;
  };
  const int l = hipemu::lane_id(), col = l & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int gg = 0; gg < 4; ++gg)
      for (int j = 0; j < 4; ++j) acc += elem(gg * 16 + row, 0, j) * elem(gg * 16 + col, 1, j);
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x16bf16_1k hipemu_mfma_16x16x16bf16
// D = A(16x32) * B(32x16) + C with bf16 operands: lane l holds A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][n=l&15], e < 8
// (verified on the device by tools/ubench/mfma_bf16x9: the split products reproduce the float64 reference)
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x4 hipemu_mfma_16x16x32bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
  unsigned long long ua[2], ub[2];
  std::memcpy(ua, &a, 16);
  std::memcpy(ub, &b, 16);
  unsigned long long A[64][2], B[64][2];
  const unsigned long long* all = hipemu::wave_publish2(ua[0], ua[1]);
  std::memcpy(A, all, sizeof(A));
  all = hipemu::wave_publish2(ub[0], ub[1]);
  std::memcpy(B, all, sizeof(B));
  auto elem = [](const unsigned long long (*M)[2], int lane, int e) {
    unsigned bits = (unsigned)((M[lane][e >> 2] >> (16 * (e & 3))) & 0xFFFFull) << 16;
    float f; std::memcpy(&f, &bits, 4); return f;
  };
  const int l = hipemu::lane_id(), col = l & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int gg = 0; gg < 4; ++gg)
      for (int e = 0; e < 8; ++e) acc += elem(A, gg * 16 + row, e) * elem(B, gg * 16 + col, e);
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_16x16x32bf16
// ds_read_b64_tr_b16 (gfx950 LDS transpose read), semantics probed on the device with tools/ubench/tr16_probe: within each group
// of 16 lanes, lane p supplies the 8-byte-aligned address of four 16-bit elements in[p][0..3]; lane i receives
// out[j] = in[4 j + (i >> 2)][i & 3], j < 4
typedef short hipemu_s16x4v __attribute__((ext_vector_type(4)));
static inline hipemu_s16x4v hipemu_ds_read_tr16(const void* p) {
  unsigned long long mine;
  std::memcpy(&mine, p, 8);
  if (((uintptr_t)p & 7) != 0) { std::fprintf(stderr, "hipemu: ds_read_b64_tr_b16 address not 8-byte aligned\n"); std::abort(); }
  const unsigned long long* all = hipemu::wave_publish2(mine, 0ull);
  const int l = hipemu::lane_id(), grp = l & ~15, i = l & 15;
  hipemu_s16x4v out;
  for (int j = 0; j < 4; ++j) out[j] = (short)((all[(grp + 4 * j + (i >> 2)) * 2] >> (16 * (i & 3))) & 0xFFFFull);
  return out;
}

#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu_ds_read_tr16((const void*)(p))
static inline unsigned long long hipemu_memtime() { return 0ull; }
#define __builtin_amdgcn_s_memtime hipemu_memtime

// buffer resources (common.hip.h "bounds-checked tile access"): base + byte count; loads past the count
// return 0, stores past it are dropped — the semantics of a raw (stride 0) gfx9 buffer descriptor
struct hipemu_rsrc { char* base; long long bytes; };
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline hipemu_rsrc hipemu_make_rsrc(void* p, short, int bytes, int) { return {static_cast<char*>(p), bytes}; }
static inline hipemu_u32x4 hipemu_buf_load128(hipemu_rsrc r, int off, int soff, int) {
  hipemu_u32x4 v = {0u, 0u, 0u, 0u};
  const long long o = (long long)(unsigned)off + (unsigned)soff;
  if (o + 16 <= r.bytes) std::memcpy(&v, r.base + o, 16);
  return v;
}
static inline hipemu_u32x2 hipemu_buf_load64(hipemu_rsrc r, int off, int soff, int) {
  hipemu_u32x2 v = {0u, 0u};
  const long long o = (long long)(unsigned)off + (unsigned)soff;
  if (o + 8 <= r.bytes) std::memcpy(&v, r.base + o, 8);
  return v;
}
static inline unsigned hipemu_buf_load32(hipemu_rsrc r, int off, int soff, int) {
  unsigned v = 0u;
  const long long o = (long long)(unsigned)off + (unsigned)soff;
  if (o + 4 <= r.bytes) std::memcpy(&v, r.base + o, 4);
  return v;
}
static inline void hipemu_buf_store32(unsigned v, hipemu_rsrc r, int off, int soff, int) {
  const long long o = (long long)(unsigned)off + (unsigned)soff;
  if (o + 4 <= r.bytes) std::memcpy(r.base + o, &v, 4);
}
static inline void hipemu_buf_store16(unsigned short v, hipemu_rsrc r, int off, int soff, int) {
  const long long o = (long long)(unsigned)off + (unsigned)soff;
  if (o + 2 <= r.bytes) std::memcpy(r.base + o, &v, 2);
}
#define __builtin_amdgcn_raw_buffer_store_b16 hipemu_buf_store16
#define __amdgpu_buffer_rsrc_t hipemu_rsrc
#define __builtin_amdgcn_make_buffer_rsrc hipemu_make_rsrc
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu_buf_load128
#define __builtin_amdgcn_raw_buffer_load_b64 hipemu_buf_load64
#define __builtin_amdgcn_raw_buffer_load_b32 hipemu_buf_load32
#define __builtin_amdgcn_raw_buffer_store_b32 hipemu_buf_store32

static inline void hipemu_setprio(int) {}
#define __builtin_amdgcn_s_setprio hipemu_setprio
static inline void hipemu_sched_group_barrier(int, int, int) {}
#define __builtin_amdgcn_sched_group_barrier hipemu_sched_group_barrier
static inline void hipemu_sched_barrier(int) {}
#define __builtin_amdgcn_sched_barrier hipemu_sched_barrier
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { auto o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline double unsafeAtomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
// (__hip_atomic_load / _store / _fetch_add are clang builtins on the host too)
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __builtin_amdgcn_s_sleep(int) { hipemu::spin_yield(); }
static inline float __ldg(const float* p) { return *p; }
static inline int hipemu_rfl(int v) { return v; }
#define __builtin_amdgcn_readfirstlane hipemu_rfl
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float fmaxf_emu(float a, float b) { return a > b ? a : b; }

// ------------------------------------------------------------------ runtime API subset
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotSupported = 801 };
struct hipemu_stream { bool capturing = false; struct hipemu_graph* g = nullptr; };
typedef hipemu_stream* hipStream_t;
struct hipemu_event { double t = 0; struct hipemu_graph* g = nullptr; };
typedef hipemu_event* hipEvent_t;
struct hipemu_graph { std::vector<std::function<void()>> nodes; std::vector<hipemu_stream*> joined; };
typedef hipemu_graph* hipGraph_t;
typedef hipemu_graph* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipEventDefault = 0, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; char gcnArchName[256]; size_t totalGlobalMem; };

hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipHostGetDevicePointer(void** dptr, void* hptr, unsigned flags = 0);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);

namespace hipemu {
void enqueue(hipStream_t st, std::function<void()> fn);  // runs now, or records when capturing
}

enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// LDS-only model of a CU with 160 KB: enough to make the per-launch grids of the conv/BN graph kernels differ from op to op
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t lds) {
  *n = (int)(163840 / (lds + 4096));
  if (*n > 8) *n = 8;
  return hipSuccess;
}
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_shared());
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                  \
  do {                                                                                               \
    dim3 hipemu_g = (grid), hipemu_b = (block);                                                      \
    hipemu::enqueue((stream), [=]() { hipemu::launch(hipemu_g, hipemu_b, (shmem), [=]() { kernel(__VA_ARGS__); }); }); \
  } while (0)
